"""pcm_amd — MI355X-native phased-consistency (PCM-LoRA) distillation for SD1.5.

Host side (Python, mirroring the reference's train_pcm_lora_sd15.py operator surface) over the
C-ABI HIP library ``libpcm_hip.so`` (include/pcm_hip.h).  There is no CPU or torch-op fallback:
importing ``pcm_amd.capi.lib()`` raises if the HIP library has not been built.
"""
__version__ = "0.1.0"
