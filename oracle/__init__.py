"""CPU oracle for the PCM-LoRA SD1.5 distillation step.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the shipped product path.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and only as the checker.  The product (``phased-consistency-model_amd/``) never imports
this package and fails loudly when its HIP library is missing.

Pinning status
--------------
* reference-OWNED math (DDIMSolver, predicted_origin, boundary scalings, add_noise,
  noise_travel, Huber/L2 loss, update_ema, hinge losses): ``oracle/pcm_math.py`` is a
  restatement; it is pinned bit-exactly against the reference's own source, executed by
  AST-slicing it out of ``/root/reference`` (``oracle/ref_slice.py``, build container only)
  and against the fixtures in ``tests/golden/`` that script generated.
* third-party-owned math (diffusers 0.26.3 ``UNet2DConditionModel``, peft 0.9.0 LoRA):
  the reference vendors neither and holds no tests / golden vectors for them, and neither
  wheel is installable here.  ``oracle/unet_sd15.py`` restates the published module
  semantics in plain torch fp32; anchors are the exact parameter counts
  (859 520 964 base / 67 252 224 LoRA r=64) and state-dict key set.  **parity unpinned**
  for that part (no reference output exists to pin against).
"""
