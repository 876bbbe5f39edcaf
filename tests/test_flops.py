"""The algorithmic-work walker behind bench.py's roofline numbers (pcm_amd/flops.py) reproduces the counts SURVEY.md section 8(d) /
BASELINE.md section 2 derive for SD1.5 by hand; SDXL and SD3-medium come from the same walk."""
from pcm_amd import flops as F
from pcm_amd.discriminator import ADAPTER_DIMS
from pcm_amd.mmdit_spec import MMDiTConfig
from pcm_amd.unet_spec import UNetConfig


def test_sd15_counts_match_the_survey():
    m = F.unet_macs(UNetConfig.sd15())
    assert abs(m["base"] / 1e9 - 401.64) < 0.01 and abs(m["lora"] / 1e9 - 47.16) < 0.01           # base forward / LoRA r=64 extra, GMAC
    assert abs(m["attn_core"] / 1e9 - (61.25 + 1.78)) < 0.01                                       # self + cross attention cores
    kinds = m["by_kind"]
    conv_resnet = (kinds["conv3x3:conv1"] + kinds["conv3x3:conv2"]) / 1e9
    assert abs(conv_resnet - 163.26) < 0.01 and abs(kinds["conv3x3:conv"] / 1e9 - 36.81) < 0.01     # resnet convs, sampler convs
    t = F.step_tflop(m)
    assert abs(t["student_fwd"] - 0.8976) < 1e-4 and abs(t["teacher_fwd"] - 0.8033) < 1e-4
    assert abs(t["backward"] - 1.12) < 5e-3 and abs(t["step"] - 4.52) < 5e-3
    heads = F.heads_macs(ADAPTER_DIMS, (32, 16, 8, 8, 8, 16, 32, 64, 64)) / 1e9                    # tap resolutions of modified_forward
    assert abs(heads - 339.8) < 0.1


def test_sdxl_and_sd3_walks():
    x = F.step_tflop(F.unet_macs(UNetConfig.sdxl(), 128, 128, 77, 64))
    s = F.step_tflop(F.mmdit_macs(MMDiTConfig.sd3_medium()))
    # SDXL at 1024 px: ~6.8 TFLOP per teacher forward, ~36.8 per sample-step; SD3-medium at 4096 + 154 tokens: ~8.4 / ~45.5
    assert 6.5 < x["teacher_fwd"] < 7.0 and 36 < x["step"] < 38
    assert 8.2 < s["teacher_fwd"] < 8.7 and 44 < s["step"] < 47
