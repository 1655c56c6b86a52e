"""flops(H,W,V,K,variant) against the constants of SURVEY.md 8(d)."""
from panst3r_amd import flops as F


def near(x, ref, tol=0.02):
    assert abs(x - ref) / ref < tol, (x, ref)


def test_survey_constants():
    near(F.encoder_flops(384, 512), 523e9)
    near(F.dino_flops(384, 512), 523e9, 0.03)
    near(F.decoder_render_flops(384, 512, 16), 177.3e9 + 21.74e9 * 16)
    near(F.memory_kv_flops(384, 512, 16), 21.74e9 * 16)
    near(F.upscaler_flops(384, 512, 'v1'), 225.2e9)
    near(F.upscaler_flops(384, 512, 'v2'), 475e9, 0.03)
    near(F.mask_einsum_flops(384, 512, 'v1'), 5.03e9)
    near(F.mask_einsum_flops(384, 512, 'v2'), 7.55e9)
    near(F.scene_flops(384, 512, 50, 16, 'v2'), 108.8e12, 0.03)
    assert F.mask_head_bytes(384, 512, 'v1') == 256 * 49152 * 2 + 200 * 49152 * 4
