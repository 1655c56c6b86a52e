"""The resize recipe of the input side (panst3r_amd/engine/images.py; reference tools/demo_panst3r.py:94-114 -> must3r.tools.image.get_resize_function,
un-vendored: restated, parity unpinned): known answers for the trained-resolution table and the crop arithmetic.  Host arithmetic only."""
import pytest

from panst3r_amd.engine.images import resize_recipe, target_resolution, RATIOS_RESOLUTIONS


@pytest.mark.parametrize('H,W,size,want', [
    (480, 640, 512, (384, 512)), (1000, 1000, 512, (384, 512)), (1080, 1920, 512, (288, 512)), (500, 1600, 512, (160, 512)), (512, 1024, 512, (256, 512)),
    (1050, 1600, 512, (336, 512)), (1920, 1080, 512, (512, 288)), (640, 480, 512, (512, 384)), (640, 480, 224, (224, 224)), (333, 517, 224, (224, 224))])
def test_trained_resolution_table(H, W, size, want):
    assert target_resolution(H, W, size) == want
    (top, left), (Hc, Wc), out = resize_recipe(size, 16, H, W)
    assert out == want and 0 < Hc <= H and 0 < Wc <= W and top == (H - Hc) // 2 and left == (W - Wc) // 2
    assert abs(Wc / Hc - want[1] / want[0]) < 2.0 / min(Hc, Wc)          # the crop has the target's aspect ratio (to the integer)
    assert Hc == H or Wc == W                                            # only one side is cropped


def test_square_image_is_cropped_not_stretched():
    (top, left), (Hc, Wc), out = resize_recipe(512, 16, 1000, 1000)
    assert out == (384, 512) and (Hc, Wc) == (750, 1000) and (top, left) == (125, 0)


@pytest.mark.parametrize('size', [336, 384, 448, 768])
def test_sizes_without_a_table_entry(size):
    """the demo's other --image_size choices (tools/demo_panst3r.py:72): long side -> size, short side scaled and centre-cropped to the patch grid"""
    assert size not in RATIOS_RESOLUTIONS
    (top, left), (Hc, Wc), (Ho, Wo) = resize_recipe(size, 16, 480, 640)
    assert Wo == size and Ho % 16 == 0 and Ho == int(round(480 * size / 640)) // 16 * 16
    assert 0 <= top and 0 <= left and Hc <= 480 and Wc <= 640
