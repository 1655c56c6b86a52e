"""SURVEY 8(f) row 2, the input side on the GPU: pst_image_prepare (ImgNorm + antialiased bilinear resize + crop of a decoded uint8 image)
against torch's own antialiased interpolation on the CPU (what torchvision.transforms.Resize applies to a tensor), and pst_patch_rows (the
patch rows of both ViTs in one launch) bit for bit against the separate patchify / DINOv2-preprocess kernels it replaces."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('Hs,Ws,size', [(480, 640, 512), (1200, 1600, 512), (333, 517, 224), (640, 480, 512), (96, 128, 512), (1000, 1000, 512), (540, 960, 512), (500, 700, 384)])
def test_image_prepare_matches_torch_antialias(Hs, Ws, size):
    from panst3r_amd.engine.images import prepare_image, resize_recipe
    g = np.random.Generator(np.random.PCG64(Hs * 7 + Ws))
    img = g.integers(0, 256, size=(Hs, Ws, 3), dtype=np.uint8)
    img[:, :, 1] = (np.arange(Ws)[None, :] * 255 // Ws).astype(np.uint8)          # a ramp: exposes coordinate / transposition mistakes
    out = prepare_image(img, size, 16, DEV)
    (top, left), (Hc, Wc), (Ho, Wo) = resize_recipe(size, 16, Hs, Ws)          # centre crop to the trained aspect ratio, then resize
    assert out.shape == (3, Ho, Wo) and max(Ho, Wo) == size and Ho % 16 == 0 and Wo % 16 == 0
    t = (torch.from_numpy(img).permute(2, 0, 1).float() / 255.0 - 0.5) / 0.5        # ToTensor + Normalize(0.5, 0.5) == ImgNorm
    ref = F.interpolate(t[None][:, :, top:top + Hc, left:left + Wc], size=(Ho, Wo), mode='bilinear', align_corners=False, antialias=True)[0]
    assert float((out.cpu() - ref).abs().max()) < 2e-5
    assert float(out.min()) >= -1.0001 and float(out.max()) <= 1.0001


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('H,W', [(64, 96), (384, 512), (96, 64)])
def test_patch_rows_bit_identical_to_separate_kernels(dtype, H, W):
    from panst3r_amd import hip
    n = 3
    g = np.random.Generator(np.random.PCG64(H + W))
    img = torch.from_numpy(g.uniform(-1, 1, size=(n, 3, H, W)).astype(np.float32)).to(DEV)
    T = (H // 16) * (W // 16)
    enc = torch.full((n * T, 768), 7.0, dtype=dtype, device=DEV)
    dino = torch.full((n * T, 640), 7.0, dtype=dtype, device=DEV)
    hip.patch_rows(img, enc=enc, dino=dino, p_enc=16, p_dino=14)
    ref_e = torch.full((n * T, 768), 7.0, dtype=dtype, device=DEV)
    hip.patchify(img, ref_e, 16)
    pre = torch.empty(n, 3, H // 16 * 14, W // 16 * 14, device=DEV)
    hip.dino_preprocess(img, pre)
    ref_d = torch.full((n * T, 640), 7.0, dtype=dtype, device=DEV)
    hip.patchify(pre, ref_d, 14)
    assert torch.equal(enc, ref_e) and torch.equal(dino, ref_d)
    # either output alone
    e2 = torch.empty_like(enc); hip.patch_rows(img, enc=e2, p_enc=16)
    d2 = torch.empty_like(dino); hip.patch_rows(img, dino=d2, p_enc=16, p_dino=14)
    assert torch.equal(e2, ref_e) and torch.equal(d2, ref_d)
    # DINOv2 on the transposed image (portrait views) without a transposed copy
    imt = img.transpose(2, 3).contiguous()
    pre_t = torch.empty(n, 3, W // 16 * 14, H // 16 * 14, device=DEV)
    hip.dino_preprocess(imt, pre_t)
    ref_t = torch.empty_like(dino); hip.patchify(pre_t, ref_t, 14)
    d3 = torch.empty_like(dino); hip.patch_rows(img, dino=d3, p_enc=16, p_dino=14, dino_transposed=True)
    assert torch.equal(d3, ref_t)


def test_load_images_pair_and_shapes(tmp_path):
    """reference load_images contract (tools/demo_panst3r.py:94-114): dict(img, true_shape), long side = size, a lone image is duplicated."""
    import PIL.Image
    from panst3r_amd.engine.images import load_images
    g = np.random.Generator(np.random.PCG64(3))
    arr = g.integers(0, 256, size=(300, 400, 3), dtype=np.uint8)
    path = tmp_path / 'a.png'
    PIL.Image.fromarray(arr).save(path)
    views = load_images([str(path)], size=512, patch_size=16, verbose=False, device=DEV)
    assert len(views) == 2 and views[0]['img'].shape == (3, 384, 512) and list(views[0]['true_shape']) == [384, 512]
    again = load_images([arr, arr[:, ::-1].copy()], size=224, verbose=False, device=DEV)
    assert again[0]['img'].shape == (3, 224, 224)              # 224 is a trained resolution: centre crop to 1:1, then resize
    assert float((again[0]['img'].flip(-1) - again[1]['img']).abs().max()) < 1e-4          # the filter is symmetric: mirrored input, mirrored output
