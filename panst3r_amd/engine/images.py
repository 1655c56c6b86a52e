"""Input side on the device: the reference's `load_images` (tools/demo_panst3r.py:94-114) with the pixel work on the GPU.

Reference flow per file:  PIL decode -> ImgNorm (ToTensor + Normalize(0.5, 0.5)) -> `get_resize_function(size, patch_size, H, W)` (resize so
that the long side is `size`, then centre-crop both sides to multiples of the patch size) -> fp32 [3, H, W] in [-1, 1].  Here the decoded
uint8 image is uploaded as is (3 bytes per pixel instead of 12) and ONE kernel (pst_image_prepare) does ToTensor + Normalize + antialiased
bilinear resize + crop; the patch rows of both ViTs then come from one more launch (hip.patch_rows, used by PanSt3R.encode_views).

`must3r.tools.image.get_resize_function` and `must3r.datasets.ImgNorm` are un-vendored ([3P-recalled]): `resize_recipe` restates the size
arithmetic (parity unpinned); the pixel arithmetic - torchvision's Resize on a tensor = torch's antialiased bilinear
interpolation - is checked against torch on the CPU in tests/test_hip_input.py.
"""
import numpy as np
import torch

from .. import hip


def resize_recipe(size, patch_size, H, W):
    """(Hr, Wr), (top, left), (Hc, Wc): resize target, crop origin and final shape for an H x W image.
    Long side -> `size` (aspect ratio kept, rounded), then centre crop to multiples of `patch_size`."""
    scale = float(size) / max(H, W)
    Hr, Wr = max(int(round(H * scale)), patch_size), max(int(round(W * scale)), patch_size)
    Hc, Wc = Hr // patch_size * patch_size, Wr // patch_size * patch_size
    return (Hr, Wr), ((Hr - Hc) // 2, (Wr - Wc) // 2), (Hc, Wc)


def prepare_image(rgb_u8, size, patch_size=16, device='cuda'):
    """decoded image uint8 [H, W, 3] (numpy or tensor) -> fp32 [3, Hc, Wc] in [-1, 1] on `device` (the model's input format)."""
    t = torch.from_numpy(np.array(rgb_u8, copy=True)) if isinstance(rgb_u8, np.ndarray) else torch.as_tensor(rgb_u8)
    assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3, 'expected a decoded RGB image, uint8 [H, W, 3]'
    t = t.to(device).contiguous()
    (Hr, Wr), (top, left), (Hc, Wc) = resize_recipe(size, patch_size, t.shape[0], t.shape[1])
    out = torch.empty(3, Hc, Wc, dtype=torch.float32, device=device)
    return hip.image_prepare(t, out, (Hr, Wr), (top, left))


def load_images(folder_content, size, patch_size=16, normalization='dust3r', verbose=True, device='cuda'):
    """Reference signature (tools/demo_panst3r.py:94) + `device`: list of file paths (or already decoded uint8 [H,W,3] arrays) ->
    list of dict(img=fp32 [3,H,W] on the device, true_shape=np.int32([H, W])); a single image is duplicated into a pair (:111-112)."""
    if normalization.lower() != 'dust3r':
        raise ValueError(f'did not recognize image {normalization=}')
    imgs = []
    for item in folder_content:
        if isinstance(item, (str, bytes)) or hasattr(item, '__fspath__'):
            import PIL.Image                      # decode on the host, as the reference does
            with PIL.Image.open(item) as im:
                rgb = np.asarray(im.convert('RGB'))
        else:
            rgb = item
        t = prepare_image(rgb, size, patch_size, device)
        imgs.append(dict(img=t, true_shape=np.int32([t.shape[-2], t.shape[-1]])))
        if verbose:
            print(f' - adding image with resolution {rgb.shape[1]}x{rgb.shape[0]} --> {t.shape[-1]}x{t.shape[-2]}')
    if len(imgs) == 1:
        imgs = imgs * 2       # create pair
    return imgs
