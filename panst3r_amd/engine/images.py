"""Input side on the device: the reference's `load_images` (tools/demo_panst3r.py:94-114) with the pixel work on the GPU.

Reference flow per file:  PIL decode -> ImgNorm (ToTensor + Normalize(0.5, 0.5)) -> `get_resize_function(size, patch_size, H, W)` (centre-crop
to the nearest trained aspect ratio, resize to that trained resolution: `resize_recipe`) -> fp32 [3, H, W] in [-1, 1].  Here the decoded
uint8 image is uploaded as is (3 bytes per pixel instead of 12) and ONE kernel (pst_image_prepare) does ToTensor + Normalize + antialiased
bilinear resize + crop; the patch rows of both ViTs then come from one more launch (hip.patch_rows, used by PanSt3R.encode_views).

`must3r.tools.image.get_resize_function` and `must3r.datasets.ImgNorm` are un-vendored ([3P-recalled]): `resize_recipe` restates the size
arithmetic (parity unpinned); the pixel arithmetic - torchvision's Resize on a tensor = torch's antialiased bilinear
interpolation - is checked against torch on the CPU in tests/test_hip_input.py.
"""
import numpy as np
import torch

from .. import hip


# Resolutions the MUSt3R / PanSt3R checkpoints were trained on, keyed by the long side and the aspect ratio ([3P-recalled] the table of
# `must3r.tools.image`: 512 -> 384x512 / 336x512 / 288x512 / 256x512 / 160x512, 224 -> 224x224; cf. the train resolutions of
# configs/base.yaml:46 of the reference: [512,384] [512,336] [512,288] [512,256] [512,160]).
RATIOS_RESOLUTIONS = {
    224: {1.0: (224, 224)},
    512: {4 / 3: (384, 512), 32 / 21: (336, 512), 16 / 9: (288, 512), 2 / 1: (256, 512), 16 / 5: (160, 512)},
}


def target_resolution(H, W, size):
    """(Ht, Wt) of the trained-resolution table for an H x W image: the entry whose aspect ratio is nearest to the image's (portrait images
    get the transposed entry).  Only for `size` in RATIOS_RESOLUTIONS."""
    table = RATIOS_RESOLUTIONS[size]
    ratios = np.array(list(table))
    landscape = W >= H
    r = W / H
    sel = ratios[np.argmin(np.abs((r if landscape else 1.0 / r) - ratios))]
    h, w = table[float(sel)]
    return (h, w) if landscape else (w, h)


def resize_recipe(size, patch_size, H, W):
    """(crop origin (top, left), crop size (Hc, Wc)) in the SOURCE image and the output shape (Ho, Wo).
    [3P-recalled, parity unpinned: `must3r.tools.image.get_resize_function` is not vendored under /root/reference]
      * size in {224, 512} (the trained tables): centre-crop the image to the aspect ratio of the nearest table entry, then resize the crop to
        that entry - square 1000 x 1000 at 512 -> 384 x 512 (not 512 x 512), 640 x 480 at 224 -> 224 x 224: shapes the checkpoints know;
      * other sizes of the demo's --image_size list (336, 384, 448, 768; tools/demo_panst3r.py:72) have no table entry: the long side goes to
        `size`, the short side is scaled with it and cropped (centre) to a multiple of the patch size - a restatement without any
        reference to pin it, flagged as such in DESIGN.md."""
    if size in RATIOS_RESOLUTIONS:
        Ho, Wo = target_resolution(H, W, size)
        assert Ho % patch_size == 0 and Wo % patch_size == 0, (Ho, Wo, patch_size)
        ratio, tr = W / H, Wo / Ho
        if abs(ratio - tr) < np.finfo(np.float32).eps:
            Hc, Wc = H, W
        elif ratio < tr:
            Hc, Wc = int(W / tr), W
        else:
            Hc, Wc = H, int(H * tr)
        return ((H - Hc) // 2, (W - Wc) // 2), (Hc, Wc), (Ho, Wo)
    from ..model.common import warn_once
    warn_once(('resize_recipe', size), 'panst3r_amd.load_images: image_size %d has no entry in the trained-resolution table (224, 512): the resize / crop rule '
                                       'for it is this build\'s restatement of un-vendored must3r code and is not pinned against upstream' % size)
    scale = float(size) / max(H, W)
    Hr, Wr = max(int(round(H * scale)), patch_size), max(int(round(W * scale)), patch_size)
    Ho, Wo = Hr // patch_size * patch_size, Wr // patch_size * patch_size
    # crop in the source so that the resize hits (Ho, Wo) exactly: the part of the resized image the centre crop would have kept
    Hc, Wc = min(H, int(round(Ho / scale))), min(W, int(round(Wo / scale)))
    return ((H - Hc) // 2, (W - Wc) // 2), (Hc, Wc), (Ho, Wo)


def prepare_image(rgb_u8, size, patch_size=16, device='cuda'):
    """decoded image uint8 [H, W, 3] (numpy or tensor) -> fp32 [3, Ho, Wo] in [-1, 1] on `device` (the model's input format)."""
    t = torch.from_numpy(np.array(rgb_u8, copy=True)) if isinstance(rgb_u8, np.ndarray) else torch.as_tensor(rgb_u8)
    assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3, 'expected a decoded RGB image, uint8 [H, W, 3]'
    (top, left), (Hc, Wc), (Ho, Wo) = resize_recipe(size, patch_size, t.shape[0], t.shape[1])
    t = t[top:top + Hc, left:left + Wc].to(device).contiguous()           # the centre crop is a view: only the kept pixels are uploaded
    out = torch.empty(3, Ho, Wo, dtype=torch.float32, device=device)
    return hip.image_prepare(t, out, (Ho, Wo), (0, 0))


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
