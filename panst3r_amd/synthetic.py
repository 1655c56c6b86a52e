"""Deterministic synthetic inputs and weights.

There is no network for checkpoints or datasets, so benchmarks and parity tests use
 * images: numpy PCG64(seed_base + view) uniform in [-1, 1], shape [3, H, W]
 * class embeddings: PCG64(99) normal -> L2-normalised [Ncls, 768]
 * weights: a filler keyed by the state-dict key (crc32(key) ^ seed), so 0.3-1.7 GB of weights are
   regenerated bit-identically on any box and never shipped.
Nothing here depends on torch's RNG.
"""
import zlib
import numpy as np
import torch

_EMB_UNIT = ('query_feat.weight', 'query_embed.weight', 'level_embed.weight', '.biases')
_EMB_SMALL = ('position_embeddings', 'cls_token', 'mask_token', 'image2_embed')
_QK_KEYS = ('attn.qkv.weight', 'projq.weight', 'projk.weight', 'query.weight', 'key.weight', 'in_proj_weight')


def _gen(key, seed):
    return np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF))


# "outlier" weight set: trained ViT-L / DINOv2 checkpoints carry a handful of residual-stream channels 10^2 - 10^3 x larger than the
# rest, fed by a few rows of the residual-writing projections (attention output, fc2).  `outlier=S` scales those rows (and their bias entries) of every
# such projection of the three backbones by S for OUTLIER_CHANNELS (taken modulo the layer's width): what an f16 operand path must survive.
OUTLIER_CHANNELS = (7, 133, 402, 911)
_RESIDUAL_WRITERS = ('attn.proj.', 'mlp.fc2.', 'cross_attn.proj.', 'attention.output.dense.')
_OUTLIER_SCOPE = ('must3r_encoder.blocks_enc.', 'must3r_decoder.blocks_dec.', 'dino_encoder.dinov2.encoder.layer.')


def _outlier_rows(key, n_out):
    if not any(s in key for s in _OUTLIER_SCOPE) or not any(s in key for s in _RESIDUAL_WRITERS):
        return None
    return sorted({c % n_out for c in OUTLIER_CHANNELS})


def fill_value(key, shape, seed=0, sharp=1.0, outlier=None):
    """fp32 numpy array for state-dict entry `key`."""
    v = _fill_value(key, shape, seed, sharp)
    if outlier and len(tuple(shape)) in (1, 2) and (key.endswith('weight') or key.endswith('bias')):
        rows = _outlier_rows(key, tuple(shape)[0])
        if rows is not None:
            v = np.array(v, copy=True)
            v[rows] *= np.float32(outlier)
    return v


def _fill_value(key, shape, seed=0, sharp=1.0):
    g = _gen(key, seed)
    shape = tuple(shape)
    if len(shape) == 0:
        return np.asarray(1.0, dtype=np.float32)
    if any(key.endswith(s) for s in _EMB_UNIT):
        return g.standard_normal(shape, dtype=np.float32)
    if any(s in key for s in _EMB_SMALL):
        return 0.02 * g.standard_normal(shape, dtype=np.float32)
    if len(shape) == 1:
        z = g.standard_normal(shape, dtype=np.float32)
        if key.endswith('bias'):
            return 0.02 * z
        return 1.0 + 0.1 * z                      # norm scales, LayerScale lambda
    fan_in = int(np.prod(shape[1:]))
    w = g.standard_normal(shape, dtype=np.float32) / np.float32(np.sqrt(fan_in))
    if sharp != 1.0 and any(key.endswith(s) for s in _QK_KEYS):
        rows = shape[0] if ('proj' in key or 'query' in key or 'key' in key) else 2 * shape[0] // 3
        w[:rows] *= np.float32(sharp)
    return w


@torch.no_grad()
def fill_module_(module, seed=0, sharp=1.0, prefix='', outlier=None):
    """In-place deterministic fill of every parameter/buffer in module.state_dict().  `outlier`: see OUTLIER_CHANNELS (keys are matched on
    prefix + key: pass the prefix the module has in the full model)."""
    for key, t in module.state_dict().items():
        if not t.dtype.is_floating_point:
            continue
        v = torch.from_numpy(fill_value(prefix + key, t.shape, seed, sharp, outlier))
        t.copy_(v.to(t.dtype))
    return module


def synth_image(view, H, W, seed_base=1234):
    g = np.random.Generator(np.random.PCG64(seed_base + view))
    return torch.from_numpy(g.uniform(-1.0, 1.0, size=(3, H, W)).astype(np.float32))


def synth_class_embeddings(n_cls=100, dim=768, seed=99):
    g = np.random.Generator(np.random.PCG64(seed))
    e = g.standard_normal((n_cls, dim)).astype(np.float32)
    e /= np.linalg.norm(e, axis=-1, keepdims=True)
    names = ['c%03d' % i for i in range(n_cls)]
    return names, torch.from_numpy(e)
