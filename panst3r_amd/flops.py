"""Algorithmic FLOP / byte model of the PanSt3R forward path; 2*M*N*K per contraction.

The counts are the DEDUPLICATED ones the roofline fraction is quoted against: memory K/V projected once per scene,
one full-resolution mask einsum per view plus 6 x 1/16 attention-mask einsums per keyframe.
"""


def _vit_block(T, D, ff, Nk=None):
    Nk = T if Nk is None else Nk
    return 2 * T * D * 3 * D + 2 * T * D * D + 2 * 2 * T * D * ff + 2 * 2 * T * Nk * D


def encoder_flops(H, W, D=1024, depth=24, patch=16):
    T = (H // patch) * (W // patch)
    return 2 * T * (3 * patch * patch) * D + depth * _vit_block(T, D, 4 * D)


def dino_flops(H, W, D=1024, depth=24):
    T = (H // 16) * (W // 16)
    return 2 * T * (3 * 14 * 14) * D + depth * _vit_block(T + 1, D, 4 * D)


def decoder_render_flops(H, W, K, De=1024, D=768, depth=12, patch=16, ch=7):
    """one rendered view against a K-keyframe memory (cached K/V)."""
    T = (H // patch) * (W // patch)
    per_layer = _vit_block(T, D, 4 * D) + 2 * T * D * D * 2 + 2 * 2 * T * (K * T) * D      # + cross q/out proj + cross attention
    return 2 * T * De * D + depth * per_layer + 2 * T * D * ch * patch * patch


def memory_kv_flops(H, W, K, D=768, depth=12, patch=16):
    T = (H // patch) * (W // patch)
    return depth * 2 * (K * T) * D * (2 * D)


def memory_build_flops(H, W, K, **kw):
    """sequential build: image j renders against j earlier images (+ feedback MLP), SURVEY: 177.3 K + 10.87 K (K-1) G."""
    T = (H // 16) * (W // 16)
    tot = 0
    for j in range(K):
        ctx = 1 if j < 2 else j
        tot += decoder_render_flops(H, W, 0, **kw) + 12 * 2 * 2 * T * (ctx * T) * 768 + 2 * 2 * T * 768 * 3072
    return tot


def upscaler_flops(H, W, variant):
    T = (H // 16) * (W // 16)
    P = (H // 2) * (W // 2)
    if variant == 'v1':
        return (2 * T * 2816 * 11264 * 2 + 2 * T * 11264 * (768 + 2048) + 2 * 4 * T * (512 * 2048 + 2048 * 1536)
                + 2 * 16 * T * (384 * 1536 + 1536 * 1024))
    mixer = 2 * T * 2816 * 768 + 3 * _vit_block(T, 768, 3072)
    convs = 2 * P * 9 * (203 * 384 + 384 * 384)
    blocks = 2 * (2 * P * 384 * 384 * 4 + 2 * 2 * P * T * 384 + 2 * 2 * T * 384 * 384)
    return mixer + 2 * T * 768 * 768 + 2 * T * 788 * 384 + convs + blocks


def mask_einsum_flops(H, W, variant, Q=200):
    return 2 * Q * (256 if variant == 'v1' else 384) * (H // 2) * (W // 2)


def query_decoder_flops(H, W, K, Q=200, d=768, ff=2048, layers=6):
    T = (H // 16) * (W // 16)
    per = 2 * 2 * (K * T) * d * d + 2 * 2 * Q * (K * T) * d + 2 * Q * d * d * 2 + 2 * Q * d * 3 * d + 2 * 2 * Q * Q * d \
        + 2 * Q * d * d + 2 * 2 * Q * d * ff
    return layers * per


def scene_flops(H, W, V, K, variant):
    per_view = (encoder_flops(H, W) + dino_flops(H, W) + decoder_render_flops(H, W, K) + upscaler_flops(H, W, variant)
                + mask_einsum_flops(H, W, variant))
    extra = memory_build_flops(H, W, K) + memory_kv_flops(H, W, K) + query_decoder_flops(H, W, K) \
        + K * 6 * mask_einsum_flops(H, W, variant) / 16
    return V * per_view + extra


def mask_head_bytes(H, W, variant, Q=200):
    P = (H // 2) * (W // 2)
    return (256 if variant == 'v1' else 384) * P * 2 + Q * P * 4
