"""The HIP kernels as torch ops (`torch.ops.panst3r_hip.*`) plus signature-compatible shims for the reference's op-level plug points.

Two layers on top of the C ABI (include/panst3r_hip.h -> panst3r_amd/hip.py):

1. `torch.ops.panst3r_hip.<op>`: every kernel registered with `torch.library.custom_op` (device type "cuda" = ROCm/HIP), mutating the
   caller's output buffers -- the form a maintainer of the reference would call from its modules.  The registered functions are
   thin: tensors in, raw pointers across the C ABI, nothing computed in torch.

2. Shims with the exact signatures of the three places where the reference (a pure-PyTorch repo) already allows or uses a fused op:
     rope_2d(tokens, positions, base, F0)                cuRoPE2D's in-place kernel  (README.md:67-71; croco `curope.rope_2d`)
     MultiheadAttention(embed_dim, num_heads)            `nn.MultiheadAttention` as MaskTransformer calls it
                                                         (mask_transformer.py:314,337-338,372,395-398: seq-first [L,B,E], bool attn_mask
                                                         [B*H,L,S] repeated over heads, returns (out, None))
     mask_einsum(mask_embed, mask_feats)                 torch.einsum("bqc,bnchw->bnqhw", ...)  (mask_transformer.py:280)
   They take and return the reference's tensor layouts (fp32 in / out) and do the layout conversions the fused pipeline of
   panst3r_amd.model avoids; the 16-bit format is the one in effect (model.common.precision, default f16).
"""
from typing import List, Optional

import torch
from torch import Tensor, nn

from . import hip
from .model.common import adt, ceil_to

NS = 'panst3r_hip'
_REGISTERED = []


def _op(name, mutates):
    def deco(fn):
        op = torch.library.custom_op('%s::%s' % (NS, name), fn, mutates_args=mutates, device_types='cuda')
        _REGISTERED.append(name)
        return op
    return deco


def _t3(v):
    return None if v is None or len(v) == 0 else tuple(int(x) for x in v)


# ---------------------------------------------------------------------------------------------------- registered ops
@_op('gemm', ('out',))
def gemm(a: Tensor, w: Tensor, out: Tensor, bias: Optional[Tensor] = None, gamma: Optional[Tensor] = None, res: Optional[Tensor] = None,
         res_mod: int = 0, act: str = 'none', trans_out: bool = False, grp: Optional[List[int]] = None, ps: Optional[List[int]] = None,
         conv: Optional[List[int]] = None, rope_pos: Optional[Tensor] = None, rope_table: Optional[Tensor] = None, kernel: int = 0,
         ln_stats: Optional[Tensor] = None, ln_colsum: Optional[Tensor] = None, ln_eps: float = 0.0) -> None:
    """C = epi(A W^T); with ln_stats the LayerNorm of the rows of A is applied in the epilogue (folded weights, include/panst3r_hip.h)."""
    hip.gemm(a, w, out, bias=bias, gamma=gamma, res=res, res_mod=res_mod, act=act, trans_out=trans_out, grp=_t3(grp), ps=_t3(ps), conv=_t3(conv),
             rope=None if rope_pos is None else (rope_pos, rope_table), kernel=kernel, ln=None if ln_stats is None else (ln_stats, ln_colsum, ln_eps))


@_op('gemm_stream', ('out', 'xcopy', 'stats_out'))
def gemm_stream(a: Tensor, w: Tensor, out: Tensor, xcopy: Tensor, stats_out: Tensor, bias: Optional[Tensor] = None, gamma: Optional[Tensor] = None,
                res: Optional[Tensor] = None, kernel: int = 0) -> None:
    """residual GEMM that writes the fp32 stream `out`, its 16-bit copy and the per-row LayerNorm statistics (fold producer)"""
    hip.gemm(a, w, out, bias=bias, gamma=gamma, res=res, kernel=kernel, xcopy=xcopy, stats_out=stats_out)


@_op('gemm_stream16', ('out', 'stats_out'))
def gemm_stream16(a: Tensor, w: Tensor, out: Tensor, stats_out: Tensor, bias: Optional[Tensor] = None, res: Optional[Tensor] = None, kernel: int = 0) -> None:
    """the same for a 16-bit residual stream (LoftUp blocks): `out` is its own operand copy"""
    hip.gemm(a, w, out, bias=bias, res=res, kernel=kernel, stats_out=stats_out)


@_op('attention', ('out',))
def attention(q: Tensor, k: Tensor, vt: Tensor, out: Tensor, B: int, H: int, Nq: int, Nk: int, hd: int, q_strides: List[int], k_strides: List[int],
              v_strides: List[int], o_strides: List[int], scale: Optional[float] = None, mask: Optional[Tensor] = None,
              mask_strides: Optional[List[int]] = None, nsplit: Optional[int] = None, ws: Optional[Tensor] = None, prescaled: bool = False) -> None:
    hip.attention(q, k, vt, out, B, H, Nq, Nk, hd, tuple(q_strides), tuple(k_strides), tuple(v_strides), tuple(o_strides), scale=scale, mask=mask,
                  mask_strides=tuple(mask_strides) if mask_strides else (0, 0), nsplit=nsplit, ws=ws, prescaled=prescaled)


@_op('layernorm', ('out',))
def layernorm(x: Tensor, gamma: Tensor, beta: Tensor, out: Tensor, eps: float, rows: Optional[int] = None, grp: Optional[List[int]] = None,
              add: Optional[Tensor] = None) -> None:
    hip.layernorm(x, gamma, beta, out, eps, rows=rows, grp=_t3(grp), add=add)


@_op('rope2d_', ('x',))
def rope2d_(x: Tensor, pos: Tensor, table: Tensor, nheads: int, hd: int) -> None:
    hip.rope2d_(x, pos, table, nheads, hd)


@_op('patchify', ('out',))
def patchify(img: Tensor, out: Tensor, p: int) -> None:
    hip.patchify(img, out, p)


@_op('dino_preprocess', ('out',))
def dino_preprocess(img: Tensor, out: Tensor) -> None:
    hip.dino_preprocess(img, out)


@_op('image_prepare', ('out',))
def image_prepare(src_u8: Tensor, out: Tensor, resized: List[int], crop_origin: List[int]) -> None:
    hip.image_prepare(src_u8, out, tuple(resized), tuple(crop_origin))


@_op('patch_rows', ('enc', 'dino'))
def patch_rows(img: Tensor, enc: Tensor, dino: Tensor, p_enc: int = 16, p_dino: int = 14, dino_transposed: bool = False) -> None:
    hip.patch_rows(img, enc=enc, dino=dino, p_enc=p_enc, p_dino=p_dino, dino_transposed=dino_transposed)


@_op('rowstats', ('xcopy', 'stats'))
def rowstats(x: Tensor, xcopy: Tensor, stats: Tensor) -> None:
    hip.rowstats(x, xcopy, stats)


@_op('add_cast', ('out',))
def add_cast(a: Tensor, out: Tensor, b: Optional[Tensor] = None, b_mod: int = 0) -> None:
    hip.add_cast(a, out, b=b, b_mod=b_mod)


@_op('l2norm_rows', ('out',))
def l2norm_rows(x: Tensor, out: Tensor, eps: float) -> None:
    hip.l2norm_rows(x, out, eps)


@_op('split3', ('out',))
def split3(x: Tensor, out: Tensor) -> None:
    hip.split3(x, out)


@_op('mean4', ('Fm',))
def mean4(F: Tensor, Fm: Tensor, nimg: int, Hm: int, Wm: int, C: int) -> None:
    hip.mean4(F, Fm, nimg, Hm, Wm, C)


@_op('resize_bilinear', ('Fd',))
def resize_bilinear(F: Tensor, Fd: Tensor, nimg: int, Hs: int, Ws: int, Hd: int, Wd: int, C: int) -> None:
    hip.resize_bilinear(F, Fd, nimg, Hs, Ws, Hd, Wd, C)


@_op('attn_mask_from_logits', ('mask',))
def attn_mask_from_logits(logits: Tensor, mask: Tensor) -> None:
    hip.attn_mask_from_logits(logits, mask)


@_op('loftup_guidance_gn', ('scratch', 'stats', 'out'))
def loftup_guidance_gn(img: Tensor, biases: Tensor, gamma: Tensor, beta: Tensor, eps: float, scratch: Tensor, stats: Tensor, out: Tensor, nf: int,
                       mm: Optional[Tensor] = None) -> None:
    hip.loftup_guidance_gn(img, biases, gamma, beta, eps, scratch, stats, out, nf, mm)


@_op('loftup_minmax', ('mm',))
def loftup_minmax(img: Tensor, mm: Tensor) -> None:
    hip.loftup_minmax(img, mm)


@_op('minmax_merge', ('out',))
def minmax_merge(mm: Tensor, scope: Tensor, out: Tensor) -> None:
    hip.minmax_merge(mm, scope, out)


@_op('groupnorm_stats', ('stats',))
def groupnorm_stats(x: Tensor, stats: Tensor, nimg: int, P: int, C: int, G: int) -> None:
    hip.groupnorm_stats(x, stats, nimg, P, C, G)


@_op('groupnorm_apply', ('out',))
def groupnorm_apply(x: Tensor, stats: Tensor, gamma: Tensor, beta: Tensor, out: Tensor, nimg: int, P: int, C: int, G: int, eps: float, relu: bool) -> None:
    hip.groupnorm_apply(x, stats, gamma, beta, out, nimg, P, C, G, eps, relu)


@_op('loftup_lr_pe', ('out',))
def loftup_lr_pe(biases: Tensor, out: Tensor, col0: int, nimg: int, h: int, w: int) -> None:
    hip.loftup_lr_pe(biases, out, col0, nimg, h, w)


@_op('pp_scores', ('scores', 'labels', 'keep'))
def pp_scores(logits: Tensor, cls_threshold: float, temperature: float, scores: Tensor, labels: Tensor, keep: Tensor) -> None:
    hip.pp_scores(logits, cls_threshold, temperature, scores, labels, keep)


@_op('pp_argmax_logits', ('best_q', 'best_m', 'cnt_orig', 'cnt_mask'))
def pp_argmax_logits(logits: Tensor, scores: Tensor, keep: Tensor, Q: int, Hm: int, Wm: int, H: int, W: int, mask_threshold: float, best_q: Tensor,
                     best_m: Tensor, cnt_orig: Tensor, cnt_mask: Tensor) -> None:
    hip.pp_argmax_logits(logits, scores, keep, Q, Hm, Wm, H, W, mask_threshold, best_q, best_m, cnt_orig, cnt_mask)


@_op('pp_sigmoid', ('probs',))
def pp_sigmoid(logits: Tensor, keep: Tensor, probs: Tensor, Q: int, P: int) -> None:
    hip.pp_sigmoid(logits, keep, probs, Q, P)


@_op('pp_argmax', ('best_q', 'best_m', 'cnt_orig', 'cnt_mask'))
def pp_argmax(probs: Tensor, scores: Tensor, keep: Tensor, Q: int, Hm: int, Wm: int, H: int, W: int, mask_threshold: float, best_q: Tensor,
              best_m: Tensor, cnt_orig: Tensor, cnt_mask: Tensor) -> None:
    hip.pp_argmax(probs, scores, keep, Q, Hm, Wm, H, W, mask_threshold, best_q, best_m, cnt_orig, cnt_mask)


@_op('pp_select', ('cnt_orig', 'cnt_mask', 'keep_out', 'seg_id'))
def pp_select(keep: Tensor, cnt_orig: Tensor, cnt_mask: Tensor, Q: int, overlap_threshold: float, keep_out: Tensor, seg_id: Tensor) -> None:
    hip.pp_select(keep, cnt_orig, cnt_mask, Q, overlap_threshold, keep_out, seg_id)


@_op('pp_finalize', ('pan', 'conf'))
def pp_finalize(best_q: Tensor, best_m: Tensor, seg_id: Tensor, n: int, mask_threshold: float, void_confidence: float, pan: Tensor, conf: Tensor) -> None:
    hip.pp_finalize(best_q, best_m, seg_id, n, mask_threshold, void_confidence, pan, conf)


@_op('layernorm_batch', ('out_all',))
def layernorm_batch(x_all: Tensor, gamma_all: Tensor, beta_all: Tensor, out_all: Tensor, eps: float, rows: Optional[int] = None,
                    grp: Optional[List[int]] = None, add: Optional[Tensor] = None) -> None:
    """the 12 per-layer norm_y(h_l + feedback) of a MUSt3R memory append in one launch"""
    hip.layernorm_batch(x_all, gamma_all, beta_all, out_all, eps, rows=rows, grp=_t3(grp), add=add)


@_op('pointmap_activate', ('pts3d', 'pts3d_local', 'conf'))
def pointmap_activate(raw: Tensor, pts3d: Tensor, pts3d_local: Tensor, conf: Tensor, mode: int = 0) -> None:
    hip.pointmap_activate(raw, pts3d, pts3d_local, conf, mode)


@_op('focal_weiszfeld', ('focal',))
def focal_weiszfeld(pts3d_local: Tensor, pp: Tensor, focal: Tensor, H: int, W: int, iters: int = 10) -> None:
    hip.focal_weiszfeld(pts3d_local, pp, focal, H, W, iters)


@_op('rigid_moments', ('out',))
def rigid_moments(x: Tensor, y: Tensor, conf: Tensor, out: Tensor, weight_offset: float = -1.0) -> None:
    hip.rigid_moments(x, y, conf, out, weight_offset)


@_op('qubo_upsample', ('probs',))
def qubo_upsample(logits: Tensor, probs: Tensor, Q: int, hm: int, wm: int, H: int, W: int) -> None:
    hip.qubo_upsample(logits, probs, Q, hm, wm, H, W)


@_op('qubo_overlap', ('Wacc',))
def qubo_overlap(probs: Tensor, Q: int, P: int, Wacc: Tensor) -> None:
    hip.qubo_overlap(probs, Q, P, Wacc)


@_op('qubo_argmax', ('conf', 'inst'))
def qubo_argmax(probs: Tensor, sel: Tensor, P: int, conf: Tensor, inst: Tensor) -> None:
    hip.qubo_argmax(probs, sel, P, conf, inst)


def registered_ops():
    """names under torch.ops.panst3r_hip"""
    return list(_REGISTERED)


# ---------------------------------------------------------------------------------------------------- reference-signature shims
_ROPE_TABLES = {}


def rope_2d(tokens, positions, base, F0=1.0):
    """cuRoPE2D drop-in: `curope.rope_2d(tokens, positions, base, F0)` (README.md:67-71) -- IN PLACE on tokens [B, N, H, D] (16-bit,
    contiguous, cuda), positions [B, N, 2] integer (y, x).  First D/2 channels of each head rotate with y, the rest with x;
    angle = pos * F0 * base^(-2i/(D/2))."""
    B, N, H, D = tokens.shape
    if tokens.dtype not in hip.H16 or not tokens.is_contiguous():
        raise RuntimeError('rope_2d: tokens must be contiguous bf16 / f16 [B, N, H, D] on the GPU (got %s)' % tokens.dtype)
    pos = positions.reshape(B * N, 2).to(torch.int32).contiguous()
    npos = int(pos.max().item()) + 1
    key = (str(tokens.device), D, float(base), float(F0))
    tab = _ROPE_TABLES.get(key)
    if tab is None or tab.shape[0] < npos:
        n = max(npos, 64)
        half = D // 2
        inv = float(F0) / (float(base) ** (torch.arange(0, half, 2, dtype=torch.float32) / half))
        ang = torch.outer(torch.arange(n, dtype=torch.float32), inv)
        tab = _ROPE_TABLES[key] = torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().to(tokens.device)
    torch.ops.panst3r_hip.rope2d_(tokens.view(B * N, H * D), pos, tab, H, D)
    return tokens


class MultiheadAttention(nn.Module):
    """`nn.MultiheadAttention(embed_dim, num_heads)` as the reference's query decoder uses it (mask_transformer.py:314,372): same
    parameters (`in_proj_weight`, `in_proj_bias`, `out_proj.weight`, `out_proj.bias` -> state dicts interchange), same call
    `mha(query, key, value=..., attn_mask=..., key_padding_mask=None)` on seq-first [L, B, E] tensors, returns (out [L, B, E], None).
    attn_mask: bool [B*H, L, S], True = blocked, identical for the H heads of a batch element (the reference builds it with
    `.repeat(1, num_heads, 1, 1).flatten(0, 1)`, mask_transformer.py:272).  Head dim must be 64 or 96."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        self._packed = {}

    def _load_from_state_dict(self, *a, **k):
        self._packed = {}
        return super()._load_from_state_dict(*a, **k)

    def _pack(self, dev):
        key = (str(dev), adt())
        if key not in self._packed:
            E = self.embed_dim
            W = self.in_proj_weight.detach().to(device=dev, dtype=adt()).contiguous()
            self._packed[key] = (W[:E], W[E:2 * E], W[2 * E:], self.in_proj_bias.detach().float().to(dev).contiguous(),
                                 self.out_proj.weight.detach().to(device=dev, dtype=adt()).contiguous(), self.out_proj.bias.detach().float().to(dev).contiguous())
        return self._packed[key]

    @torch.no_grad()
    def forward(self, query, key, value, attn_mask=None, key_padding_mask=None, need_weights=False):
        if key_padding_mask is not None or need_weights:
            raise NotImplementedError('key_padding_mask / attention weights are not used on the inference path')
        L, B, E = query.shape
        S = key.shape[0]
        H, hd, dev = self.num_heads, E // self.num_heads, query.device
        wq, wk, wv, b, wo, bo = self._pack(dev)
        to16 = lambda t, n: t.transpose(0, 1).reshape(B * n, E).to(adt()).contiguous()          # [B*n, E] batch-major rows
        q16, k16, v16 = to16(query, L), to16(key, S), to16(value, S)
        Sp = ceil_to(S, 8)
        q = torch.empty(B * L, E, dtype=adt(), device=dev)
        k = torch.empty(B * S, E, dtype=adt(), device=dev)
        vt = torch.zeros(E, B * Sp + 8, dtype=adt(), device=dev)
        ops = torch.ops.panst3r_hip
        ops.gemm(q16, wq, q, bias=b[:E])
        ops.gemm(k16, wk, k, bias=b[E:2 * E])
        for bi in range(B):            # V^T per batch element (key-contiguous rows padded to 16 bytes)
            ops.gemm(v16[bi * S:(bi + 1) * S], wv, vt[:, bi * Sp:], bias=b[2 * E:], trans_out=True)
        mask = None
        if attn_mask is not None:
            m = attn_mask.view(B, H, L, S)
            if not bool((m == m[:, :1]).all()):
                raise NotImplementedError('attn_mask must be shared by the heads of a batch element (as mask_transformer.py:272 builds it)')
            Sm = ceil_to(S, 4)
            mask = torch.zeros(B, L, Sm, dtype=torch.uint8, device=dev)
            mask[:, :, :S] = m[:, 0].to(torch.uint8)
        o = torch.empty(B * L, E, dtype=adt(), device=dev)
        ldv = vt.stride(0)
        ops.attention(q, k, vt, o, B, H, L, S, hd, [L * E, hd, E], [S * E, hd, E], [Sp, hd * ldv, ldv], [L * E, hd, E],
                      mask=mask, mask_strides=[L * mask.shape[2], mask.shape[2]] if mask is not None else None)
        out = torch.empty(B * L, E, dtype=torch.float32, device=dev)
        ops.gemm(o, wo, out, bias=bo)
        return out.view(B, L, E).transpose(0, 1).contiguous(), None


def mask_einsum(mask_embed, mask_feats):
    """torch.einsum("bqc,bnchw->bnqhw", mask_embed, mask_feats) (mask_transformer.py:280): mask_embed [B,Q,C], mask_feats [B,N,C,H,W]
    (the reference's channel-first layout) -> fp32 [B,N,Q,H,W].  The pixel-major copy of the features made here is what the fused
    pipeline never materialises (its upscaler writes [H,W,C] directly): use `mask_einsum_pixel_major` when the features already are."""
    B, N, C, H, W = mask_feats.shape
    f = mask_feats.permute(0, 1, 3, 4, 2).reshape(B, N, H * W, C).to(adt()).contiguous()
    return mask_einsum_pixel_major(mask_embed, f).view(B, N, mask_embed.shape[1], H, W)


def mask_einsum_pixel_major(mask_embed, feats):
    """mask_embed [B,Q,C] (fp32 or 16-bit), feats 16-bit [B,N,P,C] pixel-major -> fp32 logits [B,N,Q,P]: one NT GEMM per view."""
    B, Q, C = mask_embed.shape
    N, P = feats.shape[1:3]
    Cp = ceil_to(C, 64)
    e = torch.zeros(B, Q, Cp, dtype=feats.dtype, device=feats.device)
    e[:, :, :C] = mask_embed.to(feats.dtype)
    if Cp != C:
        fp = torch.zeros(B, N, P, Cp, dtype=feats.dtype, device=feats.device)
        fp[..., :C] = feats
        feats = fp
    out = torch.empty(B, N, Q, P, dtype=torch.float32, device=feats.device)
    for b in range(B):
        for n in range(N):
            torch.ops.panst3r_hip.gemm(e[b], feats[b, n], out[b, n])
    return out
