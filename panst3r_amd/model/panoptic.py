"""PanSt3R-owned panoptic half on the HIP path.

Same class names, ctor kwargs and state-dict keys as the reference modules:
  InputMixer            model/input_mixer.py:9-29
  PixelShuffleUpscaler  model/upscalers/pixel_shuffle.py:9-59
  LoftUpUpscaler        model/upscalers/loftup.py:82-190
  MaskTransformer       model/mask_transformer.py:12-288
  TextEncoder           model/text_encoder.py:33-103 (fixed-vocabulary branch)
  PanopticDecoder       model/panoptic_decoder.py:16-77

MI355X-first data layout: every feature map is pixel/token-major ([view, pixel, channel], channel contiguous, bf16), so
  * F.pixel_shuffle is a store permutation in the producing GEMM (weight rows re-ordered once to [dy][dx][c]),
  * the 3x3 convs are implicit GEMMs over NHWC,
  * the query x pixel einsum "bqc,bnchw->bnqhw" (mask_transformer.py:280) is a plain NT GEMM
    mask_embed[Q,C] x mask_feats[P,C]^T whose fp32 [Q,P] output IS pred_masks[view],
  * the attention mask of the 6 intermediate decoder layers needs only the mean of the central 2x2 pixels of each 8x8
    block (== the 8x bilinear resize, mask_transformer.py:283-287): E . mean4(F) is one tiny GEMM per layer instead of
    a full-resolution einsum + resize; the never-consumed aux_outputs are not produced.
"""
import math
import os
import torch
import torch.nn as nn

from .. import hip
from .common import (HipModule, qscale, Packed, Layout, adt, empty, attn_out, mlp_hidden, x3, vit_block, pack_croco_block, pack_norm, f32, ParamLinear,
                     grid_pos, ceil_to, grow_table, Stream, fold_ln, ln_of)
from .params import BlockP, MlpP, CrossAttnP, MHAP

VIEW_CHUNK = 16     # views per upscaler pass (bounds the [rows, 22528] / [P, 384] workspaces)
BALANCED_CHUNKS = os.environ.get('PST_BALANCED_CHUNKS', '1') != '0'      # A/B switch (0: full passes + a remainder, as before round 4)


def view_chunks(V):
    """(first view, views) of the upscaler passes over V same-shape views: as few passes as VIEW_CHUNK allows, of BALANCED sizes - the scene's 34 views that
    are not keyframes go as 12 + 11 + 11, not 16 + 16 + 2 (a 2-view pass runs the same launches at an eighth of the rows: tile quantisation and launch
    floors).  Every per-view result is independent of the pass it is computed in."""
    n = (V + VIEW_CHUNK - 1) // VIEW_CHUNK
    v0 = 0
    for i in range(n):
        c = (V // n + (1 if i < V % n else 0)) if BALANCED_CHUNKS else min(VIEW_CHUNK, V - v0)
        yield v0, c
        v0 += c


# =========================================================================================== InputMixer
class InputMixer(HipModule):
    def __init__(self, img_size, patch_size, in_dim, hidden_dim, num_heads=12, num_layers=3, ff_dim_mult=4):
        super().__init__()
        self.hidden_dim, self.num_heads = hidden_dim, num_heads
        self.in_proj = ParamLinear(in_dim, hidden_dim)
        self.mixer_blk = nn.ModuleList([BlockP(hidden_dim, ff_dim_mult, True, 1e-5) for _ in range(num_layers)])
        self.mixer_norm = nn.LayerNorm(hidden_dim)

    def _pack(self, device):
        return dict(inp=Packed(self.in_proj.weight, self.in_proj.bias, device),
                    blocks=[pack_croco_block(b, device) for b in self.mixer_blk], norm=pack_norm(self.mixer_norm, device), rope={})

    @torch.no_grad()
    def mix_tokens(self, cat, V, h, w, out):
        """cat bf16 [V*T, in_dim] -> LN'd mixer tokens written to out[:, :hidden] (bf16, row-major view)."""
        dev = cat.device
        pk = self.packed(dev)
        D, H = self.hidden_dim, self.num_heads
        lay = Layout(V, h * w)
        x = torch.zeros(lay.rows, D, dtype=torch.float32, device=dev)
        hip.gemm(cat, pk['inp'].w, x, bias=pk['inp'].b, grp=lay.grp)
        pos = grid_pos(V, h, w, lay.Tp, 0, dev)
        rope = grow_table(pk['rope'], max(h, w), lambda m: hip.rope_table(m, D // H, 100.0, dev))
        st = Stream(x).refresh()
        for bw in pk['blocks']:
            vit_block(st, bw, lay, H, D // H, pos, rope)
        hip.layernorm(x, pk['norm'][0], pk['norm'][1], out[:, :D], pk['norm'][2], rows=V * lay.T, grp=lay.grp)
        return out

    def forward(self, x, pos):
        V, T, _ = x.shape
        h = int(pos[0, :, 0].max()) + 1
        out = torch.empty(V * T, self.hidden_dim, dtype=adt(), device=x.device)
        self.mix_tokens(x.reshape(V * T, -1).to(adt()).contiguous(), V, h, T // h, out)
        return out.float().reshape(V, T, -1)


# =========================================================================================== PixelShuffle upscaler (v1)
def _ps_perm(c, p=2):
    """row permutation that turns F.pixel_shuffle's channel order c*p*p + dy*p + dx into [dy][dx][c]."""
    return torch.arange(c * p * p).reshape(c, p, p).permute(1, 2, 0).reshape(-1)


class PixelShuffleUpscaler(HipModule):
    def __init__(self, input_dim, patch_size=16, hidden_dim_factor=4, fp_dim=(768, 512, 384, 256), fp_activation=nn.GELU, **kw):
        super().__init__()
        assert fp_activation is nn.GELU
        self.patch_size, self.fp_dim, self.input_dim = patch_size, list(fp_dim), input_dim
        f = hidden_dim_factor
        self.proj_8 = MlpP(input_dim, int(f * input_dim), fp_dim[1] * 4)
        self.proj_4 = MlpP(fp_dim[1], int(f * fp_dim[1]), fp_dim[2] * 4)
        self.proj_2 = MlpP(fp_dim[2], int(f * fp_dim[2]), fp_dim[3] * 4)
        self.proj_16 = MlpP(input_dim, int(f * input_dim), fp_dim[0])
        self.mask_dim = fp_dim[3]
        self.fpn_dim = fp_dim[0]

    def _pack(self, device):
        P = lambda lin, perm=None: Packed(lin.weight, lin.bias, device, row_perm=perm)
        fc1 = Packed(torch.cat([self.proj_16.fc1.weight, self.proj_8.fc1.weight]),
                     torch.cat([self.proj_16.fc1.bias, self.proj_8.fc1.bias]), device)     # one GEMM, N = 2*hidden
        return dict(fc1=fc1, hid=self.proj_16.fc1.weight.shape[0], p16=P(self.proj_16.fc2),
                    p8=P(self.proj_8.fc2, _ps_perm(self.fp_dim[1])),
                    p4a=P(self.proj_4.fc1), p4b=P(self.proj_4.fc2, _ps_perm(self.fp_dim[2])),
                    p2a=P(self.proj_2.fc1), p2b=P(self.proj_2.fc2, _ps_perm(self.fp_dim[3])))

    @torch.no_grad()
    def upscale_tokens(self, cat, imgs, V, h, w, fpn_out, mask_out):
        """cat bf16 [V*T, input_dim] -> fpn_out bf16 [V*T, 768], mask_out bf16 [V, 8h, 8w, 256] (pixel-major)."""
        dev = cat.device
        pk = self.packed(dev)
        T = h * w
        d1, d2, d3 = self.fp_dim[1], self.fp_dim[2], self.fp_dim[3]
        for v0, n in view_chunks(V):
            a = cat[v0 * T:(v0 + n) * T]
            hid = empty(n * T, pk['fc1'].n, adt(), dev)
            hip.gemm(a, pk['fc1'].w, hid, bias=pk['fc1'].b, act='gelu')
            hip.gemm(hid[:, :pk['hid']], pk['p16'].w, fpn_out[v0 * T:(v0 + n) * T], bias=pk['p16'].b)
            f8 = empty(n * 4 * T, d1, adt(), dev)
            hip.gemm(hid[:, pk['hid']:], pk['p8'].w, f8, bias=pk['p8'].b, ps=(2, d1, h, w))
            del hid
            h4 = empty(n * 4 * T, pk['p4a'].n, adt(), dev)
            hip.gemm(f8, pk['p4a'].w, h4, bias=pk['p4a'].b, act='gelu')
            f4 = empty(n * 16 * T, d2, adt(), dev)
            hip.gemm(h4, pk['p4b'].w, f4, bias=pk['p4b'].b, ps=(2, d2, 2 * h, 2 * w))
            del h4, f8
            h2 = empty(n * 16 * T, pk['p2a'].n, adt(), dev)
            hip.gemm(f4, pk['p2a'].w, h2, bias=pk['p2a'].b, act='gelu')
            hip.gemm(h2, pk['p2b'].w, mask_out[v0:v0 + n], bias=pk['p2b'].b, ps=(2, d3, 4 * h, 4 * w))
            del h2, f4
        return fpn_out, mask_out

    def forward(self, feats, img_shape):
        """Reference signature: feats=(tokens [b,T,C], ...), img_shape (H,W) -> ([f16 [b,768,h,w]], mask_feats [b,256,H/2,W/2])."""
        x = feats[0]
        V, T, _ = x.shape
        H, W = img_shape
        h, w = H // self.patch_size, W // self.patch_size
        fpn = torch.empty(V * T, self.fpn_dim, dtype=adt(), device=x.device)
        mf = torch.empty(V, 8 * h, 8 * w, self.mask_dim, dtype=adt(), device=x.device)
        self.upscale_tokens(x.reshape(V * T, -1).to(adt()).contiguous(), None, V, h, w, fpn, mf)
        return [fpn.float().reshape(V, h, w, -1).permute(0, 3, 1, 2)], mf.float().permute(0, 3, 1, 2)


# =========================================================================================== LoftUp upscaler (v2)
class _Featurizer(nn.Module):
    def __init__(self, dm, nf):
        super().__init__()
        self.biases = nn.Parameter(torch.randn(2, dm, nf))


class _CrossOnlyP(nn.Module):
    """CrossonlyDecoderBlock parameters (model/blocks.py:9-35): cross_attn (no qkv bias), norm2, norm3, mlp, norm_y."""

    def __init__(self, dim, mlp_ratio):
        super().__init__()
        self.cross_attn = CrossAttnP(dim, qkv_bias=False)
        self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.mlp = MlpP(dim, int(dim * mlp_ratio))
        self.norm_y = nn.LayerNorm(dim)


class LoftUpUpscaler(HipModule):
    def __init__(self, input_dim, dim, output_stride=2, patch_size=16, color_feats=True, n_freqs=20, num_heads=4, num_layers=2,
                 lr_pe_type='sine'):
        super().__init__()
        assert lr_pe_type == 'sine' and color_feats and output_stride == 2, 'released LoftUp configuration only'
        self.input_dim, self.dim, self.patch_size, self.n_freqs, self.num_heads = input_dim, dim, patch_size, n_freqs, num_heads
        self.patch_embed = nn.Conv2d(input_dim, input_dim, kernel_size=1)
        start = 5 * n_freqs * 2 + 3
        self.start_dim = start
        self.lr_pe = _Featurizer(2, 5)
        self.lr_input_proj = nn.Sequential(ParamLinear(input_dim + 20, dim), nn.LayerNorm(dim))
        self.fourier_feat = nn.Sequential(nn.Identity(), _Featurizer(5, n_freqs))
        self.first_conv = nn.Sequential(nn.GroupNorm(1, start), nn.Conv2d(start, dim, 3, padding=1), nn.GroupNorm(8, dim), nn.ReLU(),
                                        nn.Conv2d(dim, dim, 3, padding=1), nn.GroupNorm(8, dim), nn.ReLU())
        self.ca_transformer_blocks = nn.ModuleList([_CrossOnlyP(dim, 1) for _ in range(num_layers)])
        self.ca_transformer_norm = nn.LayerNorm(dim)
        self.mask_dim, self.fpn_dim = dim, input_dim

    def _pack(self, device):
        def conv_w(conv, cpad):      # [Cout, Cin, 3, 3] -> [Cout, 9, cpad] tap-major / channel-minor, zero padded
            wt = conv.weight.detach().float().permute(0, 2, 3, 1)
            out = torch.zeros(wt.shape[0], 3, 3, cpad)
            out[..., :wt.shape[-1]] = wt
            return Packed(out.reshape(wt.shape[0], -1), conv.bias, device, taps=9)
        c0 = ceil_to(self.start_dim, 64)
        fc = self.first_conv
        blocks = []
        for b in self.ca_transformer_blocks:
            c = b.cross_attn
            blocks.append(dict(norm2=pack_norm(b.norm2, device), norm3=pack_norm(b.norm3, device), norm_y=pack_norm(b.norm_y, device),
                               q=fold_ln(c.projq.weight, c.projq.bias, b.norm2, device), k=Packed(c.projk.weight, c.projk.bias, device),
                               v=Packed(c.projv.weight, c.projv.bias, device), proj=Packed(c.proj.weight, c.proj.bias, device),
                               fc1=fold_ln(b.mlp.fc1.weight, b.mlp.fc1.bias, b.norm3, device), fc2=Packed(b.mlp.fc2.weight, b.mlp.fc2.bias, device)))
        gn = lambda g: (f32(g.weight, device), f32(g.bias, device), float(g.eps))
        return dict(pe=Packed(self.patch_embed.weight, self.patch_embed.bias, device), c0=c0,
                    lr_bias=f32(self.lr_pe.biases, device), ff_bias=f32(self.fourier_feat[1].biases, device),
                    lr_proj=Packed(self.lr_input_proj[0].weight, self.lr_input_proj[0].bias, device),
                    lr_norm=pack_norm(self.lr_input_proj[1], device),
                    gn0=gn(fc[0]), conv1=conv_w(fc[1], c0), gn1=gn(fc[2]), conv2=conv_w(fc[4], self.dim), gn2=gn(fc[5]),
                    blocks=blocks, norm=pack_norm(self.ca_transformer_norm, device))

    def lr_width(self):
        return ceil_to(self.input_dim + 20, 64)

    @torch.no_grad()
    def guidance_tokens(self, imgs, h, w, mm=None):
        """Guidance branch (loftup.py:154-156,117-130): Fourier features -> GN(1) -> conv3x3 -> GN(8)+ReLU -> conv3x3 -> GN(8)+ReLU.
        imgs fp32 [V,3,H,W] -> bf16 [V*P, dim] pixel-major, P = H/2 * W/2.  It depends on the IMAGES only (not on the memory, the
        decoder or the mixer), so the scene runner computes it with the other memory-independent work (SceneRunner._encode_rest).
        A tall token grid (h > w) takes the image transposed (loftup.py:147-149).
        mm: None = MinMaxScaler per view (the demo's max_bs=1); else fp32 [V,3,2] (min, max) per (view, channel) pooled over the chunk of views the
        reference would have scaled together (loftup.py:14-19: min / max over the batch it is handed; `minmax_tables`)."""
        dev = imgs.device
        pk = self.packed(dev)
        C = self.dim
        if h > w:
            imgs = imgs.transpose(2, 3).contiguous()
        V = imgs.shape[0]
        H2, W2 = imgs.shape[2] // 2, imgs.shape[3] // 2
        P = H2 * W2
        out = empty(V * P, C, adt(), dev)
        for v0, n in view_chunks(V):
            # Fourier features + GroupNorm(1) in two recomputing passes straight to bf16: no [n, P, 203] fp32 feature buffer
            # (pst_loftup_guidance + pst_groupnorm_apply did the same through a 639 MB round trip: 1076 -> 254 us per 16 views)
            st0 = hip.stats_buffer(n, 1, dev)
            g0 = empty(n * P, pk['c0'], adt(), dev)
            scratch = torch.empty(n * (3 * P + 6) + 16, dtype=torch.float32, device=dev)
            hip.loftup_guidance_gn(imgs[v0:v0 + n].contiguous(), pk['ff_bias'], pk['gn0'][0], pk['gn0'][1], pk['gn0'][2], scratch, st0, g0,
                                   self.n_freqs, mm=None if mm is None else mm[v0:v0 + n].contiguous())
            del scratch
            c1 = empty(n * P, C, adt(), dev)
            hip.gemm(g0, pk['conv1'].w, c1, bias=pk['conv1'].b, conv=(pk['c0'], H2, W2))
            st = hip.stats_buffer(n, 8, dev)
            hip.groupnorm_stats(c1, st, n, P, C, 8)
            g1 = out[v0 * P:(v0 + n) * P]
            if x3():     # the first GroupNorm's result only feeds conv2: written as that GEMM's split A operand (rows [hi | hi | lo]: no fp32 round trip)
                g1s = torch.empty(n * P, 3 * C, dtype=hip.X3_FMT, device=dev)
                hip.groupnorm_apply(c1, st, pk['gn1'][0], pk['gn1'][1], g1s, n, P, C, 8, pk['gn1'][2], True, split=True)
                hip.gemm(g1s, pk['conv2'].w, c1, bias=pk['conv2'].b, conv=(3 * C, H2, W2))
                del g1s
            else:
                hip.groupnorm_apply(c1, st, pk['gn1'][0], pk['gn1'][1], g1, n, P, C, 8, pk['gn1'][2], True)
                hip.gemm(g1, pk['conv2'].w, c1, bias=pk['conv2'].b, conv=(C, H2, W2))
            hip.groupnorm_stats(c1, st, n, P, C, 8)
            hip.groupnorm_apply(c1, st, pk['gn2'][0], pk['gn2'][1], g1, n, P, C, 8, pk['gn2'][2], True)
            del g0, c1
        return out

    @torch.no_grad()
    def upscale_tokens(self, lr, imgs, V, h, w, fpn_out, mask_out, guidance=None, mm=None):
        """lr bf16 [V*T, lr_width()] with the mixer tokens in columns [0, input_dim) (the rest is filled here);
        imgs fp32 [V,3,H,W] -> fpn_out bf16 [V*T, input_dim] (h x w raster), mask_out bf16 [V, H/2, W/2, dim].
        A tall token grid (h > w) takes the guidance image transposed (loftup.py:147-149): the mask features then come
        out landscape-shaped, mask_out bf16 [V, W/2, H/2, dim] (the low-res tokens only enter through cross-attention,
        which does not care about their raster).  `guidance`: guidance_tokens(imgs, h, w) when the caller computed it earlier
        (it is consumed: the two blocks update it in place as their residual stream)."""
        dev = lr.device
        pk = self.packed(dev)
        T, D, C, Hh = h * w, self.input_dim, self.dim, self.num_heads
        hd = C // Hh
        H2, W2 = (imgs.shape[3] // 2, imgs.shape[2] // 2) if h > w else (imgs.shape[2] // 2, imgs.shape[3] // 2)
        assert tuple(mask_out.shape) == (V, H2, W2, C), (tuple(mask_out.shape), (V, H2, W2, C))
        P = H2 * W2
        if guidance is None:
            guidance = self.guidance_tokens(imgs, h, w, mm=mm)
        assert tuple(guidance.shape) == (V * P, C)
        hip.gemm(lr[:, :D], pk['pe'].w, fpn_out, bias=pk['pe'].b)
        lr[:, D:].zero_()
        hip.loftup_lr_pe(pk['lr_bias'], lr, D, V, h, w)
        lay = Layout(V, T)
        kv = torch.zeros(lay.rows, C, dtype=torch.float32, device=dev)
        hip.gemm(lr, pk['lr_proj'].w, kv, bias=pk['lr_proj'].b, grp=lay.grp)
        kvn = empty(lay.rows, C, torch.float32, dev)
        hip.layernorm(kv, pk['lr_norm'][0], pk['lr_norm'][1], kvn, pk['lr_norm'][2])
        vts = {}            # V^T scratch per chunk size, zeroed once (the transposed store writes the real columns; the 8 pad columns stay 0)
        for v0, n in view_chunks(V):
            # ---- 2 x cross-only blocks: 49k queries per view attend to the view's T low-res tokens (hd 96).
            # The residual stream of these two blocks is kept in bf16: they are HBM-bound over P x C elements per view
            # (fp32 would double the read-modify-write traffic of both residual GEMMs and of every LayerNorm).
            x = guidance[v0 * P:(v0 + n) * P]
            q, o = empty(n * P, C, adt(), dev), attn_out(n * P, C, dev)
            ldo = o.stride(0)
            # LayerNorm fold on the 16-bit stream: norm2 / norm3 live in projq / fc1; in f16 their row statistics come out of the epilogues
            # of the two residual GEMMs (no stand-alone pass over the [n*P, C] tensor: 4 of the 16 passes per chunk gone)
            s = Stream(x).refresh()
            rows0, rows1 = v0 * lay.Tp, (v0 + n) * lay.Tp
            for bw in pk['blocks']:
                y = empty(rows1 - rows0, C, adt(), dev)
                hip.layernorm(kvn[rows0:rows1], bw['norm_y'][0], bw['norm_y'][1], y, bw['norm_y'][2])
                kk = empty(rows1 - rows0, C, adt(), dev)
                hip.gemm(y, bw['k'].w, kk, bias=bw['k'].b)
                if rows1 - rows0 not in vts:
                    vts[rows1 - rows0] = torch.zeros(C, rows1 - rows0 + 8, dtype=adt(), device=dev)
                vt = vts[rows1 - rows0]
                hip.gemm(y, bw['v'].w, vt, bias=bw['v'].b, trans_out=True)
                a, ln = s.operand(bw['q'])
                hip.gemm(a, bw['q'].w, q, bias=bw['q'].b, gamma=qscale(C, C, hd, dev), ln=ln)
                ldv = vt.stride(0)
                hip.attention(q, kk, vt, o, n, Hh, P, T, hd, q_strides=(P * C, hd, C), k_strides=(lay.Tp * C, hd, C),
                              v_strides=(lay.Tp, hd * ldv, ldv), o_strides=(P * ldo, hd, ldo), prescaled=True)
                s.residual(o, bw['proj'])
                if x3():            # (the split-operand fp32 mode: the hidden activation leaves fc1 as fc2's split A operand)
                    a, w_, hmid, kw = mlp_hidden(s, bw['fc1'], n * P, dev)
                    hip.gemm(a, w_, hmid, **kw)
                    s.residual(hmid, bw['fc2'])
                    del hmid
                else:
                    a, ln = s.operand(bw['fc1'])
                    hip.gemm(a, bw['fc1'].w, q, bias=bw['fc1'].b, act='gelu', ln=ln)
                    s.residual(q, bw['fc2'])
            hip.layernorm(x, pk['norm'][0], pk['norm'][1], mask_out[v0:v0 + n].view(n * P, C), pk['norm'][2])
            del x, q, o, s
        return fpn_out, mask_out

    def forward(self, inputs, img_shape):
        tok, img = inputs
        V, T, _ = tok.shape
        H, W = img_shape
        h, w = H // self.patch_size, W // self.patch_size
        lr = torch.zeros(V * T, self.lr_width(), dtype=adt(), device=tok.device)
        lr[:, :self.input_dim] = tok.reshape(V * T, -1).to(adt())
        fpn = torch.empty(V * T, self.fpn_dim, dtype=adt(), device=tok.device)
        mf = torch.empty(V, min(H, W) // 2, max(H, W) // 2, self.dim, dtype=adt(), device=tok.device)   # loftup.py:147-149,184
        self.upscale_tokens(lr, img.float(), V, h, w, fpn, mf)
        return [fpn.float().reshape(V, h, w, -1).permute(0, 3, 1, 2)], mf.float().permute(0, 3, 1, 2)


# =========================================================================================== MaskTransformer
def sine_pe(h, w, dim, temperature=10000.0):
    """PositionEmbeddingSine(dim/2, normalize=True) of an h x w grid as [h*w, dim] (mask_transformer.py:488-527):
    channels [0, dim/2) encode y, [dim/2, dim) encode x; 1-based cumulative index / (size + 1e-6) * 2*pi."""
    npf = dim // 2
    ye = torch.arange(1, h + 1, dtype=torch.float32).view(h, 1).expand(h, w) / (h + 1e-6) * (2 * math.pi)
    xe = torch.arange(1, w + 1, dtype=torch.float32).view(1, w).expand(h, w) / (w + 1e-6) * (2 * math.pi)
    i = torch.arange(npf, dtype=torch.float32)
    div = temperature ** (2 * torch.div(i, 2, rounding_mode='floor') / npf)

    def enc(e):
        p = e[..., None] / div
        return torch.stack([p[..., 0::2].sin(), p[..., 1::2].cos()], dim=3).flatten(2)
    return torch.cat([enc(ye), enc(xe)], dim=2).reshape(h * w, dim)


class _SelfLayerP(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.self_attn, self.norm = MHAP(d), nn.LayerNorm(d)


class _CrossLayerP(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.multihead_attn, self.norm = MHAP(d), nn.LayerNorm(d)


class _FFNP(nn.Module):
    def __init__(self, d, ff):
        super().__init__()
        self.linear1, self.linear2, self.norm = ParamLinear(d, ff), ParamLinear(ff, d), nn.LayerNorm(d)


class _MLP3P(nn.Module):
    def __init__(self, i, h, o, n):
        super().__init__()
        dims = [i] + [h] * (n - 1) + [o]
        self.layers = nn.ModuleList(ParamLinear(a, b) for a, b in zip(dims[:-1], dims[1:]))


class HeadState:
    """Per-scene products of the frozen queries: class logits and the mask embedding (reused by every rendered view)."""
    __slots__ = ('logits', 'embed')


class MaskTransformer(HipModule):
    def __init__(self, in_dim, hidden_dim, ff_dim, mask_dim, num_queries, num_heads, dec_layers, lang_dim=768,
                 num_feature_levels=1, landscape_only=False, **kw):
        super().__init__()
        in_dim = [in_dim] if isinstance(in_dim, int) else list(in_dim)
        assert num_feature_levels == 1 and in_dim[0] == hidden_dim
        self.two_stage = bool(kw.get('two_stage', False))
        if mask_dim % 64 or (hidden_dim // num_heads) not in (64, 96):
            raise NotImplementedError('HIP MaskTransformer: mask_dim %% 64 == 0 and head dim 64/96 (got %d, %d)'
                                      % (mask_dim, hidden_dim // num_heads))
        self.hidden_dim, self.mask_dim, self.num_heads, self.num_layers = hidden_dim, mask_dim, num_heads, dec_layers
        self.num_queries, self.landscape_only = num_queries, landscape_only
        self.self_attn_layers = nn.ModuleList(_SelfLayerP(hidden_dim) for _ in range(dec_layers))
        self.cross_attn_layers = nn.ModuleList(_CrossLayerP(hidden_dim) for _ in range(dec_layers))
        self.ffn_layers = nn.ModuleList(_FFNP(hidden_dim, ff_dim) for _ in range(dec_layers))
        self.decoder_norm = nn.LayerNorm(hidden_dim)
        if not self.two_stage:             # learnt queries (mask_transformer.py:61-65); two_stage selects them from the keyframe tokens (:85-104)
            self.query_feat = nn.Embedding(num_queries, hidden_dim)
            self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.level_embed = nn.Embedding(1, hidden_dim)
        self.input_proj = nn.ModuleList([nn.Sequential()])
        self.lang_embed = ParamLinear(hidden_dim, lang_dim)
        self.cls_logit_scale = nn.Parameter(torch.ones([]))
        self.mask_embed = _MLP3P(hidden_dim, hidden_dim, mask_dim, 3)

    def _pack(self, device):
        d = self.hidden_dim

        def mha(m):
            w = Packed(m.in_proj_weight, m.in_proj_bias, device)
            return dict(q=w.rows(0, d), k=w.rows(d, 2 * d), v=w.rows(2 * d, 3 * d), qk=w.rows(0, 2 * d),
                        o=Packed(m.out_proj.weight, m.out_proj.bias, device))
        layers = []
        for i in range(self.num_layers):
            ca, sa, ff = self.cross_attn_layers[i], self.self_attn_layers[i], self.ffn_layers[i]
            layers.append(dict(ca=mha(ca.multihead_attn), ca_norm=pack_norm(ca.norm, device), sa=mha(sa.self_attn),
                               sa_norm=pack_norm(sa.norm, device), l1=Packed(ff.linear1.weight, ff.linear1.bias, device),
                               l2=Packed(ff.linear2.weight, ff.linear2.bias, device), ff_norm=pack_norm(ff.norm, device)))
        return dict(layers=layers, dn=pack_norm(self.decoder_norm, device), qf=None if self.two_stage else f32(self.query_feat.weight, device),
                    qe=None if self.two_stage else f32(self.query_embed.weight, device), lvl=f32(self.level_embed.weight, device),
                    lang=Packed(self.lang_embed.weight, self.lang_embed.bias, device),
                    me=[Packed(l.weight, l.bias, device) for l in self.mask_embed.layers],
                    # the same MLP for split-precision evaluation: weights [W_hi | W_lo | W_hi] bf16, fp32 bias
                    me3=None if adt() == torch.float32 else
                    [(hip.pack_split3(l.weight, adt()).to(device), f32(l.bias, device)) for l in self.mask_embed.layers],
                    scale=float(self.cls_logit_scale.detach().exp()), pe={})

    def _pe(self, pk, h, w, portrait, device):
        key = (h, w, bool(portrait))
        if key not in pk['pe']:
            # portrait views use the PE of the transposed grid, flattened in ITS raster order (mask_transformer.py:106-119)
            pe = sine_pe(w, h, self.hidden_dim) if portrait else sine_pe(h, w, self.hidden_dim)
            pk['pe'][key] = pe.to(device).contiguous()
        return pk['pe'][key]

    # ---- prediction heads
    def _embed(self, pk, out):
        dev = out.device
        Q, d = out.shape
        dn = empty(Q, d, adt(), dev)
        hip.layernorm(out, pk['dn'][0], pk['dn'][1], dn, pk['dn'][2])
        # mask_embed MLP in split precision (x = x_hi + x_lo, W = W_hi + W_lo, fp32 between the layers): its 200 x C result is one
        # factor of the ill-conditioned query x pixel product, where an embedding error of 8e-3 shows up as 1.9e-2 on the mask logits
        # (tests/diag/parity_maskhead.py).  The three GEMMs are tiny, so the 3x longer K costs nothing.
        a = empty(Q, d, torch.float32, dev)
        hip.layernorm(out, pk['dn'][0], pk['dn'][1], a, pk['dn'][2])
        if pk['me3'] is None:                   # fp32 operands (amp=False): the plain MLP already carries 24 mantissa bits
            for j, w in enumerate(pk['me']):
                b = empty(Q, w.n, torch.float32, dev)
                hip.gemm(a, w.w, b, bias=w.b, act=None if j == len(pk['me']) - 1 else 'relu')
                a = b
            return dn, a
        for j, (w3, b3) in enumerate(pk['me3']):
            a3 = empty(Q, w3.shape[1], adt(), dev)
            hip.split3(a, a3)
            b = empty(Q, w3.shape[0], torch.float32, dev)
            hip.gemm(a3, w3, b, bias=b3, act=None if j == len(pk['me3']) - 1 else 'relu')
            a = b
        emb = empty(Q, a.shape[1], adt(), dev)
        hip.add_cast(a, emb)
        return dn, emb

    def _class_logits(self, pk, dn, cls_bf16):
        dev = dn.device
        Q = dn.shape[0]
        lang = empty(Q, pk['lang'].n, torch.float32, dev)
        hip.gemm(dn, pk['lang'].w, lang, bias=pk['lang'].b)
        ln = torch.zeros(Q, cls_bf16.shape[1], dtype=adt(), device=dev)
        hip.l2norm_rows(lang, ln[:, :lang.shape[1]], 1e-7)
        # the GEMM wants N % 4 == 0: class counts like 133 (COCO panoptic) or 101 (100 classes + the softmax mode's "no object" row) run with up to three zero
        # rows behind the real ones - already there in the storage of the matrices this module hands out (class_rows / normalized_bf16), else appended here
        n = cls_bf16.shape[0]
        n4 = ceil_to(n, 4)
        w = cls_bf16
        if n4 != n:
            room = cls_bf16.untyped_storage().nbytes() // cls_bf16.element_size() - cls_bf16.storage_offset()
            if cls_bf16.stride(1) == 1 and room >= (n4 - 1) * cls_bf16.stride(0) + cls_bf16.shape[1] and getattr(cls_bf16, '_pst_zero_tail', False):
                w = cls_bf16.as_strided((n4, cls_bf16.shape[1]), cls_bf16.stride())
            else:
                w = torch.zeros(n4, cls_bf16.shape[1], dtype=cls_bf16.dtype, device=dev)
                w[:n].copy_(cls_bf16)
        logits = empty(Q, n4, torch.float32, dev)
        gams = pk.setdefault('gam', {})            # keyed by class count, entries never replaced (graph-captured addresses)
        if n4 not in gams:
            gams[n4] = torch.full((n4,), pk['scale'], dtype=torch.float32, device=dev)
        hip.gemm(ln, w, logits, gamma=gams[n4])
        return logits if n4 == n else logits[:, :n].contiguous()

    def _select_queries(self, pk, fpn, grids, portrait, cls_bf16):
        """two_stage (mask_transformer.py:85-104): decoder_norm -> lang_embed -> unit norm -> class logits of EVERY keyframe token; the num_queries tokens
        with the largest best-class logit, in descending order, are the initial queries (their fp32 `fpn + level_embed` rows) and their sine encodings
        the query positions.  The logits run through the same kernels as the class head; the top-k and the two row gathers are ATen ops."""
        dev = fpn.device
        NK, d = fpn.shape
        if NK < self.num_queries:
            raise ValueError('two_stage: %d keyframe tokens cannot supply %d queries (torch.topk fails the same way in the reference)' % (NK, self.num_queries))
        src32 = empty(NK, d, torch.float32, dev)
        hip.add_cast(fpn, src32, b=pk['lvl'], b_mod=1)
        pos32 = torch.cat([self._pe(pk, h, w, pt, dev) for (h, w), pt in zip(grids, portrait)])
        dn = empty(NK, d, adt(), dev)
        hip.layernorm(src32, pk['dn'][0], pk['dn'][1], dn, pk['dn'][2])
        best = self._class_logits(pk, dn, cls_bf16).max(-1)[0]
        idx = torch.topk(best, self.num_queries)[1]
        return src32.index_select(0, idx), pos32.index_select(0, idx), idx

    @torch.no_grad()
    def head_state(self, out_queries, cls_bf16):
        """decoder_norm -> class logits + mask embedding for a fixed set of queries (mask_transformer.py:222-230);
        computed once per scene, the reference recomputes it per rendered chunk (panoptic_decoder.py:71)."""
        pk = self.packed(out_queries.device)
        hs = HeadState()
        dn, hs.embed = self._embed(pk, out_queries)
        hs.logits = self._class_logits(pk, dn, cls_bf16)
        return hs

    @torch.no_grad()
    def masks_for(self, embed, mask_feats, out=None):
        """pred_masks of one view: embed bf16 [Q,C] x mask_feats bf16 [Hm,Wm,C] -> fp32 [Q,Hm,Wm]."""
        Hm, Wm, C = mask_feats.shape
        if out is None:
            out = torch.empty(embed.shape[0], Hm, Wm, dtype=torch.float32, device=embed.device)
        hip.gemm(embed, mask_feats.view(Hm * Wm, C), out.view(embed.shape[0], Hm * Wm))
        return out

    @torch.no_grad()
    def masks_for_group(self, embed, mask_feats):
        """pred_masks of ALL views of a shape group: embed 16-bit [Q,C] x mask_feats 16-bit [n,Hm,Wm,C] -> fp32 [n,Q,Hm,Wm].  One launch of the
        streaming mask-head kernel (pst_mask_head: the query matrix stays in registers, the features are read once) where it applies
        (C in {256, 384}: the released configurations), else one GEMM per view; the two are bit-identical."""
        n, Hm, Wm, C = mask_feats.shape
        Q = embed.shape[0]
        out = torch.empty(n, Q, Hm, Wm, dtype=torch.float32, device=embed.device)
        if embed.dtype != torch.float32 and hip.mask_head_supported(Q, Hm * Wm, C) and mask_feats.is_contiguous():
            hip.mask_head(embed, mask_feats, out)
        else:
            for i in range(n):
                self.masks_for(embed, mask_feats[i], out[i])
        return out

    @torch.no_grad()
    def attn_feats(self, mask_feats, grid=None):
        """Mask features bilinearly resized to the key grid: bf16 [n, Hm, Wm, C] -> [n*T, C] (the only part of the
        full-resolution masks the 6 intermediate decoder layers look at, mask_transformer.py:283-287; the resize is
        linear, so resize(E.F) == E.resize(F)).  The usual 8x case is the mean of the central 2x2 pixels of each 8x8
        block; `grid` = (rows, cols) of the FPN level when it differs (portrait views of the LoftUp variant: native
        masks against the transposed key grid, utils.py:47-49)."""
        n, Hm, Wm, C = mask_feats.shape
        gh, gw = grid if grid is not None else (Hm // 8, Wm // 8)
        fm = torch.empty(n * gh * gw, C, dtype=adt(), device=mask_feats.device)
        if (gh * 8, gw * 8) == (Hm, Wm):
            hip.mean4(mask_feats, fm, n, Hm, Wm, C)
        else:
            hip.resize_bilinear(mask_feats, fm, n, Hm, Wm, gh, gw, C)
        return fm

    _instr = None          # (forced attention-mask bits per layer | None, list collecting the bits | None): set ONLY inside instrument()

    def instrument(self, forced=None, log=None):
        """Parity instrumentation of decode_tokens (tests / bench.py, never the product path): `forced` = per-layer attention-mask bits
        to use instead of the thresholded logits (isolates arithmetic from the hard threshold at logit 0), `log` = a list that receives
        the bits of every layer.  A context manager: reset on exit even when the body raises, refused under graph capture."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            prev, self._instr = self._instr, (forced, log)
            try:
                yield self
            finally:
                self._instr = prev
        return ctx()

    @torch.no_grad()
    def decode_tokens(self, fpn, fm, grids, cls_bf16, portrait=None):
        """Query decoding on the keyframes.
        fpn bf16 [NK, d] (keyframe FPN tokens, concatenated);  fm bf16 [NK, C] = attn_feats of their mask features;
        grids: list (per keyframe) of (h, w).  Returns out_queries fp32 [Q,d] and the HeadState (logits, mask embedding)."""
        dev = fpn.device
        pk = self.packed(dev)
        d, H, Q, C = self.hidden_dim, self.num_heads, self.num_queries, self.mask_dim
        hd = d // H
        portrait = portrait or [False] * len(grids)
        NK = fpn.shape[0]
        assert NK == sum(h * w for h, w in grids) and fm.shape[0] == NK
        src = empty(NK, d, adt(), dev)           # value input: fpn + level_embed
        srcpos = empty(NK, d, adt(), dev)        # key input:   ... + sine PE of the view's grid
        hip.add_cast(fpn, src, b=pk['lvl'], b_mod=1)
        if len(set(zip(grids, portrait))) == 1:
            h, w = grids[0]
            hip.add_cast(src, srcpos, b=self._pe(pk, h, w, portrait[0], dev), b_mod=h * w)
        else:
            o0 = 0
            for (h, w), pt in zip(grids, portrait):
                hip.add_cast(src[o0:o0 + h * w], srcpos[o0:o0 + h * w], b=self._pe(pk, h, w, pt, dev))
                o0 += h * w
        if self.two_stage:
            out, qpos, _ = self._select_queries(pk, fpn, grids, portrait, cls_bf16)
        else:
            out, qpos = pk['qf'].clone(), pk['qe']
        NKm = ceil_to(NK, 4)                   # mask rows are 4-byte aligned and the logit GEMM wants N % 4 == 0: odd token grids (21 x 21 at 336 x 336)
        mask = torch.zeros(Q, NKm, dtype=torch.uint8, device=dev)
        logits_attn = empty(Q, NKm, torch.float32, dev)[:, :NK]
        if NKm != NK:                          # ... get up to 3 zero key rows behind the real ones (their logits are never looked at)
            fm_p = torch.zeros(NKm, C, dtype=fm.dtype, device=dev)
            fm_p[:NK].copy_(fm)
            fm = fm_p

        forced, log = self._instr if self._instr is not None else (None, None)
        if (forced is not None or log is not None) and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('MaskTransformer.instrument(): parity instrumentation must not be baked into a captured HIP graph')
        step = [0]

        def next_mask(o):
            dn, emb = self._embed(pk, o)
            if forced is not None:           # parity instrumentation (tests / bench.py): take the attention-mask BITS of this decoder layer from
                mask[:, :NK].copy_(forced[step[0]])  # outside, so that the hard threshold at logit 0 cannot amplify a 1e-3 difference into another query
            else:
                hip.gemm(emb, fm, logits_attn if NKm == NK else logits_attn.as_strided((Q, NKm), (NKm, 1)))
                hip.attn_mask_from_logits(logits_attn, mask)          # row stride NKm = NK rounded up to 4 bytes; columns >= NK are never read
            if log is not None:
                log.append(mask[:, :NK].clone())
            step[0] += 1
            return dn, emb
        next_mask(out)
        qin, ob = empty(Q, d, adt(), dev), empty(Q, d, adt(), dev)
        t32 = empty(Q, d, torch.float32, dev)
        vt = torch.zeros(d, ceil_to(NK, 8) + 8, dtype=adt(), device=dev)        # V^T scratch of the cross- / self-attention, shared by the layers (pad columns stay 0)
        vts = torch.zeros(d, ceil_to(Q, 8) + 8, dtype=adt(), device=dev)
        dn = emb = None
        for i, L in enumerate(pk['layers']):
            # masked cross-attention (post-LN): K from src+pos, V from src
            kc = empty(NK, d, adt(), dev)
            hip.gemm(srcpos, L['ca']['k'].w, kc, bias=L['ca']['k'].b)
            hip.gemm(src, L['ca']['v'].w, vt, bias=L['ca']['v'].b, trans_out=True)
            hip.add_cast(out, qin, b=qpos)
            q = empty(Q, d, adt(), dev)
            hip.gemm(qin, L['ca']['q'].w, q, bias=L['ca']['q'].b, gamma=qscale(d, d, hd, dev))
            ldv = vt.stride(0)
            hip.attention(q, kc, vt, ob, 1, H, Q, NK, hd, (0, hd, d), (0, hd, d), (0, hd * ldv, ldv), (0, hd, d),
                          mask=mask, mask_strides=(0, NKm), prescaled=True)
            hip.gemm(ob, L['ca']['o'].w, t32, bias=L['ca']['o'].b, res=out)
            hip.layernorm(t32, L['ca_norm'][0], L['ca_norm'][1], out, L['ca_norm'][2])
            # self-attention: q = k = out + query_pos, v = out
            hip.add_cast(out, qin, b=qpos)
            qk = empty(Q, 2 * d, adt(), dev)
            hip.gemm(qin, L['sa']['qk'].w, qk, bias=L['sa']['qk'].b, gamma=qscale(d, 2 * d, hd, dev))
            hip.add_cast(out, ob)
            hip.gemm(ob, L['sa']['v'].w, vts, bias=L['sa']['v'].b, trans_out=True)
            o2 = empty(Q, d, adt(), dev)
            lds = vts.stride(0)
            hip.attention(qk, qk[:, d:], vts, o2, 1, H, Q, Q, hd, (0, hd, 2 * d), (0, hd, 2 * d), (0, hd * lds, lds), (0, hd, d), prescaled=True)
            hip.gemm(o2, L['sa']['o'].w, t32, bias=L['sa']['o'].b, res=out)
            hip.layernorm(t32, L['sa_norm'][0], L['sa_norm'][1], out, L['sa_norm'][2])
            # FFN
            hip.add_cast(out, ob)
            hmid = empty(Q, L['l1'].n, adt(), dev)
            hip.gemm(ob, L['l1'].w, hmid, bias=L['l1'].b, act='relu')
            hip.gemm(hmid, L['l2'].w, t32, bias=L['l2'].b, res=out)
            hip.layernorm(t32, L['ff_norm'][0], L['ff_norm'][1], out, L['ff_norm'][2])
            dn, emb = next_mask(out) if i + 1 < self.num_layers else self._embed(pk, out)
        hs = HeadState()
        hs.embed = emb
        hs.logits = self._class_logits(pk, dn, cls_bf16)
        return out, hs


# =========================================================================================== text + PanopticDecoder
class TextEncoder(nn.Module):
    """Fixed-vocabulary text encoder (text_encoder.py:44-47,94-101): cached class embeddings -> unit-norm rows.
    Live SigLIP inference (fixed_vocab=False) needs HF weights that cannot be fetched offline: set `class_embeddings`."""

    def __init__(self, model_name='siglip', out_dim=768, fixed_vocab=True):
        super().__init__()
        self.model_name = model_name
        self.embed_dim = {'siglip': 768, 'siglip2': 768, 'clip': 512}[model_name]
        self.fixed_vocab = True
        self.class_embeddings = {}

    def change_mode(self, fixed_vocab):
        if not fixed_vocab:
            raise NotImplementedError('live text-encoder inference is outside the hot path; inject class_embeddings')

    def set_vocab(self, classes, embeddings=None, device=None):
        if embeddings is None:
            # reference text_encoder.py:44-47 would run the SigLIP text tower; no HF weights offline: the classes must already be in
            # the fixed-vocabulary store (class_embeddings), else pass the pooled text embeddings
            unknown = [c for c in classes if c not in self.class_embeddings]
            if unknown:
                raise NotImplementedError('no SigLIP text tower in this build: pass embeddings= (pooled text embeddings [Ncls, %d]) or '
                                          'fill class_embeddings; unknown classes: %s' % (self.embed_dim, unknown[:5]))
            return
        self.class_embeddings = {c: e for c, e in zip(classes, embeddings)}
        self._cls_gen = getattr(self, '_cls_gen', 0) + 1       # new vocabulary generation: cached device copies of the old one are
        #                                                        no longer handed out (runners that captured them keep them alive)

    def forward(self, classes):
        assert all(c in self.class_embeddings for c in classes), \
            "Missing classes in vocabulary. 'set_vocab' must be called if using fixed vocabulary"
        e = torch.stack([self.class_embeddings[c] for c in classes])
        return e / e.norm(dim=-1, keepdim=True)

    def normalized_bf16(self, classes, device):
        """unit-norm class embeddings as the bf16 [Ncls, 768] W operand of the class-logit GEMM (normalised on device)."""
        assert all(c in self.class_embeddings for c in classes), \
            "Missing classes in vocabulary. 'set_vocab' must be called if using fixed vocabulary"
        # identity of the embeddings in use: vocabulary generation (set_vocab), plus per class the storage address and torch's in-place edit
        # counter - an in-place edit or a direct dict assignment gets fresh device copies too
        ver = tuple((self.class_embeddings[c].data_ptr(), self.class_embeddings[c]._version) for c in classes)
        key = (tuple(classes), str(device), adt(), getattr(self, '_cls_gen', 0), ver)
        cache = self.__dict__.setdefault('_cls_cache', {})     # keyed, entries never replaced: a captured graph may hold the address
        if not isinstance(cache, dict):
            cache = self.__dict__['_cls_cache'] = {}
        if key not in cache:
            raw = torch.stack([self.class_embeddings[c] for c in classes]).float().to(device).contiguous()
            full = torch.zeros(ceil_to(raw.shape[0], 4), ceil_to(raw.shape[1], 64), dtype=adt(), device=device)
            out = full[:raw.shape[0]]                # up to three zero rows stay behind the classes (MaskTransformer._class_logits: N % 4 == 0)
            out._pst_zero_tail = True
            hip.l2norm_rows(raw, out[:, :raw.shape[1]], 0.0)
            cache[key] = out
        return cache[key]


class PanopticDecoder(HipModule):
    def __init__(self, input_mixer=None, upscaler=None, fpn_dim=(768,), hidden_dim=768, mask_dim=256, ff_dim=2048, num_queries=200,
                 num_heads=8, dec_layers=6, text_encoder='siglip', fixed_vocab=True, label_mode='sigmoid', two_stage=False,
                 landscape_only=True, deep_supervision=True):
        super().__init__()
        assert upscaler is not None, 'Upscaler module must be provided'
        if label_mode not in ('sigmoid', 'softmax'):
            raise ValueError("label_mode must be 'sigmoid' or 'softmax' (panoptic_decoder.py:20,30), got %r" % (label_mode,))
        self.input_mixer, self.upscaler = input_mixer, upscaler
        self.text_encoder = TextEncoder(text_encoder, out_dim=hidden_dim, fixed_vocab=fixed_vocab)
        self.label_mode, self.landscape_only = label_mode, landscape_only
        if label_mode == 'softmax':        # a learnt "no object" class row behind the vocabulary, NOT normalised (panoptic_decoder.py:30-31,66-67)
            self.nocls_token = nn.Parameter(torch.randn(self.text_encoder.embed_dim))
        self.mask_transformer = MaskTransformer(list(fpn_dim), hidden_dim, ff_dim, mask_dim, num_queries, num_heads, dec_layers,
                                                lang_dim=self.text_encoder.embed_dim, num_feature_levels=len(fpn_dim),
                                                landscape_only=landscape_only, two_stage=two_stage)

    def class_rows(self, classes, device):
        """the W operand of the class-logit GEMM: unit-norm class embeddings [Ncls, 768 padded to 64] in the operand format, plus - label_mode='softmax' -
        the raw `nocls_token` as row Ncls (panoptic_decoder.py:65-67).  Cached per vocabulary / token version; entries are never replaced (captured graphs)."""
        rows = self.text_encoder.normalized_bf16(classes, device)
        if self.label_mode != 'softmax':
            return rows
        cache = self.__dict__.setdefault('_rows_cache', {})
        key = (rows.data_ptr(), rows.shape, rows.dtype, self.nocls_token.data_ptr(), self.nocls_token._version)
        if key not in cache:
            full = torch.zeros(ceil_to(rows.shape[0] + 1, 4), rows.shape[1], dtype=rows.dtype, device=device)
            out = full[:rows.shape[0] + 1]
            out._pst_zero_tail = True
            out[:-1].copy_(rows)
            out[-1, :self.nocls_token.numel()].copy_(self.nocls_token.detach().to(device))
            cache[key] = (out, rows)               # `rows` kept alive: its address is part of the key
        return cache[key][0]

    def _pack(self, device):
        return {}

    def cat_width(self):
        return None

    def fpn_grid(self, h, w):
        """(rows, cols) of the FPN level / key grid the MaskTransformer sees for an h x w token grid, and whether the
        view counts as portrait: `transpose_to_landscape(upscaler)` hands portrait results back transposed (utils.py:47-49)."""
        portrait = bool(self.landscape_only and h > w)
        return ((w, h) if portrait else (h, w)), portrait

    @torch.no_grad()
    def guidance_tokens(self, imgs, h, w, mm=None):
        """Image-only part of the upscaler (LoftUp's guidance branch) or None: can run before / beside anything token-dependent."""
        up = self.upscaler
        return up.guidance_tokens(imgs, h, w, mm=mm) if isinstance(up, LoftUpUpscaler) else None

    def minmax_scaled(self):
        """True when the upscaler scales its guidance image with batch statistics (LoftUp's MinMaxScaler): results then depend on which views are
        scaled together; the pixel-shuffle upscaler is batch-invariant."""
        return isinstance(self.upscaler, LoftUpUpscaler)

    @torch.no_grad()
    def minmax_tables(self, img_stacks, scope):
        """MinMaxScaler statistics for scaling scopes wider than one view: img_stacks = list of fp32 [n_i,3,H_i,W_i] stacks, scope = int32 device tensor
        [sum n_i] of scope ids (views with equal ids are scaled together, as one chunk of the reference's batched_map, panoptic_decoder.py:50-62).
        Returns one [n_i,3,2] (min, max) table per stack, or None per stack for a batch-invariant upscaler."""
        if not self.minmax_scaled():
            return [None] * len(img_stacks)
        dev = img_stacks[0].device
        n = sum(int(t.shape[0]) for t in img_stacks)
        assert scope.numel() == n
        per_view = torch.empty(n, 3, 2, dtype=torch.float32, device=dev)
        o = 0
        for t in img_stacks:
            hip.loftup_minmax(t.float().contiguous(), per_view[o:o + t.shape[0]])
            o += t.shape[0]
        pooled = hip.minmax_merge(per_view, scope, torch.empty_like(per_view))
        out, o = [], 0
        for t in img_stacks:
            out.append(pooled[o:o + t.shape[0]])
            o += t.shape[0]
        return out

    @torch.no_grad()
    def features_tokens(self, cat, imgs, V, h, w, guidance=None, mm=None):
        """cat bf16 [V*T, 2816] (enc | dec | dino) -> (fpn bf16 [V*T, d], mask_feats bf16 [V, Hm, Wm, C]).
        Portrait views (h > w) with landscape_only=True follow `transpose_to_landscape(upscaler, dims=(2,3))`
        (panoptic_decoder.py:26,56): the upscaler runs on the tall grid and both results are handed back transposed --
        FPN tokens in the raster of the w x h grid; mask features [V, 4w, 4h, C] for the pixel-shuffle upscaler and
        (its guidance image being transposed inside, loftup.py:147-149) native [V, H/2, W/2, C] for LoftUp."""
        dev = cat.device
        up = self.upscaler
        T = h * w
        fpn = torch.empty(V * T, up.fpn_dim, dtype=adt(), device=dev)
        if isinstance(up, LoftUpUpscaler):
            lr = torch.zeros(V * T, up.lr_width(), dtype=adt(), device=dev)
            if self.input_mixer is not None:
                self.input_mixer.mix_tokens(cat, V, h, w, lr)
            else:
                lr[:, :up.input_dim] = cat
            H2, W2 = imgs.shape[2] // 2, imgs.shape[3] // 2
            mf = torch.empty(V, min(H2, W2) if h > w else H2, max(H2, W2) if h > w else W2, up.mask_dim, dtype=adt(), device=dev)
            up.upscale_tokens(lr, imgs, V, h, w, fpn, mf, guidance=guidance, mm=mm)
        else:
            x = cat
            if self.input_mixer is not None:
                x = torch.empty(V * T, self.input_mixer.hidden_dim, dtype=adt(), device=dev)
                self.input_mixer.mix_tokens(cat, V, h, w, x)
            mf = torch.empty(V, 8 * h, 8 * w, up.mask_dim, dtype=adt(), device=dev)
            up.upscale_tokens(x, imgs, V, h, w, fpn, mf)
        if self.fpn_grid(h, w)[1]:
            fpn = fpn.view(V, h, w, -1).transpose(1, 2).reshape(V * T, -1)          # copies (pure data movement)
            mf = mf.transpose(1, 2).contiguous()
        return fpn, mf

    def forward(self, in_feats, in_imgs, pos, true_shape, classes, max_bs=None, outdevice=None, memory_queries=None, multi_ar=False, _mm=None):
        """Reference signature (panoptic_decoder.py:41), images in native orientation.
        multi_ar=False: in_feats = (x_enc, y_dec, x_dino) each [B,n,T,*], in_imgs [B,n,3,H,W], true_shape [B,n,2]
            -> pred_logits [B,Q,Ncls], pred_masks [B,n,Q,H/2,W/2], out_queries [Q,B,d]; scenes of a batch are independent.
        multi_ar=True (panoptic_decoder.py:45-47, panst3r.py:248-249): every argument is a LIST of same-shape stacks ([1,n_i,...]);
            the queries are decoded against the tokens of ALL stacks, pred_masks is the list of per-stack [1,n_i,Q,h_i,w_i].
        `max_bs` as in the reference (panoptic_decoder.py:50-62, utils.py batched_map): the views of a stack (flattened over B and n) are processed
        in chunks of max_bs, None = the whole stack at once.  All the work here is batched regardless - what the chunking CHANGES is LoftUp's
        MinMaxScaler, which takes min / max over the chunk it is handed (loftup.py:14-19): views of one chunk share their scale
        (max_bs=1, the demo's setting: per view).  The pixel-shuffle variant is chunk-invariant."""
        feats = in_feats if multi_ar else tuple([f] for f in in_feats)
        imgs = in_imgs if multi_ar else [in_imgs]
        shapes = true_shape if multi_ar else [true_shape]
        B = feats[0][0].shape[0]
        # MinMaxScaler scopes: chunk ids over the flattened (B, n) views of every stack
        mms = _mm
        if mms is None and self.minmax_scaled() and max_bs != 1:
            ids, nxt = [], 0
            for im in imgs:
                tot = int(im.shape[0]) * int(im.shape[1])
                bs = tot if max_bs is None else int(max_bs)
                ids += [nxt + i // bs for i in range(tot)]
                nxt = ids[-1] + 1
            if len(set(ids)) < len(ids):
                scope = torch.tensor(ids, dtype=torch.int32).to(imgs[0].device)
                mms = self.minmax_tables([im.flatten(0, 1) for im in imgs], scope)
        if B != 1:
            if multi_ar:
                raise NotImplementedError('multi_ar stacks carry one scene (B == 1), as in the reference call site panst3r.py:248')
            n = int(in_imgs.shape[1])
            outs = [self.forward(tuple(f[b:b + 1] for f in in_feats), in_imgs[b:b + 1], None, true_shape[b:b + 1], classes, 1, outdevice,
                                 None if memory_queries is None else memory_queries[:, b:b + 1], False,
                                 _mm=None if mms is None else [mms[0][b * n:(b + 1) * n]]) for b in range(B)]
            res = {'pred_logits': torch.cat([o['pred_logits'] for o in outs]), 'pred_masks': torch.cat([o['pred_masks'] for o in outs])}
            if memory_queries is None:
                res['out_queries'] = torch.cat([o['out_queries'] for o in outs], dim=1)
                res['aux_outputs'] = []
            return res
        dev = feats[0][0].device
        p = self.upscaler.patch_size
        mt = self.mask_transformer
        cls = self.class_rows(classes, dev)
        fpns, fms, mfs, grids, portraits = [], [], [], [], []
        for si in range(len(imgs)):
            n, T = feats[0][si].shape[1:3]
            H, W = [int(v) for v in shapes[si].reshape(-1, 2)[0].tolist()]
            h, w = H // p, W // p
            cat = torch.cat([f[si].reshape(n * T, -1) for f in feats], dim=-1).to(adt()).contiguous()
            fpn, mf = self.features_tokens(cat, imgs[si][0].float().contiguous(), n, h, w, mm=None if mms is None else mms[si])
            grid, portrait = self.fpn_grid(h, w)
            fpns.append(fpn); mfs.append(mf); grids += [grid] * n; portraits += [portrait] * n
            if memory_queries is None:
                fms.append(mt.attn_feats(mf, grid))
        if memory_queries is None:
            outq, hs = mt.decode_tokens(torch.cat(fpns) if len(fpns) > 1 else fpns[0], torch.cat(fms) if len(fms) > 1 else fms[0], grids, cls, portraits)
        else:
            outq = memory_queries.reshape(-1, mt.hidden_dim).float().to(dev).contiguous()
            hs = mt.head_state(outq, cls)
        masks = [mt.masks_for_group(hs.embed, mf)[None] for mf in mfs]
        if outdevice is not None:
            masks = [m.to(outdevice) for m in masks]
        res = {'pred_logits': hs.logits[None], 'pred_masks': masks if multi_ar else masks[0]}
        if memory_queries is None:
            res['out_queries'] = outq[:, None].clone()
            res['aux_outputs'] = []
        return res
