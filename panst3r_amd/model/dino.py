"""DinoV2Encoder on the HIP path (reference model/dino.py:50-71 around HF Dinov2Model).

Own parameter tree with the HF key names under `dinov2.` (no transformers import, no network): ViT-L/14, CLS token,
learned position embedding (bicubic-interpolated to the token grid once per shape and cached), LayerScale, GELU MLP.
Normalise + bilinear resize + 14x14 patchify are HIP kernels; the patch-embed GEMM adds the position embedding as a
broadcast residual and writes behind each view's CLS row.
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip
from .common import HipModule, Packed, Layout, adt, BlockW, empty, vit_block, pack_norm, f32, ParamLinear, fold_ln, Stream


class _Proj(nn.Module):
    def __init__(self, dim, p):
        super().__init__()
        self.projection = nn.Conv2d(3, dim, kernel_size=p, stride=p)


class _Emb(nn.Module):
    def __init__(self, dim, p, image_size):
        super().__init__()
        n = (image_size // p) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.mask_token = nn.Parameter(torch.zeros(1, dim))
        self.position_embeddings = nn.Parameter(torch.zeros(1, n + 1, dim))
        self.patch_embeddings = _Proj(dim, p)


class _QKV(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.query, self.key, self.value = ParamLinear(dim, dim), ParamLinear(dim, dim), ParamLinear(dim, dim)


class _Out(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dense = ParamLinear(dim, dim)


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.attention, self.output = _QKV(dim), _Out(dim)


class _LS(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.lambda1 = nn.Parameter(torch.ones(dim))


class _Mlp(nn.Module):
    def __init__(self, dim, r):
        super().__init__()
        self.fc1, self.fc2 = ParamLinear(dim, dim * r), ParamLinear(dim * r, dim)


class _Layer(nn.Module):
    def __init__(self, dim, r, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attention = _Attn(dim)
        self.layer_scale1 = _LS(dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _Mlp(dim, r)
        self.layer_scale2 = _LS(dim)


class _Enc(nn.Module):
    def __init__(self, dim, depth, r, eps):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(dim, r, eps) for _ in range(depth)])


class _Dinov2P(nn.Module):
    def __init__(self, hidden_size, num_hidden_layers, mlp_ratio, patch_size, image_size, layer_norm_eps):
        super().__init__()
        self.embeddings = _Emb(hidden_size, patch_size, image_size)
        self.encoder = _Enc(hidden_size, num_hidden_layers, mlp_ratio, layer_norm_eps)
        self.layernorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)


TAPS = None        # diagnostics (tests/diag): a list that receives clones of the residual stream after every block


class DinoV2Encoder(HipModule):
    def __init__(self, dino_model='facebook/dinov2-large', output_stride=16, landscape_only=True, hidden_size=1024,
                 num_hidden_layers=24, num_attention_heads=16, mlp_ratio=4, patch_size=14, image_size=518, layer_norm_eps=1e-6):
        super().__init__()
        self.dinov2 = _Dinov2P(hidden_size, num_hidden_layers, mlp_ratio, patch_size, image_size, layer_norm_eps)
        self.embed_dim, self.num_heads, self.patch_size = hidden_size, num_attention_heads, patch_size
        self.output_stride, self.landscape_only = output_stride, landscape_only

    def _pack(self, device):
        d = self.dinov2
        blocks = []
        for L in d.encoder.layer:
            a = L.attention.attention
            qk = fold_ln(torch.cat([a.query.weight, a.key.weight]), torch.cat([a.query.bias, a.key.bias]), L.norm1, device)
            blocks.append(BlockW(pack_norm(L.norm1, device), qk, fold_ln(a.value.weight, a.value.bias, L.norm1, device),
                                 Packed(L.attention.output.dense.weight, L.attention.output.dense.bias, device),
                                 pack_norm(L.norm2, device), fold_ln(L.mlp.fc1.weight, L.mlp.fc1.bias, L.norm2, device),
                                 Packed(L.mlp.fc2.weight, L.mlp.fc2.bias, device),
                                 f32(L.layer_scale1.lambda1, device), f32(L.layer_scale2.lambda1, device)))
        pe = d.embeddings.patch_embeddings.projection
        return dict(patch=Packed(pe.weight, pe.bias, device), blocks=blocks, norm=pack_norm(d.layernorm, device), pos={})

    def _pos(self, pk, gh, gw, device):
        """(cls row [1,D], patch rows [gh*gw, D]) fp32; weight preparation, cached per token grid
        (HF interpolate_pos_encoding: bicubic, align_corners=False, fp32)."""
        key = (gh, gw)
        if key not in pk['pos']:
            e = self.dinov2.embeddings
            pe = e.position_embeddings.detach().float()
            n = pe.shape[1] - 1
            s = int(round(math.sqrt(n)))
            if gh * gw == n and gh == gw:
                patch = pe[0, 1:]
            else:
                g = pe[:, 1:].reshape(1, s, s, -1).permute(0, 3, 1, 2)
                patch = F.interpolate(g, size=(gh, gw), mode='bicubic', align_corners=False).permute(0, 2, 3, 1).reshape(gh * gw, -1)
            cls = e.cls_token.detach().float()[0] + pe[0, :1]
            pk['pos'][key] = (cls.to(device).contiguous(), patch.to(device).contiguous())
        return pk['pos'][key]

    @torch.no_grad()
    def patch_width(self, device):
        return self.packed(device)['patch'].k

    def begin_tokens(self, img, patches=None, transposed=False):
        """embeddings + everything the layers need, as a state dict (see Dust3rEncoder.begin_tokens: the lock-step form of encode_tokens)"""
        dev = img.device
        pk = self.packed(dev)
        V, _, H, W = img.shape
        if transposed:
            H, W = W, H
        p, D = self.patch_size, self.embed_dim
        gh, gw = H // self.output_stride, W // self.output_stride
        lay = Layout(V, gh * gw, extra=1)
        if patches is None:
            patches = empty(V * lay.T, pk['patch'].k, adt(), dev)
            hip.patch_rows(img.contiguous(), dino=patches, p_enc=self.output_stride, p_dino=p, dino_transposed=transposed)
        cls, pospatch = self._pos(pk, gh, gw, dev)
        x = torch.zeros(lay.rows, D, dtype=torch.float32, device=dev)
        x.view(V, lay.Tp, D)[:, 0] = cls
        hip.gemm(patches, pk['patch'].w, x, bias=pk['patch'].b, res=pospatch, res_mod=lay.T, grp=lay.grp)
        if TAPS is not None:
            TAPS.append(('patches', patches.clone())); TAPS.append(('embed', x.clone()))
        return dict(pk=pk, x=x, s=Stream(x).refresh(), lay=lay, V=V)

    def blocks(self, st):
        D, Hh = self.embed_dim, self.num_heads
        return [(st['s'], bw, st['lay'], Hh, D // Hh, None, None) for bw in st['pk']['blocks']]

    def finish_tokens(self, st, out, col0=0):
        pk, lay, D = st['pk'], st['lay'], self.embed_dim
        hip.layernorm(st['x'], pk['norm'][0], pk['norm'][1], out[:, col0:col0 + D], pk['norm'][2], rows=st['V'] * lay.T, grp=lay.grp)
        return out

    def encode_tokens(self, img, out, col0=0, patches=None, transposed=False):
        """img fp32 [V,3,H,W] in [-1,1] -> 16-bit tokens written to out[:, col0:col0+D].  `transposed`: DINOv2 runs on the TRANSPOSED image
        (portrait views with landscape_only, model/dino.py:15-47) - sampled with swapped axes, no transposed copy.  ImageNet
        normalisation + bilinear resize to the 14-pixel grid + 14x14 patch rows are one kernel (hip.patch_rows); `patches` when the
        caller already produced them together with the encoder's."""
        st = self.begin_tokens(img, patches, transposed)
        for i, args in enumerate(self.blocks(st)):
            vit_block(*args)
            if TAPS is not None:
                TAPS.append(('block %d' % i, st['x'].clone()))
        return self.finish_tokens(st, out, col0)

    def forward(self, image, true_shape):
        """Reference signature (model/dino.py:59-71): [b,3,H,W], true_shape [b,2] -> [b,T,1024] (CLS dropped)."""
        V = image.shape[0]
        hh, ww = true_shape.T
        land = (ww >= hh)
        x = image.float().contiguous()
        tr = False
        if self.landscape_only and not bool(land.all()):
            if not bool((~land).all()):
                raise NotImplementedError('mixed-orientation batches: call once per orientation')
            tr = True                                # dinov2_transpose (model/dino.py:15-47): portrait views run transposed
        out = torch.empty(V * (x.shape[2] // self.output_stride) * (x.shape[3] // self.output_stride), self.embed_dim,
                          dtype=adt(), device=x.device)
        self.encode_tokens(x, out, transposed=tr)
        return out.float().reshape(V, -1, self.embed_dim)
