"""Dust3rEncoder on the HIP path: CroCo ViT-L/16 with 2-D RoPE ([3P-recalled] -- see oracle/must3r.py).

Interface kept from the reference call sites: `encoder(img [b,3,H,W], true_shape [b,2]) -> (x [b,T,1024], pos [b,T,2])`
(engine/must3r.py:17-19), attribute `patch_size` (tools/demo_panst3r.py:207), ctor kwargs of configs/base.yaml:7-10.
All views of a call are batched through every GEMM (M = b*T rows); attention is per view.
"""
import torch
import torch.nn as nn

from .. import hip
from .common import (HipModule, Packed, Layout, adt, empty, vit_block, pack_croco_block, pack_norm, grid_pos, grow_table, Stream)
from .params import BlockP


class _PatchEmbedP(nn.Module):
    def __init__(self, patch_size, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=patch_size, stride=patch_size)


class Dust3rEncoder(HipModule):
    def __init__(self, img_size=(224, 224), patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0,
                 patch_embed='PatchEmbedDust3R', pos_embed='RoPE100', **kw):
        super().__init__()
        assert patch_embed == 'PatchEmbedDust3R' and pos_embed.startswith('RoPE')
        self.patch_size, self.embed_dim, self.num_heads = patch_size, embed_dim, num_heads
        self.rope_base = float(pos_embed[4:])
        self.patch_embed = _PatchEmbedP(patch_size, embed_dim)
        self.blocks_enc = nn.ModuleList([BlockP(embed_dim, mlp_ratio, True, 1e-6) for _ in range(depth)])
        self.norm_enc = nn.LayerNorm(embed_dim, eps=1e-6)

    def _pack(self, device):
        return dict(patch=Packed(self.patch_embed.proj.weight, self.patch_embed.proj.bias, device),
                    blocks=[pack_croco_block(b, device) for b in self.blocks_enc],
                    norm=pack_norm(self.norm_enc, device), rope={})

    def rope_table(self, pk, n, hd, device):
        return grow_table(pk['rope'], n, lambda m: hip.rope_table(m, hd, self.rope_base, device))

    @torch.no_grad()
    def begin_tokens(self, img, patches=None):
        """patch embedding + everything the blocks need, as a state dict; `blocks(state)` yields one vit_block argument tuple per layer and
        `finish_tokens` applies the final norm - split so that PanSt3R.encode_views can run this ViT in lock-step with DINOv2 (vit_block_pair)."""
        dev = img.device
        pk = self.packed(dev)
        V, _, H, W = img.shape
        p, D, Hh = self.patch_size, self.embed_dim, self.num_heads
        gh, gw = H // p, W // p
        lay = Layout(V, gh * gw)
        if patches is None:
            patches = empty(V * lay.T, pk['patch'].k, adt(), dev)
            hip.patch_rows(img.contiguous(), enc=patches, p_enc=p)
        x = torch.zeros(lay.rows, D, dtype=torch.float32, device=dev)
        hip.gemm(patches, pk['patch'].w, x, bias=pk['patch'].b, grp=lay.grp)
        pos = grid_pos(V, gh, gw, lay.Tp, 0, dev)
        rope = self.rope_table(pk, max(gh, gw), D // Hh, dev)
        return dict(pk=pk, x=x, s=Stream(x).refresh(), lay=lay, pos=pos, rope=rope, V=V, gh=gh, gw=gw, dev=dev)

    def blocks(self, st):
        D, Hh = self.embed_dim, self.num_heads
        return [(st['s'], bw, st['lay'], Hh, D // Hh, st['pos'], st['rope']) for bw in st['pk']['blocks']]

    def finish_tokens(self, st, out=None, copy=None):
        """final norm -> out[:, :D]; `copy`: a second buffer (another format) that receives the SAME fp32 result rounded once to its own format - the
        decoder's copy of the encoder tokens when the feature concat is kept in the panoptic decoder's format (one more launch of a 40 us kernel
        instead of a rounding of a rounding)"""
        pk, lay, D = st['pk'], st['lay'], self.embed_dim
        if out is None:
            out = empty(st['V'] * lay.T, D, adt(), st['dev'])
        for dst in (out, copy):
            if dst is not None:
                hip.layernorm(st['x'], pk['norm'][0], pk['norm'][1], dst[:, :D] if dst.shape[1] != D else dst, pk['norm'][2],
                              rows=st['V'] * lay.T, grp=lay.grp)
        return out, grid_pos(st['V'], st['gh'], st['gw'], lay.T, 0, st['dev'])

    @torch.no_grad()
    def encode_tokens(self, img, out=None, patches=None, copy=None):
        """img fp32 [V,3,H,W] (one shape) -> 16-bit tokens [V*T, out_ld] written into `out[:, :D]` (or a new buffer),
        plus int32 positions [V*T, 2].  `patches`: the 16x16 patch rows when the caller already produced them (hip.patch_rows makes the
        rows of both ViTs in one launch)."""
        st = self.begin_tokens(img, patches)
        for args in self.blocks(st):
            vit_block(*args)
        return self.finish_tokens(st, out, copy)

    def forward(self, img, true_shape=None):
        V = img.shape[0]
        tok, pos = self.encode_tokens(img.float())
        return tok.float().reshape(V, -1, self.embed_dim), pos.reshape(V, -1, 2).long()
