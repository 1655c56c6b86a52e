"""Oracle restatement of the must3r pieces PanSt3R calls (TEST INFRASTRUCTURE).

[3P-recalled -- parity unpinned]  `must3r` / `croco` / `dust3r` are absent from
/root/reference (pyproject.toml:14, no pin).  What is fixed by the reference's own
call sites, and followed here:
  * encoder(img [b,3,H,W], true_shape [b,2]) -> (x [b,T,1024], pos [b,T,2])      engine/must3r.py:17-19
  * decoder(x [B,n,T,1024], pos, true_shape, mem|None, render=, return_feats=True)
        -> (mem, pointmaps [B,n,H,W,7], feats) with feats[-1] [B,n,T,768]        engine/must3r.py:45-46,93-94
  * mem = (mem_vals: list[L] of [B,Nmem,768], mem_labels [B,Nmem], mem_nimgs,
           mem_protected_imgs, mem_protected_tokens)                              engine/must3r.py:76-80
  * memory batches [2,1,1,...]                                                   panst3r.py:65-70
  * ctor kwargs Dust3rEncoder(img_size, patch_embed='PatchEmbedDust3R'),
    MUSt3R(img_size, feedback_type='single_mlp', memory_mode='norm_y')           configs/base.yaml:7-15
Everything else (see DESIGN.md "restated third-party spec") is this project's
own specification of the un-pinned parts; the HIP path implements exactly this.

Restated MUSt3R decoder spec
----------------------------
  t0      = feat_embed_enc_to_dec(x) (+ image2_embed unless this is scene image 0 in update mode)
  layer l : h_l = tokens entering block l   (candidate memory entry of that layer)
            x  += self_attn(norm1(x), pos)                      (RoPE-2D, base 100)
            ctx = cat(mem_vals[l], h_l of the *other* images of this call if not render)
            x  += cross_attn(norm2(x), norm_y(ctx), norm_y(ctx))   (no RoPE: the mem tuple carries no positions)
            x  += mlp(norm3(x))
  out     = norm_dec(x);  pointmaps = pixel_shuffle16(head_dec.proj(out)) -> [H,W,7] (xyz, local xyz, conf; raw)
  update  : fb = feedback_layer(feedback_norm(out))  ('single_mlp': one MLP for all layers)
            mem_vals[l] <- cat(mem_vals[l], h_l + fb)   for every layer l
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from functools import partial

from .blocks import Block, Mlp, Attention, CrossAttention, get_pos_embed

LN6 = partial(nn.LayerNorm, eps=1e-6)


class PatchEmbedDust3R(nn.Module):
    def __init__(self, img_size=(224, 224), patch_size=16, in_chans=3, embed_dim=1024):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, img, true_shape=None):
        B, _, H, W = img.shape
        p = self.patch_size
        assert H % p == 0 and W % p == 0, (H, W)
        x = self.proj(img).flatten(2).transpose(1, 2)
        ys, xs = torch.meshgrid(torch.arange(H // p, device=img.device), torch.arange(W // p, device=img.device), indexing='ij')
        pos = torch.stack([ys, xs], dim=-1).reshape(1, -1, 2).expand(B, -1, -1)
        return x, pos


class Dust3rEncoder(nn.Module):
    """CroCo ViT-L/16 with 2-D RoPE (base 100): patch-embed -> 24 pre-LN blocks -> LN."""

    def __init__(self, img_size=(224, 224), patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.0,
                 patch_embed='PatchEmbedDust3R', pos_embed='RoPE100', **kw):
        super().__init__()
        assert patch_embed == 'PatchEmbedDust3R'
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.patch_embed = PatchEmbedDust3R(tuple(img_size), patch_size, 3, embed_dim)
        self.rope = get_pos_embed(pos_embed)
        self.blocks_enc = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=LN6, rope=self.rope) for _ in range(depth)])
        self.norm_enc = LN6(embed_dim)

    def forward(self, img, true_shape=None):
        x, pos = self.patch_embed(img, true_shape)
        for blk in self.blocks_enc:
            x = blk(x, pos)
        return self.norm_enc(x), pos


class MemDecoderBlock(nn.Module):
    """croco DecoderBlock with the memory as cross-attention context."""

    def __init__(self, dim, num_heads, mlp_ratio, rope):
        super().__init__()
        self.norm1 = LN6(dim)
        self.attn = Attention(dim, rope=rope, num_heads=num_heads, qkv_bias=True)
        self.norm2 = LN6(dim)
        self.cross_attn = CrossAttention(dim, rope=None, num_heads=num_heads, qkv_bias=True)
        self.norm3 = LN6(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.norm_y = LN6(dim)

    def forward(self, x, pos, ctx):
        x = x + self.attn(self.norm1(x), pos)
        y = self.norm_y(ctx)
        x = x + self.cross_attn(self.norm2(x), y, y, None, None)
        return x + self.mlp(self.norm3(x))


class LinearHead(nn.Module):
    def __init__(self, dim, patch_size, out_ch):
        super().__init__()
        self.patch_size = patch_size
        self.out_ch = out_ch
        self.proj = nn.Linear(dim, out_ch * patch_size ** 2)

    def forward(self, tokens, H, W):
        p = self.patch_size
        B = tokens.shape[0]
        f = self.proj(tokens).transpose(-1, -2).reshape(B, -1, H // p, W // p)
        return F.pixel_shuffle(f, p).permute(0, 2, 3, 1)


class MUSt3R(nn.Module):
    def __init__(self, img_size=(224, 224), patch_size=16, enc_embed_dim=1024, embed_dim=768, depth=12, num_heads=12,
                 mlp_ratio=4.0, pos_embed='RoPE100', feedback_type='single_mlp', memory_mode='norm_y',
                 pointmap_channels=7, **kw):
        super().__init__()
        assert feedback_type in ('single_mlp', None) and memory_mode == 'norm_y'
        self.patch_size = patch_size
        self.embed_dim = embed_dim
        self.depth = depth
        self.feedback_type = feedback_type
        self.rope = get_pos_embed(pos_embed)
        self.feat_embed_enc_to_dec = nn.Linear(enc_embed_dim, embed_dim)
        self.image2_embed = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.blocks_dec = nn.ModuleList([MemDecoderBlock(embed_dim, num_heads, mlp_ratio, self.rope) for _ in range(depth)])
        self.norm_dec = LN6(embed_dim)
        if feedback_type == 'single_mlp':
            self.feedback_norm = LN6(embed_dim)
            self.feedback_layer = Mlp(embed_dim, int(mlp_ratio * embed_dim), embed_dim)
        self.head_dec = LinearHead(embed_dim, patch_size, pointmap_channels)

    # ---- core on per-image lists: xs[i] [B,T_i,Denc] ----
    def forward_list(self, xs, poss, shapes, mem, render):
        L = self.depth
        n = len(xs)
        B = xs[0].shape[0]
        if mem is None:
            mem_vals = [None] * L
            mem_labels = torch.zeros(B, 0, dtype=torch.long, device=xs[0].device)
            mem_nimgs, prot_i, prot_t = 0, 0, 0
        else:
            mem_vals, mem_labels, mem_nimgs, prot_i, prot_t = mem
            mem_vals = list(mem_vals)
        toks = []
        for i, x in enumerate(xs):
            t = self.feat_embed_enc_to_dec(x)
            if render or (mem_nimgs + i) > 0:
                t = t + self.image2_embed
            toks.append(t)
        entries = [[None] * n for _ in range(L)]
        for l, blk in enumerate(self.blocks_dec):
            h = list(toks)
            for i in range(n):
                parts = [] if mem_vals[l] is None else [mem_vals[l]]
                if not render:
                    parts += [h[j] for j in range(n) if j != i]
                ctx = torch.cat(parts, dim=1)
                toks[i] = blk(h[i], poss[i], ctx)
                entries[l][i] = h[i]
        outs = [self.norm_dec(t) for t in toks]
        pms = [self.head_dec(o, int(s[0]), int(s[1])) for o, s in zip(outs, shapes)]
        if not render:
            for i in range(n):
                fb = self.feedback_layer(self.feedback_norm(outs[i])) if self.feedback_type else 0.0
                for l in range(L):
                    e = entries[l][i] + fb
                    mem_vals[l] = e if mem_vals[l] is None else torch.cat([mem_vals[l], e], dim=1)
                lab = torch.full((B, xs[i].shape[1]), mem_nimgs + i, dtype=torch.long, device=xs[i].device)
                mem_labels = torch.cat([mem_labels, lab], dim=1)
            mem_nimgs = mem_nimgs + n
        return (mem_vals, mem_labels, mem_nimgs, prot_i, prot_t), pms, outs

    def forward(self, x, pos, true_shape, mem=None, render=False, return_feats=False):
        """x [B,n,T,Denc]; pos [B,n,T,2]; true_shape [B,n,2] (all images of the call share one shape)."""
        n = x.shape[1]
        shapes = [true_shape[0, i].tolist() for i in range(n)]
        mem, pms, outs = self.forward_list([x[:, i] for i in range(n)], [pos[:, i] for i in range(n)], shapes, mem, render)
        pointmaps = torch.stack(pms, dim=1)
        feats = [torch.stack(outs, dim=1)]
        return (mem, pointmaps, feats) if return_feats else (mem, pointmaps)


# ---- engine helpers restated from their call sites (panst3r.py:205-216, engine/must3r.py:13-15) ----
@torch.no_grad()
def encoder_multi_ar(encoder, imgs, true_shape, **kw):
    xs, ps = [], []
    for im, ts in zip(imgs, true_shape):
        x, p = encoder(im[None], ts[None])
        xs.append(x[0])
        ps.append(p[0])
    return xs, ps


def mem_batches_for(n_imgs, init_num_views=2, batch_num_views=1):
    """panst3r.py:65-70 (for n_imgs >= 2)."""
    out = [init_num_views]
    while sum(out) != n_imgs:
        out.append(min(batch_num_views, n_imgs - sum(out)))
    return out


@torch.no_grad()
def build_memory(decoder, xs, poss, shapes, mem_batches):
    """Sequential keyframe memory build (stands for must3r inference_multi_ar(..., to_render=[], return_mem=True))."""
    mem, start = None, 0
    for nb in mem_batches:
        sl = slice(start, start + nb)
        mem, _, _ = decoder.forward_list([x[None] for x in xs[sl]], [p[None] for p in poss[sl]], shapes[sl], mem, render=False)
        start += nb
    return mem
