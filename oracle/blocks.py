"""Oracle restatement of the croco / must3r primitive blocks (TEST INFRASTRUCTURE).

[3P-recalled -- parity unpinned]  The upstream `croco.models.blocks` and
`must3r.model.blocks.pos_embed` are not vendored in /root/reference; what is
restated here follows the reference's call sites:
  * Mlp ctor use          -- model/upscalers/pixel_shuffle.py:17-27
  * Block(dim, heads, mlp_ratio, rope=, qkv_bias=) called blk(x, pos)
                           -- model/input_mixer.py:18-20,25-26
  * CrossAttention(dim, rope=, num_heads=, qkv_bias=, attn_drop=, proj_drop=)
    called (q, k, v, qpos, kpos)          -- model/blocks.py:18-19,32
  * get_pos_embed('RoPE100')             -- model/input_mixer.py:16
State-dict sub-names (fc1/fc2, qkv/proj, projq/projk/projv) are the public
croco names.
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F


class DropPath(nn.Module):
    """Stochastic depth; identity at inference (the only mode the oracle runs)."""

    def __init__(self, p=0.0):
        super().__init__()
        self.p = p

    def forward(self, x):
        return x


class RoPE2D(nn.Module):
    """2-D rotary embedding, 'RoPE<freq>' with F0=1.

    Per head the first hd/2 channels rotate with the y coordinate (pos[...,0]),
    the last hd/2 with x (pos[...,1]).  Each half is a 1-D RoPE over D=hd/2:
    inv_freq_i = freq^(-2i/D), i in [0, D/2); angle table = cat(ang, ang);
    out = t*cos + rotate_half(t)*sin, rotate_half(t) = cat(-t[D/2:], t[:D/2]).
    """

    def __init__(self, freq=100.0, F0=1.0):
        super().__init__()
        self.base = freq
        self.F0 = F0

    def _tables(self, D, npos, device):
        inv = 1.0 / (self.base ** (torch.arange(0, D, 2, dtype=torch.float32, device=device) / D))
        t = torch.arange(npos, dtype=torch.float32, device=device)
        ang = torch.outer(t, inv) * self.F0
        ang = torch.cat([ang, ang], dim=-1)
        return ang.cos(), ang.sin()

    @staticmethod
    def _rot_half(t):
        a, b = t[..., : t.shape[-1] // 2], t[..., t.shape[-1] // 2:]
        return torch.cat([-b, a], dim=-1)

    def _apply1d(self, t, p, cos, sin):
        # t [B,h,N,D]; p [B,N] int
        c = F.embedding(p, cos)[:, None]
        s = F.embedding(p, sin)[:, None]
        return t * c + self._rot_half(t) * s

    def forward(self, tokens, positions):
        """tokens [B,h,N,hd]; positions [B,N,2] (y,x) integer."""
        hd = tokens.shape[-1]
        D = hd // 2
        cos, sin = self._tables(D, int(positions.max()) + 1, tokens.device)
        ty, tx = tokens[..., :D], tokens[..., D:]
        ty = self._apply1d(ty, positions[..., 0], cos, sin)
        tx = self._apply1d(tx, positions[..., 1], cos, sin)
        return torch.cat([ty, tx], dim=-1)


def get_pos_embed(name):
    assert name.startswith('RoPE'), name
    return RoPE2D(freq=float(name[len('RoPE'):]))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, bias=True, drop=0.0):
        super().__init__()
        hidden_features = hidden_features or in_features
        out_features = out_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def _sdpa(q, k, v, mask=None):
    """softmax(q k^T / sqrt(d)) v, explicit (no fused backend) so it is a plain oracle."""
    s = torch.matmul(q, k.transpose(-1, -2)) * (q.shape[-1] ** -0.5)
    if mask is not None:
        s = s.masked_fill(mask, float('-inf'))
    return torch.matmul(s.softmax(dim=-1), v)


class Attention(nn.Module):
    def __init__(self, dim, rope=None, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, 3 * dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.rope = rope

    def forward(self, x, xpos):
        B, N, C = x.shape
        h = self.num_heads
        qkv = self.qkv(x).reshape(B, N, 3, h, C // h).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if self.rope is not None:
            q = self.rope(q, xpos)
            k = self.rope(k, xpos)
        o = _sdpa(q, k, v).transpose(1, 2).reshape(B, N, C)
        return self.proj(o)


class Block(nn.Module):
    """pre-LN ViT block: x += attn(norm1(x), pos); x += mlp(norm2(x))."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 act_layer=nn.GELU, norm_layer=nn.LayerNorm, rope=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, rope=rope, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer)

    def forward(self, x, xpos):
        x = x + self.attn(self.norm1(x), xpos)
        return x + self.mlp(self.norm2(x))


class CrossAttention(nn.Module):
    def __init__(self, dim, rope=None, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        self.projq = nn.Linear(dim, dim, bias=qkv_bias)
        self.projk = nn.Linear(dim, dim, bias=qkv_bias)
        self.projv = nn.Linear(dim, dim, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)
        self.rope = rope

    def forward(self, query, key, value, qpos, kpos):
        B, Nq, C = query.shape
        Nk = key.shape[1]
        h = self.num_heads
        q = self.projq(query).reshape(B, Nq, h, C // h).transpose(1, 2)
        k = self.projk(key).reshape(B, Nk, h, C // h).transpose(1, 2)
        v = self.projv(value).reshape(B, Nk, h, C // h).transpose(1, 2)
        if self.rope is not None:
            q = self.rope(q, qpos)
            k = self.rope(k, kpos)
        o = _sdpa(q, k, v).transpose(1, 2).reshape(B, Nq, C)
        return self.proj(o)


class CrossonlyDecoderBlock(nn.Module):
    """croco DecoderBlock minus the self-attention line (reference model/blocks.py:9-35):
    y_ = norm_y(y); x += cross_attn(norm2(x), y_, y_); x += mlp(norm3(x))."""

    def __init__(self, dim, num_heads, pos_embed=None, mlp_ratio=4.0, qkv_bias=False, norm_layer=nn.LayerNorm,
                 act_layer=nn.GELU, norm_mem=True):
        super().__init__()
        self.cross_attn = CrossAttention(dim, rope=pos_embed, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.norm3 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer)
        self.norm_y = norm_layer(dim) if norm_mem else nn.Identity()

    def forward(self, x, y, xpos, ypos):
        y_ = self.norm_y(y)
        x = x + self.cross_attn(self.norm2(x), y_, y_, xpos, ypos)
        x = x + self.mlp(self.norm3(x))
        return x, y
