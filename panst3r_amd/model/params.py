"""Parameter containers with the croco / must3r / HF state-dict key names (no compute)."""
import torch
import torch.nn as nn

from .common import ParamLinear


class MlpP(nn.Module):
    """croco Mlp: fc1, fc2 (reference ctor use: model/upscalers/pixel_shuffle.py:17-27)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, **kw):
        super().__init__()
        self.fc1 = ParamLinear(in_features, hidden_features or in_features)
        self.fc2 = ParamLinear(hidden_features or in_features, out_features or in_features)


class AttnP(nn.Module):
    def __init__(self, dim, qkv_bias=True):
        super().__init__()
        self.qkv = ParamLinear(dim, 3 * dim, bias=qkv_bias)
        self.proj = ParamLinear(dim, dim)


class BlockP(nn.Module):
    """croco Block(dim, heads, mlp_ratio, qkv_bias, rope): norm1, attn.{qkv,proj}, norm2, mlp.{fc1,fc2}."""

    def __init__(self, dim, mlp_ratio=4.0, qkv_bias=True, eps=1e-5):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = AttnP(dim, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = MlpP(dim, int(dim * mlp_ratio))


class CrossAttnP(nn.Module):
    def __init__(self, dim, qkv_bias=False):
        super().__init__()
        self.projq = ParamLinear(dim, dim, bias=qkv_bias)
        self.projk = ParamLinear(dim, dim, bias=qkv_bias)
        self.projv = ParamLinear(dim, dim, bias=qkv_bias)
        self.proj = ParamLinear(dim, dim)


class MHAP(nn.Module):
    """nn.MultiheadAttention parameter layout (mask_transformer.py:314,372)."""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = ParamLinear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
