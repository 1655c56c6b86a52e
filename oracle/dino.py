"""Oracle restatement of the DINOv2 branch (TEST INFRASTRUCTURE).

Follows reference model/dino.py:50-71 (DinoV2Encoder: [-1,1] -> ImageNet normalise -> bilinear resize to
(H/16*14, W/16*14) -> HF Dinov2Model -> drop CLS) and the HF `Dinov2Model` forward it wraps
(transformers 5.15.0 modeling_dinov2.py: CLS + bicubic-interpolated learned pos-emb, pre-LN blocks with
separate query/key/value Linears, LayerScale, GELU MLP, LN eps 1e-6).  State-dict keys are the HF ones under
`dinov2.` so a reference checkpoint loads unchanged.  Pinned by tests/golden/dino_tiny.npz (generated from the
reference wrapper around the installed HF implementation).
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Proj(nn.Module):
    def __init__(self, cin, cout, p):
        super().__init__()
        self.projection = nn.Conv2d(cin, cout, kernel_size=p, stride=p)


class _Embeddings(nn.Module):
    def __init__(self, dim, patch, image_size):
        super().__init__()
        npos = (image_size // patch) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.mask_token = nn.Parameter(torch.zeros(1, dim))
        self.position_embeddings = nn.Parameter(torch.zeros(1, npos + 1, dim))
        self.patch_embeddings = _Proj(3, dim, patch)
        self.patch = patch

    def pos_for(self, gh, gw):
        pe = self.position_embeddings
        npos = pe.shape[1] - 1
        if gh * gw == npos and gh == gw:
            return pe
        s = int(round(math.sqrt(npos)))
        grid = pe[:, 1:].reshape(1, s, s, -1).permute(0, 3, 1, 2).float()
        grid = F.interpolate(grid, size=(gh, gw), mode='bicubic', align_corners=False)
        return torch.cat([pe[:, :1], grid.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], dim=1)

    def forward(self, px):
        B, _, H, W = px.shape
        x = self.patch_embeddings.projection(px).flatten(2).transpose(1, 2)
        x = torch.cat([self.cls_token.expand(B, -1, -1), x], dim=1)
        return x + self.pos_for(H // self.patch, W // self.patch)


class _SelfAttn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.query = nn.Linear(dim, dim)
        self.key = nn.Linear(dim, dim)
        self.value = nn.Linear(dim, dim)


class _AttnOut(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dense = nn.Linear(dim, dim)


class _Attn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.attention = _SelfAttn(dim)
        self.output = _AttnOut(dim)
        self.heads = heads

    def forward(self, x):
        B, N, C = x.shape
        h = self.heads
        sp = lambda t: t.reshape(B, N, h, C // h).transpose(1, 2)
        q, k, v = sp(self.attention.query(x)), sp(self.attention.key(x)), sp(self.attention.value(x))
        a = (q @ k.transpose(-1, -2)) * ((C // h) ** -0.5)
        o = (a.softmax(-1) @ v).transpose(1, 2).reshape(B, N, C)
        return self.output.dense(o)


class _LS(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.lambda1 = nn.Parameter(torch.ones(dim))


class _MLP(nn.Module):
    def __init__(self, dim, ratio):
        super().__init__()
        self.fc1 = nn.Linear(dim, dim * ratio)
        self.fc2 = nn.Linear(dim * ratio, dim)


class _Layer(nn.Module):
    def __init__(self, dim, heads, ratio, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attention = _Attn(dim, heads)
        self.layer_scale1 = _LS(dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = _MLP(dim, ratio)
        self.layer_scale2 = _LS(dim)

    def forward(self, x):
        x = x + self.layer_scale1.lambda1 * self.attention(self.norm1(x))
        y = self.mlp.fc2(F.gelu(self.mlp.fc1(self.norm2(x))))
        return x + self.layer_scale2.lambda1 * y


class _Encoder(nn.Module):
    def __init__(self, dim, depth, heads, ratio, eps):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(dim, heads, ratio, eps) for _ in range(depth)])


class Dinov2(nn.Module):
    def __init__(self, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, mlp_ratio=4, patch_size=14,
                 image_size=518, layer_norm_eps=1e-6):
        super().__init__()
        self.hidden_size = hidden_size
        self.patch_size = patch_size
        self.embeddings = _Embeddings(hidden_size, patch_size, image_size)
        self.encoder = _Encoder(hidden_size, num_hidden_layers, num_attention_heads, mlp_ratio, layer_norm_eps)
        self.layernorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)

    def forward(self, px):
        x = self.embeddings(px)
        for lyr in self.encoder.layer:
            x = lyr(x)
        return self.layernorm(x)


IMNET_MEAN = (0.485, 0.456, 0.406)
IMNET_STD = (0.229, 0.224, 0.225)


class DinoV2Encoder(nn.Module):
    def __init__(self, dino_model='facebook/dinov2-large', output_stride=16, landscape_only=True, **cfg):
        super().__init__()
        self.dinov2 = Dinov2(**cfg)
        self.embed_dim = self.dinov2.hidden_size
        self.output_stride = output_stride
        self.landscape_only = landscape_only

    def _run(self, x, true_shape):
        # reference dinov2_transpose (model/dino.py:15-47): portrait views are fed transposed
        if not self.landscape_only:
            return self.dinov2(x)
        h, w = true_shape.T
        land = w >= h
        if bool(land.all()):
            return self.dinov2(x)
        if bool((~land).all()):
            return self.dinov2(x.transpose(2, 3))
        lo = self.dinov2(x[land])
        po = self.dinov2(x[~land].transpose(2, 3))
        out = lo.new_zeros(x.shape[0], *lo.shape[1:])
        out[land] = lo
        out[~land] = po
        return out

    def forward(self, image, true_shape):
        mean = image.new_tensor(IMNET_MEAN).view(1, 3, 1, 1)
        std = image.new_tensor(IMNET_STD).view(1, 3, 1, 1)
        x = ((image * 0.5 + 0.5) - mean) / std
        p = self.dinov2.patch_size
        h, w = [s // self.output_stride * p for s in image.shape[-2:]]
        x = F.interpolate(x, size=(h, w), mode='bilinear', align_corners=False)
        return self._run(x, true_shape)[:, 1:]
