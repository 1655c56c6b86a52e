"""Oracle restatement of the PanSt3R-owned panoptic half (TEST INFRASTRUCTURE).

Pinned against golden vectors generated from the reference's own modules
(tests/golden/make_golden.py).  Citations are to /root/reference/src/panst3r/.

  InputMixer            model/input_mixer.py:9-29
  PixelShuffleUpscaler  model/upscalers/pixel_shuffle.py:9-59
  LoftUpUpscaler        model/upscalers/loftup.py:9-190  (+ MinMaxScaler :9-19, ImplicitFeaturizer :21-79)
  MaskTransformer       model/mask_transformer.py:12-288 (+ layers :309-470, MLP :473-485, sine PE :488-527)
  TextEncoder           model/text_encoder.py:94-103 (fixed-vocab branch only; no HF weights offline)
  PanopticDecoder       model/panoptic_decoder.py:16-77
  transpose_to_landscape utils.py:8-61
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F

from .blocks import Block, Mlp, CrossonlyDecoderBlock, get_pos_embed


# --------------------------------------------------------------------------- host glue
def _swap(val, dims):
    if isinstance(val, dict):
        return {k: _swap(v, dims) for k, v in val.items()}
    if isinstance(val, (list, tuple)):
        return type(val)(_swap(v, dims) for v in val)
    return val.swapaxes(*dims)


def _merge(l_res, p_res, land):
    if isinstance(l_res, dict):
        return {k: _merge(l_res[k], p_res[k], land) for k in l_res}
    if isinstance(l_res, (list, tuple)):
        return type(l_res)(_merge(a, b, land) for a, b in zip(l_res, p_res))
    out = l_res.new_empty(l_res.shape[0] + p_res.shape[0], *l_res.shape[1:])
    out[land] = l_res
    out[~land] = p_res
    return out


def call_in_landscape(head, decout, true_shape, dims, activate=True):
    """utils.py:8-61: run `head` per orientation, portrait results are swapped back on `dims`."""
    if not activate:
        H, W = true_shape[0].tolist()
        return head(decout, (H, W))
    H, W = int(true_shape.min()), int(true_shape.max())
    hh, ww = true_shape.T
    land = ww >= hh
    if bool(land.all()):
        return head(decout, (H, W))
    if bool((~land).all()):
        return _swap(head(decout, (W, H)), dims)
    l_res = head([d[land] for d in decout], (H, W))
    p_res = _swap(head([d[~land] for d in decout], (W, H)), dims)
    return _merge(l_res, p_res, land)


def _chunks(n, bs):
    bs = n if bs is None else bs
    return [(s, min(s + bs, n)) for s in range(0, n, bs)]


# --------------------------------------------------------------------------- mixer + upscalers
class InputMixer(nn.Module):
    def __init__(self, img_size, patch_size, in_dim, hidden_dim, num_heads=12, num_layers=3, ff_dim_mult=4):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.in_proj = nn.Linear(in_dim, hidden_dim)
        self.rope = get_pos_embed('RoPE100')
        self.mixer_blk = nn.ModuleList([Block(hidden_dim, num_heads, mlp_ratio=ff_dim_mult, rope=self.rope, qkv_bias=True)
                                        for _ in range(num_layers)])
        self.mixer_norm = nn.LayerNorm(hidden_dim)

    def forward(self, x, pos):
        x = self.in_proj(x)
        for b in self.mixer_blk:
            x = b(x, pos)
        return self.mixer_norm(x)


class PixelShuffleUpscaler(nn.Module):
    def __init__(self, input_dim, patch_size=16, hidden_dim_factor=4, fp_dim=(768, 512, 384, 256), fp_activation=nn.GELU, **kw):
        super().__init__()
        self.patch_size = patch_size
        f = hidden_dim_factor
        self.proj_8 = Mlp(input_dim, int(f * input_dim), fp_dim[1] * 4, act_layer=fp_activation)
        self.proj_4 = Mlp(fp_dim[1], int(f * fp_dim[1]), fp_dim[2] * 4, act_layer=fp_activation)
        self.proj_2 = Mlp(fp_dim[2], int(f * fp_dim[2]), fp_dim[3] * 4, act_layer=fp_activation)
        self.proj_16 = Mlp(input_dim, int(f * input_dim), fp_dim[0], act_layer=fp_activation)

    @staticmethod
    def _up(tok, B, h, w):
        # tokens [B, h*w, 4C] -> pixel-shuffled map [B, C, 2h, 2w]
        return F.pixel_shuffle(tok.transpose(1, 2).reshape(B, -1, h, w), 2)

    def forward(self, feats, img_shape):
        x = feats[0]
        H, W = img_shape
        h, w = H // self.patch_size, W // self.patch_size
        B = x.shape[0]
        m8 = self._up(self.proj_8(x), B, h, w)
        m4 = self._up(self.proj_4(m8.flatten(2).transpose(1, 2)), B, 2 * h, 2 * w)
        m2 = self._up(self.proj_2(m4.flatten(2).transpose(1, 2)), B, 4 * h, 4 * w)
        f16 = self.proj_16(x).transpose(1, 2).reshape(B, -1, h, w)
        return [f16], m2


# Operation order of the guidance image's x0.5 bilinear down-sampling (loftup.py:154 F.interpolate(..., scale_factor=0.5, mode='bilinear')).
# All four weights are 0.5, so the result is a quarter of a sum of four pixels - in ONE of two association orders, one ulp apart in ~30 % of the
# pixels, and LoftUp's Fourier featurizer multiplies the (min-max scaled) value by frequencies up to e^10: that ulp is 1e-4 on the features.
#   'nested'  0.5 (0.5 a00 + 0.5 a01) + 0.5 (0.5 a10 + 0.5 a11): the CUDA kernel the reference runs (ATen UpSampleBilinear2d.cu) and torch's generic
#             CPU kernel (large images) - row sums first.  The oracle's default and what the HIP path implements, at EVERY size.
#   'torch'   whatever F.interpolate does on this host: for small images / one thread torch's CPU dispatch takes its vectorised 4-tap kernel,
#             ((a00 + a01) + a10) + a11.  Used by tests/test_oracle_golden.py, whose goldens were generated by the reference's code on the CPU.
HALF_BILINEAR = 'nested'


def half_bilinear(img):
    H, W = img.shape[-2:]
    if HALF_BILINEAR == 'torch' or H % 2 or W % 2:
        return F.interpolate(img, scale_factor=0.5, mode='bilinear', align_corners=False)
    a, b = img[..., 0::2, :], img[..., 1::2, :]
    return 0.5 * (0.5 * a[..., 0::2] + 0.5 * a[..., 1::2]) + 0.5 * (0.5 * b[..., 0::2] + 0.5 * b[..., 1::2])


class MinMaxScaler(nn.Module):
    def forward(self, x):
        lo = x.amin(dim=(0, 2, 3), keepdim=True)
        hi = x.amax(dim=(0, 2, 3), keepdim=True)
        return (x - lo) / (hi - lo).clamp_min(1e-4) - 0.5


class ImplicitFeaturizer(nn.Module):
    def __init__(self, color_feats=True, n_freqs=10, learn_bias=False):
        super().__init__()
        self.color_feats = color_feats
        self.n_freqs = n_freqs
        self.dim_multiplier = 2 + (3 if color_feats else 0)
        self.learn_bias = learn_bias
        if learn_bias:
            self.biases = nn.Parameter(torch.randn(2, self.dim_multiplier, n_freqs))

    def forward(self, im):
        b, _, h, w = im.shape
        gy = torch.linspace(-1, 1, h, device=im.device)
        gx = torch.linspace(-1, 1, w, device=im.device)
        yy, xx = torch.meshgrid(gy, gx, indexing='ij')
        base = torch.stack([yy, xx])[None].expand(b, -1, -1, -1)
        if self.color_feats:
            base = torch.cat([base, im], dim=1)
        fr = torch.exp(torch.linspace(-2, 10, self.n_freqs, device=im.device)).view(1, -1, 1, 1, 1)
        ph = base[:, None] * fr                                     # [b, nf, dm, h, w]
        nf, dm = self.n_freqs, self.dim_multiplier
        if self.learn_bias:                                         # NB: a reshape of [dm,nf] storage, not a transpose (loftup.py:62-63)
            s_in = ph + self.biases[0].reshape(1, nf, dm, 1, 1)
            c_in = ph + self.biases[1].reshape(1, nf, dm, 1, 1)
        else:
            s_in = c_in = ph
        out = [torch.sin(s_in.reshape(b, nf * dm, h, w)), torch.cos(c_in.reshape(b, nf * dm, h, w))]
        if self.color_feats:
            out.append(im)
        return torch.cat(out, dim=1)


class LoftUpUpscaler(nn.Module):
    def __init__(self, input_dim, dim, output_stride=2, patch_size=16, color_feats=True, n_freqs=20, num_heads=4,
                 num_layers=2, lr_pe_type='sine'):
        super().__init__()
        assert lr_pe_type == 'sine', 'only the released configuration is restated'
        self.output_stride = output_stride
        self.patch_size = patch_size
        self.patch_embed = nn.Conv2d(input_dim, input_dim, kernel_size=1)
        start_dim = 5 * n_freqs * 2 + 3 if color_feats else 2 * n_freqs * 2
        self.lr_pe = ImplicitFeaturizer(color_feats=False, n_freqs=5, learn_bias=True)
        self.lr_input_proj = nn.Sequential(nn.Linear(input_dim + 20, dim), nn.LayerNorm(dim))
        self.fourier_feat = nn.Sequential(MinMaxScaler(), ImplicitFeaturizer(color_feats, n_freqs=n_freqs, learn_bias=True))
        self.first_conv = nn.Sequential(
            nn.GroupNorm(1, start_dim), nn.Conv2d(start_dim, dim, 3, padding=1),
            nn.GroupNorm(8, dim), nn.ReLU(), nn.Conv2d(dim, dim, 3, padding=1),
            nn.GroupNorm(8, dim), nn.ReLU())
        self.ca_transformer_blocks = nn.ModuleList([CrossonlyDecoderBlock(dim, num_heads, mlp_ratio=1) for _ in range(num_layers)])
        self.ca_transformer_norm = nn.LayerNorm(dim)

    def forward(self, inputs, img_shape):
        tok, img = inputs
        H, W = img_shape
        B = tok.shape[0]
        lr = tok.transpose(1, 2).reshape(B, -1, H // self.patch_size, W // self.patch_size)
        fpn = self.patch_embed(lr)
        if H > W:
            img = img.transpose(2, 3)
        if self.output_stride == 2:
            img = half_bilinear(img)
        elif self.output_stride != 1:
            img = F.interpolate(img, scale_factor=1.0 / self.output_stride, mode='bilinear', align_corners=False)
        g = self.first_conv(self.fourier_feat(img))
        _, C, Ho, Wo = g.shape
        q = g.flatten(2).transpose(1, 2)
        kv = torch.cat([lr, self.lr_pe(lr)], dim=1).flatten(2).transpose(1, 2)
        kv = self.lr_input_proj(kv)
        for blk in self.ca_transformer_blocks:
            q, _ = blk(q, kv, None, None)
        q = self.ca_transformer_norm(q)
        return [fpn], q.transpose(1, 2).reshape(B, C, Ho, Wo)


# --------------------------------------------------------------------------- mask transformer
class PositionEmbeddingSine(nn.Module):
    def __init__(self, num_pos_feats=64, temperature=10000, normalize=False, scale=None):
        super().__init__()
        self.npf, self.temp, self.normalize = num_pos_feats, temperature, normalize
        self.scale = 2 * math.pi if scale is None else scale

    def forward(self, x, mask=None):
        B, h, w = x.shape[0], x.shape[-2], x.shape[-1]
        ye = torch.arange(1, h + 1, dtype=torch.float32, device=x.device).view(1, h, 1).expand(B, h, w)
        xe = torch.arange(1, w + 1, dtype=torch.float32, device=x.device).view(1, 1, w).expand(B, h, w)
        if self.normalize:
            ye = ye / (h + 1e-6) * self.scale
            xe = xe / (w + 1e-6) * self.scale
        i = torch.arange(self.npf, dtype=torch.float32, device=x.device)
        div = self.temp ** (2 * torch.div(i, 2, rounding_mode='floor') / self.npf)

        def enc(e):
            p = e[..., None] / div
            return torch.stack([p[..., 0::2].sin(), p[..., 1::2].cos()], dim=4).flatten(3)
        return torch.cat([enc(ye), enc(xe)], dim=3).permute(0, 3, 1, 2)


class _MHA(nn.Module):
    """nn.MultiheadAttention parameter layout, sequence-first [L,B,E] (mask_transformer.py:314,372)."""

    def __init__(self, d, h):
        super().__init__()
        self.h = h
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)

    def forward(self, q_in, k_in, v_in, attn_mask=None):
        d = q_in.shape[-1]
        Wq, Wk, Wv = self.in_proj_weight.split(d)
        bq, bk, bv = self.in_proj_bias.split(d)
        Lq, B, _ = q_in.shape
        Lk = k_in.shape[0]
        h, hd = self.h, d // self.h
        q = F.linear(q_in, Wq, bq).reshape(Lq, B * h, hd).transpose(0, 1)
        k = F.linear(k_in, Wk, bk).reshape(Lk, B * h, hd).transpose(0, 1)
        v = F.linear(v_in, Wv, bv).reshape(Lk, B * h, hd).transpose(0, 1)
        s = torch.bmm(q * (hd ** -0.5), k.transpose(1, 2))
        if attn_mask is not None:
            s = s.masked_fill(attn_mask, float('-inf'))
        o = torch.bmm(s.softmax(-1), v).transpose(0, 1).reshape(Lq, B, d)
        return self.out_proj(o)


class SelfAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.self_attn = _MHA(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt, query_pos):
        qk = tgt + query_pos
        return self.norm(tgt + self.self_attn(qk, qk, tgt))


class CrossAttentionLayer(nn.Module):
    def __init__(self, d_model, nhead):
        super().__init__()
        self.multihead_attn = _MHA(d_model, nhead)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt, memory, memory_mask, pos, query_pos):
        a = self.multihead_attn(tgt + query_pos, memory + pos, memory, attn_mask=memory_mask)
        return self.norm(tgt + a)


class FFNLayer(nn.Module):
    def __init__(self, d_model, dim_feedforward):
        super().__init__()
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, tgt):
        return self.norm(tgt + self.linear2(F.relu(self.linear1(tgt))))


class MLP(nn.Module):
    def __init__(self, i, h, o, n):
        super().__init__()
        dims = [i] + [h] * (n - 1) + [o]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for j, l in enumerate(self.layers):
            x = l(x) if j == len(self.layers) - 1 else F.relu(l(x))
        return x


class MaskTransformer(nn.Module):
    def __init__(self, in_dim, hidden_dim, ff_dim, mask_dim, num_queries, num_heads, dec_layers, lang_dim=768,
                 num_feature_levels=1, landscape_only=False, **kw):
        super().__init__()
        assert num_feature_levels == 1
        self.two_stage = bool(kw.get('two_stage', False))
        in_dim = [in_dim] if isinstance(in_dim, int) else list(in_dim)
        assert in_dim[0] == hidden_dim, 'released configs: fpn_dim == hidden_dim, input_proj is empty (mask_transformer.py:72-77)'
        self.pe_layer = PositionEmbeddingSine(hidden_dim // 2, normalize=True)
        self.num_heads, self.num_layers, self.landscape_only = num_heads, dec_layers, landscape_only
        self.self_attn_layers = nn.ModuleList(SelfAttentionLayer(hidden_dim, num_heads) for _ in range(dec_layers))
        self.cross_attn_layers = nn.ModuleList(CrossAttentionLayer(hidden_dim, num_heads) for _ in range(dec_layers))
        self.ffn_layers = nn.ModuleList(FFNLayer(hidden_dim, ff_dim) for _ in range(dec_layers))
        self.decoder_norm = nn.LayerNorm(hidden_dim)
        self.num_queries = num_queries
        if not self.two_stage:                                      # learnt queries (mask_transformer.py:61-65); two-stage: selected from the keyframe tokens
            self.query_feat = nn.Embedding(num_queries, hidden_dim)
            self.query_embed = nn.Embedding(num_queries, hidden_dim)
        self.level_embed = nn.Embedding(1, hidden_dim)
        self.input_proj = nn.ModuleList([nn.Sequential()])
        self.lang_embed = nn.Linear(hidden_dim, lang_dim)
        self.cls_logit_scale = nn.Parameter(torch.ones([]))
        self.mask_embed = MLP(hidden_dim, hidden_dim, mask_dim, 3)

    def _pos(self, x, true_shape):
        # mask_transformer.py:106-119
        land_pe = self.pe_layer(x).flatten(2).permute(2, 0, 1)
        if not self.landscape_only:
            return land_pe
        port_pe = self.pe_layer(x.transpose(-2, -1)).flatten(2).permute(2, 0, 1)
        hh, ww = true_shape.T
        return torch.where((ww >= hh).to(x.device)[None, :, None], land_pe, port_pe)

    def forward(self, fpn_f, mask_feats, true_shape, cls_embeddings, max_bs=None, outdevice=None, multi_ar=False, **kw):
        groups_f = fpn_f[0] if multi_ar else [fpn_f[0]]           # one FPN level; list over aspect-ratio groups
        groups_s = true_shape if multi_ar else [true_shape]
        src, pos, sizes = [], [], []
        for f, ts in zip(groups_f, groups_s):                       # f [B,N,C,h,w]
            B, N = f.shape[:2]
            sizes.append(tuple(f.shape[-2:]))
            pos.append(self._pos(f[:, 0], ts[:, 0]).repeat(N, 1, 1))
            s = f.permute(0, 2, 1, 3, 4).flatten(-3).permute(2, 0, 1)
            src.append(s + self.level_embed.weight[0][None, None])
        src, pos = torch.cat(src, 0), torch.cat(pos, 0)
        bs = src.shape[1]
        if self.two_stage:                                          # :143-148
            out, qpos = self.query_selection(src, pos, cls_embeddings)
        else:
            qpos = self.query_embed.weight[:, None].repeat(1, bs, 1)
            out = self.query_feat.weight[:, None].repeat(1, bs, 1)
        cls, masks, amask = self.forward_prediction_heads(out, mask_feats, cls_embeddings, sizes, max_bs, outdevice, multi_ar)
        all_cls, all_masks = [cls], [masks]
        for i in range(self.num_layers):
            amask = amask.clone()
            amask[amask.all(-1)] = False                            # fully-blocked rows attend everywhere (:172)
            out = self.cross_attn_layers[i](out, src, amask, pos, qpos)
            out = self.self_attn_layers[i](out, qpos)
            out = self.ffn_layers[i](out)
            cls, masks, amask = self.forward_prediction_heads(out, mask_feats, cls_embeddings, sizes, max_bs, outdevice, multi_ar)
            all_cls.append(cls)
            all_masks.append(masks)
        return {'pred_logits': all_cls[-1], 'pred_masks': all_masks[-1],
                'aux_outputs': [{'pred_logits': a, 'pred_masks': b} for a, b in zip(all_cls[:-1], all_masks[:-1])],
                'out_queries': out.detach()}

    def query_selection(self, feats, pos, cls_embeddings):
        """two_stage (mask_transformer.py:85-104): the num_queries tokens whose best class logit is largest become the initial queries, their
        positional encodings the query positions.  feats, pos [NK, B, C] -> ([Q, B, C], [Q, B, C]), in descending order of that logit."""
        lang, _ = self.class_and_embed(feats, embed=False)                                        # [B, NK, L], unit norm (:91-96)
        logit = self.cls_logit_scale.exp() * lang @ cls_embeddings[None].transpose(1, 2)          # :97
        idx = torch.topk(logit.max(-1)[0], self.num_queries, dim=1)[1]                            # [B, Q] (:99-100)
        idx = idx.T[:, :, None].expand(-1, -1, feats.shape[-1])
        return torch.gather(feats, 0, idx), torch.gather(pos, 0, idx)                             # :101-103

    def class_and_embed(self, output, embed=True):
        """decoder_norm -> (unit-norm language embedding, mask embedding); mask_transformer.py:222-230."""
        d = self.decoder_norm(output).transpose(0, 1)
        lang = self.lang_embed(d)
        lang = lang / (lang.norm(dim=-1, keepdim=True) + 1e-7)
        return lang, (self.mask_embed(d) if embed else None)

    def forward_prediction_heads(self, output, mask_feats, cls_embeddings, attn_mask_target_size=None, max_bs=None,
                                 outdevice=None, multi_ar=False):
        groups = mask_feats if multi_ar else [mask_feats]
        lang, memb = self.class_and_embed(output)                   # [B,Q,L], [B,Q,C]
        cls = self.cls_logit_scale.exp() * lang @ cls_embeddings[None].transpose(1, 2)
        masks_out, am_out = [], []
        for g, mf in enumerate(groups):                             # mf [B,N,C,H,W]
            m = torch.einsum('bqc,bnchw->bnqhw', memb, mf)
            masks_out.append(m)
            if attn_mask_target_size is not None:
                B, N, Q = m.shape[:3]
                a = F.interpolate(m.flatten(0, 1), size=tuple(attn_mask_target_size[g]), mode='bilinear', align_corners=False)
                am_out.append(a.view(B, N, Q, *a.shape[-2:]).permute(0, 2, 1, 3, 4).flatten(-3))
        amask = None
        if am_out:
            a = torch.cat(am_out, dim=2)
            amask = (a.sigmoid()[:, None].repeat(1, self.num_heads, 1, 1).flatten(0, 1) < 0.5)
        return cls, (masks_out if multi_ar else masks_out[0]), amask


class TextEncoder(nn.Module):
    """Fixed-vocabulary branch only (text_encoder.py:94-101): lookup + L2 normalise."""

    def __init__(self, model_name='siglip', out_dim=768, fixed_vocab=True):
        super().__init__()
        self.embed_dim = {'siglip': 768, 'siglip2': 768, 'clip': 512}[model_name]
        self.fixed_vocab = True
        self.class_embeddings = {}

    def forward(self, classes):
        e = torch.stack([self.class_embeddings[c] for c in classes])
        return e / e.norm(dim=-1, keepdim=True)


class PanopticDecoder(nn.Module):
    def __init__(self, input_mixer=None, upscaler=None, fpn_dim=(768,), hidden_dim=768, mask_dim=256, ff_dim=2048,
                 num_queries=200, num_heads=8, dec_layers=6, text_encoder='siglip', fixed_vocab=True, label_mode='sigmoid',
                 two_stage=False, landscape_only=True, deep_supervision=True):
        super().__init__()
        assert upscaler is not None and label_mode in ('sigmoid', 'softmax')
        self.input_mixer = input_mixer
        self.upscaler = upscaler
        self.landscape_only = landscape_only
        self.text_encoder = TextEncoder(text_encoder, out_dim=hidden_dim)
        self.label_mode = label_mode
        if label_mode == 'softmax':                                 # one more class row, "no object", learnt and NOT normalised (panoptic_decoder.py:30-31,66-67)
            self.nocls_token = nn.Parameter(torch.randn(self.text_encoder.embed_dim))
        self.mask_transformer = MaskTransformer(list(fpn_dim), hidden_dim, ff_dim, mask_dim, num_queries, num_heads, dec_layers,
                                                lang_dim=self.text_encoder.embed_dim, num_feature_levels=len(fpn_dim),
                                                landscape_only=landscape_only, two_stage=two_stage)

    def class_matrix(self, classes):
        e = self.text_encoder(classes)
        return torch.cat([e, self.nocls_token[None]], dim=0) if self.label_mode == 'softmax' else e

    def features(self, cat_feats, imgs, pos, true_shape, max_bs=None):
        """mixer + upscaler over a [B,n,...] stack in chunks of max_bs (panoptic_decoder.py:50-62)."""
        B, n = cat_feats.shape[:2]
        cf, im, ps, ts = cat_feats.flatten(0, 1), imgs.flatten(0, 1), pos.flatten(0, 1), true_shape.flatten(0, 1)
        mfs, fps = [], []
        for s, e in _chunks(B * n, max_bs):
            x = cf[s:e] if self.input_mixer is None else self.input_mixer(cf[s:e], ps[s:e])
            fpn, mf = call_in_landscape(self.upscaler, (x, im[s:e]), ts[s:e], dims=(2, 3), activate=self.landscape_only)
            mfs.append(mf)
            fps.append(fpn[0])
        return torch.cat(fps).unflatten(0, (B, n)), torch.cat(mfs).unflatten(0, (B, n))

    def forward(self, in_feats, in_imgs, pos, true_shape, classes, max_bs=None, outdevice=None, memory_queries=None, multi_ar=False):
        if multi_ar:
            cat = [torch.cat(t, dim=-1) for t in zip(*in_feats)]
            res = [self.features(c, im, p, ts, max_bs) for c, im, p, ts in zip(cat, in_imgs, pos, true_shape)]
            fpn, mask_f = [[r[0] for r in res]], [r[1] for r in res]
        else:
            f, mask_f = self.features(torch.cat(in_feats, dim=-1), in_imgs, pos, true_shape, max_bs)
            fpn = [f]
        cls_emb = self.class_matrix(classes)
        if memory_queries is None:
            return self.mask_transformer(fpn, mask_f, true_shape, cls_emb, max_bs=max_bs, outdevice=outdevice, multi_ar=multi_ar)
        cls, masks, _ = self.mask_transformer.forward_prediction_heads(memory_queries, mask_f, cls_emb, multi_ar=multi_ar)
        return {'pred_logits': cls, 'pred_masks': masks}
