#!/usr/bin/env python
"""Where does the fp32 mode's full-size mask-logit residue (2e-4) come from?  GPU box:  python tests/diag/fp32_bisect.py
Full-size v2 panoptic decoder, fp32 mode (amp=False) against the fp32 oracle on the SAME fp32 inputs, stage by stage: mixer tokens, guidance
features (Fourier + GN + convs), FPN tokens, mask features, then the mask head given the ORACLE's mask embedding and the HIP features and vice versa."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.nn.functional as F

import bench
from oracle.pipeline import build as build_oracle
from panst3r_amd.model.common import precision
from panst3r_amd.panst3r import CONFIG_V2, build_from_config
from panst3r_amd.synthetic import fill_module_, synth_image

dev = torch.device('cuda:0')
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
torch.set_num_threads(bench.usable_cores())
n, H, W = 2, 384, 512
h, w = H // 16, W // 16
T = h * w
model = build_from_config(CONFIG_V2).eval()
fill_module_(model, seed=1)
state = {k: v.clone() for k, v in model.state_dict().items()}
o = build_oracle('v2')
o.load_state_dict(state, strict=True)
model.to(dev)
g = torch.Generator().manual_seed(11)
cat = torch.randn(1, n, T, 2816, generator=g)
imgs = torch.stack([synth_image(i, H, W) for i in range(n)])
ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
pos = torch.stack([ys, xs], -1).reshape(1, 1, T, 2).expand(1, n, -1, -1).contiguous()
ts = torch.tensor([[[H, W]] * n])
out = {}
taps = {}
pdo, pdh = o.panoptic_decoder, model.panoptic_decoder
hooks = [pdo.input_mixer.register_forward_hook(lambda m, i, r: taps.__setitem__('mixer', r)),
         pdo.upscaler.fourier_feat.register_forward_hook(lambda m, i, r: taps.__setitem__('fourier', r)),
         pdo.upscaler.fourier_feat[0].register_forward_hook(lambda m, i, r: taps.__setitem__('minmax', r)),
         pdo.upscaler.first_conv[0].register_forward_hook(lambda m, i, r: taps.__setitem__('gn0', r)),
         pdo.upscaler.first_conv[1].register_forward_hook(lambda m, i, r: taps.__setitem__('conv1', r)),
         pdo.upscaler.first_conv.register_forward_hook(lambda m, i, r: taps.__setitem__('guidance', r))]
with torch.no_grad():
    chunks = [pdo.features(cat[:, i:i + 1], imgs[None, i:i + 1], pos[:, i:i + 1], ts[:, i:i + 1], max_bs=1) for i in range(1)]      # view 0 (hooks keep the last call)
    fpn_o, mf_o = chunks[0]
    with precision(False):
        c32 = cat[0, :1].reshape(T, -1).float().to(dev).contiguous()
        im = imgs[:1].to(dev)
        x = torch.empty(T, pdh.input_mixer.hidden_dim, dtype=torch.float32, device=dev)
        pdh.input_mixer.mix_tokens(c32, 1, h, w, x)
        out['mixer tokens'] = rel(x.reshape(1, T, -1), taps['mixer'])
        gd = pdh.upscaler.guidance_tokens(im, h, w)                      # [P, C] pixel-major
        gref = taps['guidance'][0].flatten(1).T                          # [P, C]
        out['guidance (GN0 -> conv -> GN -> ReLU -> conv -> GN -> ReLU)'] = rel(gd, gref)
        # the first stage alone: Fourier features + GN0
        from panst3r_amd import hip
        up = pdh.upscaler
        pk = up.packed(dev)
        P = (H // 2) * (W // 2)
        st0 = hip.stats_buffer(1, 1, dev)
        g0 = torch.empty(P, pk['c0'], dtype=torch.float32, device=dev)
        scratch = torch.empty(3 * P + 6 + 16, dtype=torch.float32, device=dev)
        hip.loftup_guidance_gn(im.contiguous(), pk['ff_bias'], pk['gn0'][0], pk['gn0'][1], pk['gn0'][2], scratch, st0, g0, up.n_freqs)
        r0 = taps['gn0'][0].flatten(1).T
        out['fourier + GN0'] = rel(g0[:, :r0.shape[1]], r0)
        d = (g0[:, :r0.shape[1]].cpu().double() - r0.double())
        per_ch = d.norm(dim=0) / r0.double().norm(dim=0)
        nf = up.n_freqs
        out['fourier + GN0 per frequency (sin rows, max over the 5 inputs)'] = [float(per_ch[f * 5:(f + 1) * 5].max()) for f in range(nf)]
        out['fourier + GN0 per input d (max over frequencies)'] = [float(max(per_ch[f * 5 + dd] for f in range(nf))) for dd in range(5)]
        # the 2x2 mean + min-max scaling, bit for bit?
        img2 = scratch[:3 * P].reshape(3, H // 2, W // 2).cpu()
        ref2 = F.interpolate(imgs[:1], scale_factor=0.5, mode='bilinear', align_corners=False)[0]
        out['2x2 mean vs torch bilinear: max abs diff / fraction of pixels that differ'] = [float((img2 - ref2).abs().max()), float((img2 != ref2).float().mean())]
        fpn_h, mf_h = pdh.features_tokens(c32, im, 1, h, w)
        out['fpn tokens'] = rel(fpn_h.reshape(h, w, -1).permute(2, 0, 1), fpn_o[0, 0])
        out['mask features'] = rel(mf_h[0].permute(2, 0, 1), mf_o[0, 0])
        # mask head: a fixed embedding against both feature sets
        E = torch.randn(200, mf_h.shape[-1], generator=g)
        mt = pdh.mask_transformer
        m_h = mt.masks_for(E.to(dev), mf_h[0]).cpu()
        m_o = torch.einsum('qc,chw->qhw', E, mf_o[0, 0])
        m_x = torch.einsum('qc,chw->qhw', E.double(), mf_h[0].permute(2, 0, 1).cpu().double())
        out['mask logits: HIP head on HIP features vs oracle'] = rel(m_h, m_o)
        out['mask logits: fp64 einsum on HIP features vs oracle (= the feature error seen through the dot product)'] = rel(m_x, m_o)
        out['mask logits: HIP head vs fp64 einsum on the SAME HIP features (= the head alone)'] = rel(m_h, m_x)
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'fp32_bisect.json'), 'w'), indent=1)
