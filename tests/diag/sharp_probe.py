"""How do the 'sharp' weight sets (QK rows scaled: attention logits x s^2) behave?  tiny + full-size encoder-only probes, f16 vs oracle."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import tiny
DEV = 'cuda:0'
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
H, W = 64, 96
imgs = tiny.images(3, H, W)
ts = torch.tensor([[H, W]] * 3)
for sharp in (1.0, 1.41, 2.0):
    o = tiny.build(tiny.OracleNS, 'v2', sharp=sharp)
    h = tiny.build(tiny.hip_ns(), 'v2', sharp=sharp).to(DEV)
    with torch.no_grad():
        xo, po = o.must3r_encoder(torch.stack(imgs), ts)
        xh, ph = h.must3r_encoder(torch.stack(imgs).to(DEV), ts)
        do = o.dino_encoder(torch.stack(imgs), ts); dh = h.dino_encoder(torch.stack(imgs).to(DEV), ts)
        pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=2)
        pm_h, pan_h = h.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, tiny.NAMES, num_keyframes=2)
    print('tiny sharp=%.2f: enc %.2e dino %.2e pointmaps %.2e masks %.2e queries %.2e' % (sharp, rel(xh, xo), rel(dh, do), max(rel(a, b) for a, b in zip(pm_h, pm_o)),
          max(rel(a, b) for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks'])), rel(pan_h['out_queries'], pan_o['out_queries'])))
# full-size encoder only (24 layers), one view
from panst3r_amd.model import Dust3rEncoder
from oracle.must3r import Dust3rEncoder as OEnc
from panst3r_amd.synthetic import fill_module_, synth_image
img = synth_image(0, 384, 512)[None]
tsf = torch.tensor([[384, 512]])
for sharp in (1.0, 1.19, 1.41, 1.68, 2.0, 8 ** 0.5):
    eo = fill_module_(OEnc(img_size=[512, 512], patch_embed='PatchEmbedDust3R').eval(), seed=1, sharp=sharp, prefix='must3r_encoder.')
    eh = fill_module_(Dust3rEncoder(img_size=[512, 512], patch_embed='PatchEmbedDust3R').eval(), seed=1, sharp=sharp, prefix='must3r_encoder.').to(DEV)
    with torch.no_grad():
        a, _ = eo(img, tsf); b, _ = eh(img.to(DEV), tsf)
    print('full-size encoder sharp=%.2f: rel %.2e  max|x| %.1f' % (sharp, rel(b, a), float(a.abs().max())))

# full-dimension MaskTransformer alone (6 layers): HIP vs oracle, free running, at several sharpness levels
from panst3r_amd.model import MaskTransformer
from panst3r_amd.model.common import adt, precision
from oracle.panoptic import MaskTransformer as OMT
import numpy as np
def rnd(seed, *shape):
    g = np.random.Generator(np.random.PCG64(seed)); return torch.from_numpy(g.standard_normal(shape).astype(np.float32))
n, hh, ww = 2, 6, 8
fpn, mf = rnd(215, 1, n, 768, hh, ww), rnd(225, 1, n, 384, 8 * hh, 8 * ww)
tsm = torch.tensor([[[16 * hh, 16 * ww]] * n])
cls = torch.nn.functional.normalize(rnd(235, 20, 768), dim=-1)
for sharp in (1.0, 1.41, 2.0, 8 ** 0.5, 4.0, 8.0):
    mo = fill_module_(OMT([768], 768, 2048, 384, 200, 8, 6, lang_dim=768, num_feature_levels=1, landscape_only=True).eval(), seed=15, sharp=sharp)
    mh = fill_module_(MaskTransformer([768], 768, 2048, 384, 200, 8, 6, lang_dim=768, num_feature_levels=1, landscape_only=True).eval(), seed=15, sharp=sharp).to(DEV)
    with torch.no_grad(), precision('fp16'):
        out = mo([fpn], mf, tsm, cls)
        tok = fpn[0].flatten(2).permute(0, 2, 1).reshape(n * hh * ww, 768).to(adt()).to(DEV).contiguous()
        mfp = mf[0].permute(0, 2, 3, 1).to(adt()).to(DEV).contiguous()
        outq, hs = mh.decode_tokens(tok, mh.attn_feats(mfp, (hh, ww)), [(hh, ww)] * n, cls.to(adt()).to(DEV).contiguous(), [False] * n)
    dq = (outq.cpu().double() - out['out_queries'].reshape(200, 768).double()).norm(dim=-1) / out['out_queries'].reshape(200, 768).double().norm(dim=-1)
    print('full-dim MaskTransformer sharp=%.2f: queries rel %.2e (median per query %.2e, max %.2e)' % (sharp, rel(outq, out['out_queries'].reshape(200, 768)), float(dq.median()), float(dq.max())))
