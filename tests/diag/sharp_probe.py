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
for sharp in (1.0, 2.0, 8 ** 0.5, 4.0, 8.0):
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
for sharp in (1.0, 8 ** 0.5, 8.0):
    eo = fill_module_(OEnc(img_size=[512, 512], patch_embed='PatchEmbedDust3R').eval(), seed=1, sharp=sharp, prefix='must3r_encoder.')
    eh = fill_module_(Dust3rEncoder(img_size=[512, 512], patch_embed='PatchEmbedDust3R').eval(), seed=1, sharp=sharp, prefix='must3r_encoder.').to(DEV)
    with torch.no_grad():
        a, _ = eo(img, tsf); b, _ = eh(img.to(DEV), tsf)
    print('full-size encoder sharp=%.2f: rel %.2e  max|x| %.1f' % (sharp, rel(b, a), float(a.abs().max())))
