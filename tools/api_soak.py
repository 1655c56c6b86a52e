#!/usr/bin/env python
"""Bit-for-bit soak of the streamed API call (outdevice='cpu': outputs copied to pinned memory on a copy stream WHILE the scene computes) against the same call
without an output device, at the bench scene's size:   python tools/api_soak.py [calls] [graphs 0|1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panst3r_amd.panst3r import CONFIG_V2, build_from_config                  # noqa: E402
from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings   # noqa: E402

V, K, H, W = int(os.environ.get('PST_V', 50)), int(os.environ.get('PST_K', 16)), 384, 512
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
graphs = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
dev = torch.device('cuda:0')
model = build_from_config(CONFIG_V2).eval()
fill_module_(model, seed=1)
names, emb = synth_class_embeddings(100)
model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
model.to(dev)
imgs = [synth_image(i, H, W).to(dev) for i in range(V)]
ts = torch.tensor([[H, W]] * V)
kw = dict(num_keyframes=K, max_bs=1, amp='fp16', cache_graphs=graphs)
pm_d, pan_d = model.forward_inference_multi_ar(imgs, ts, names, **kw)
ref_pm = [p.cpu() for p in pm_d]
ref_mk = [m.cpu() for m in pan_d['pred_masks']]
ref_q, ref_l = pan_d['out_queries'].cpu(), pan_d['pred_logits'].cpu()
del pm_d, pan_d
bad = 0
for it in range(N):
    pm, pan = model.forward_inference_multi_ar(imgs, ts, names, outdevice='cpu', **kw)
    dv = [i for i in range(V) if not (torch.equal(pm[i], ref_pm[i]) and torch.equal(pan['pred_masks'][i], ref_mk[i]))]
    ok = not dv and torch.equal(pan['out_queries'].cpu(), ref_q) and torch.equal(pan['pred_logits'].cpu(), ref_l)
    bad += not ok
    if not ok:
        print('call %d deviates: views %s' % (it, dv[:10]), flush=True)
print('streamed API call (cache_graphs=%s): %d of %d calls deviate from the plain call' % (graphs, bad, N), flush=True)
