#!/usr/bin/env python
"""frames/s of the API entry exactly as the reference's demo calls it (tools/demo_panst3r.py:232-233: max_bs=1, outdevice='cpu'), incl. stacking the inputs, the
finite check and the device -> host copy of every pointmap and mask tensor (2.2 GB at 50 views of 384x512):   python tools/api_bench.py [calls]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panst3r_amd.panst3r import CONFIG_V2, build_from_config                  # noqa: E402
from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings   # noqa: E402

V, K, H, W = int(os.environ.get('PST_V', 50)), int(os.environ.get('PST_K', 16)), 384, 512
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda:0')
model = build_from_config(CONFIG_V2).eval()
fill_module_(model, seed=1)
names, emb = synth_class_embeddings(100)
model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
model.to(dev)
imgs = [synth_image(i, H, W).to(dev) for i in range(V)]
ts = torch.tensor([[H, W]] * V)
ONLY = os.environ.get('PST_API_ONLY')            # e.g. "fp16,False,cpu": one configuration (profiler runs)
for amp in ('fp16', False):
    for graphs in (False, True):
        for out in ('cpu', None):
            if ONLY and ONLY != '%s,%s,%s' % (amp, graphs, out):
                continue
            model.clear_runners()
            for _ in range(2 if graphs else 1):
                r = model.forward_inference_multi_ar(imgs, ts, names, num_keyframes=K, max_bs=1, outdevice=out, amp=amp, cache_graphs=graphs)
            del r
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(N):
                r = model.forward_inference_multi_ar(imgs, ts, names, num_keyframes=K, max_bs=1, outdevice=out, amp=amp, cache_graphs=graphs)
                del r
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / N
            print('amp=%-5s cache_graphs=%-5s outdevice=%-4s %8.1f ms per call %7.2f frames/s' % (amp, graphs, out, 1e3 * dt, V / dt), flush=True)
