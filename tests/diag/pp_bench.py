#!/usr/bin/env python
"""Time the GPU panoptic post-processing (panst3r_amd.engine.panoptic_inference_v2) on the bench workload's outputs:
200 queries x V views at 384x512 (mask logits [1,200,192,256] per view), synthetic blob masks, 100 classes.
Prints one JSON line (ms per scene, per-kernel HIP-event times are taken with rocprofv3 separately)."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from panst3r_amd.engine import panoptic_inference_v2

V = int(sys.argv[1]) if len(sys.argv) > 1 else 50
Q, NCLS, H, W = 200, 100, 384, 512
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(1)
logits = torch.randn(1, Q, NCLS, generator=g, device=dev) * 2
masks = []
rng = np.random.Generator(np.random.PCG64(3))
boxes = [(int(rng.integers(0, 150)), int(rng.integers(0, 200)), int(rng.integers(8, 60)), int(rng.integers(8, 80))) for _ in range(Q)]
for v in range(V):
    m = torch.randn(1, Q, H // 2, W // 2, generator=g, device=dev) * 1.5 - 3.0
    for q, (y, x, hh, ww) in enumerate(boxes):
        m[0, q, y:y + hh, x:x + ww] += 6.0
    masks.append(m)
size = np.array([[H, W]] * V)
for _ in range(2):
    res = panoptic_inference_v2(logits, masks, size, multi_ar=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 5
for _ in range(N):
    res = panoptic_inference_v2(logits, masks, size, multi_ar=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / N * 1e3
out = {'workload': 'panoptic_inference_v2: %d queries x %d views at %dx%d, niters=2' % (Q, V, H, W), 'ms_per_scene': round(ms, 3),
       'segments': len(res[0]['segments_info']), 'views_per_s': round(V / ms * 1e3, 1)}
if '--cpu' in sys.argv:      # the oracle on 2 views (the reference's CPU formulation materialises [Q,V,H,W] fp32)
    from oracle.postprocess import panoptic_inference_v2 as ref_fn
    t0 = time.perf_counter()
    ref_fn(logits.cpu(), [m.cpu() for m in masks[:2]], size[:2])
    out['cpu_oracle_ms_per_view'] = round((time.perf_counter() - t0) / 2 * 1e3, 1)
print(json.dumps(out))
