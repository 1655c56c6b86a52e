#!/usr/bin/env python
"""Split-K choice of the memory build's cross-attention (768 queries x n*768 memory keys, 12 heads): time per launch (attention + combine)
for nsplit = 1..10 and the current hip.auto_nsplit choice."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from tools.kbench import timeit
dev, dt = 'cuda:0', torch.float16
B, H, Nq, hd = 1, 12, 768, 64
D = H * hd
for nkf in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 15):
    Nk = nkf * 768
    q = (torch.randn(Nq, D, device=dev) * hd ** -0.5 * hip.LOG2E).to(dt); k = torch.randn(Nk + 8, D, device=dev).to(dt); vt = torch.randn(D, Nk + 8, device=dev).to(dt)
    o = torch.empty(Nq, D, dtype=dt, device=dev)
    ws = torch.empty(32 * B * H * Nq * (hd + 2), dtype=torch.float32, device=dev)
    res = []
    for ns in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16):
        if ns > 1 and Nk // ns < 128:
            continue
        t = timeit(lambda: hip.attention(q, k, vt, o, B, H, Nq, Nk, hd, (0, hd, D), (0, hd, D), (0, hd * vt.stride(0), vt.stride(0)), (0, hd, D), nsplit=ns, ws=ws, prescaled=True), n=50)
        res.append('%d:%5.1f' % (ns, t * 1e6))
    print('Nk=%5d auto=%d  us per launch  %s' % (Nk, hip.auto_nsplit(B, H, Nq, Nk), '  '.join(res)), flush=True)
