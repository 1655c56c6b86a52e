#!/usr/bin/env python
"""few-query / many-key attention of the memory build (768 queries x 12 heads against 768 ... 11 520 memory keys): key-range splits and block size"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from tools.dispatch_bench import compare
hip.lib()
dev, dt = 'cuda:0', torch.float16
B, H, Nq, hd = 1, 12, 768, 64
D = H * hd
tot = {}
for Nk in (768, 1536, 3072, 6144, 9216, 11520, 24576):
    q = (torch.randn(B * Nq, D, device=dev) * (hd ** -0.5 * hip.LOG2E)).to(dt)
    k = torch.randn(Nk + 8, D, device=dev).to(dt)
    vt = torch.randn(D, Nk + 8, device=dev).to(dt)
    o = torch.zeros(B * Nq, D, dtype=dt, device=dev)
    ws = torch.empty(64 * B * H * Nq * (hd + 2), dtype=torch.float32, device=dev)
    cand = [None, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16]
    fns = [(lambda ns=ns: hip.attention(q, k, vt, o, B, H, Nq, Nk, hd, (Nq * D, hd, D), (Nk * D, hd, D), (Nk, hd * vt.stride(0), vt.stride(0)), (Nq * D, hd, D),
                                        prescaled=True, nsplit=ns, ws=ws)) for ns in cand if ns is None or Nk // max(ns, 1) >= 256]
    names = [ns for ns in cand if ns is None or Nk // max(ns, 1) >= 256]
    ts = compare(fns, rounds=5)
    print('Nk %5d (auto = %d): ' % (Nk, hip.auto_nsplit(B, H, Nq, Nk)) + '  '.join('%s:%.1f' % ('auto' if n is None else n, t) for n, t in zip(names, ts)), flush=True)
