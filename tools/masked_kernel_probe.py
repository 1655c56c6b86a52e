#!/usr/bin/env python
"""Which kernel of the memory build is slow on a small CU-masked stream?  One graph of 200 dependent launches per kernel kind, on masked streams of several sizes:
us per launch of the 768-row GEMMs (64 x 64 tiles), the build's split-K cross-attention + combine, LayerNorm."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from panst3r_amd.scene import HipBackend
dev = torch.device('cuda:0')
b = HipBackend.__new__(HipBackend)
T, D, H, hd = 768, 768, 12, 64
x = (torch.randn(T, D, device=dev) * 0.5).half()
w768 = (torch.randn(D, D, device=dev) * 0.03).half()
w3072 = (torch.randn(3072, D, device=dev) * 0.03).half()
w2 = (torch.randn(D, 3072, device=dev) * 0.03).half()
o768 = torch.empty(T, D, device=dev, dtype=torch.float16)
o3072 = torch.empty(T, 3072, device=dev, dtype=torch.float16)
res = torch.zeros(T, D, device=dev)
Nk = 9216
kb = (torch.randn(Nk, D, device=dev) * 0.5).half()
vt = (torch.randn(D, Nk + 8, device=dev) * 0.5).half()
oa = torch.empty(T, D, device=dev, dtype=torch.float16)
g32, b32 = torch.ones(D, device=dev), torch.zeros(D, device=dev)
KINDS = {
    'gemm 768x768x768': lambda: hip.gemm(x, w768, o768),
    'gemm 768x3072x768 gelu': lambda: hip.gemm(x, w3072, o3072, act='gelu'),
    'gemm 768x768x3072 +res': lambda: hip.gemm(o3072, w2, res, res=res),
    'attention 768 q x 9216 keys (split-K)': lambda: hip.attention(x, kb, vt, oa, 1, H, T, Nk, hd, (0, hd, D), (0, hd, D), (0, hd * vt.stride(0), vt.stride(0)), (0, hd, D)),
    'layernorm 768 rows': lambda: hip.layernorm(res, g32, b32, o768, 1e-6),
}


def bench(fn, stream, cus):
    hip.tune(hip.TUNE_CUS, cus)
    try:
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(200):
                fn()
    finally:
        hip.tune(hip.TUNE_CUS, 0)
    cur = torch.cuda.current_stream()

    def go():
        if stream is None:
            g.replay(); return
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            g.replay()
        cur.wait_stream(stream)
    go(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e6 / 200


sizes = [None, 192, 128, 96, 64, 48]
print('%-40s' % 'us per launch' + ''.join('%10s' % ('all' if c is None else '%d CUs' % c) for c in sizes))
for name, fn in KINDS.items():
    row = []
    for c in sizes:
        s = None if c is None else b.masked_streams(dev, c)[0]
        row.append(bench(fn, s, 0 if c is None else c))
    print('%-40s' % name + ''.join('%10.1f' % v for v in row), flush=True)
