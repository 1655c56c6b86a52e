#!/usr/bin/env python
"""Floor of a dependent kernel chain inside a replayed HIP graph on this stack: microseconds per node for a trivial kernel, for this library's
smallest kernels and for the 768-row GEMMs of the memory build -- how much of the build's ~11 us per GEMM is the kernel boundary."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
dev, dt = 'cuda:0', torch.float16
N = 500


def chain(fn, tag):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(N):
                fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            g.replay()
        b.record(); torch.cuda.synchronize()
    print('%-58s %6.2f us per node' % (tag, a.elapsed_time(b) * 1e3 / (5 * N)), flush=True)


x = torch.zeros(64, device=dev)
chain(lambda: x.add_(1.0), 'torch add_ on 64 floats (1 block)')
big = torch.zeros(768 * 768, device=dev)
chain(lambda: big.add_(1.0), 'torch add_ on 768x768 floats (2.4 MB)')
a768 = torch.randn(768, 768, device=dev).to(dt); w = (torch.randn(768, 768, device=dev) * 768 ** -0.5).to(dt); bias = torch.randn(768, device=dev)
o16 = torch.empty(768, 768, dtype=dt, device=dev)
chain(lambda: hip.gemm(a768, w, o16, bias=bias), 'pst gemm 768x768x768 16-bit out')
y = torch.zeros(768, 768, device=dev); xc = torch.empty(768, 768, dtype=dt, device=dev); st = torch.empty(768, 12, 2, device=dev)
chain(lambda: hip.gemm(a768, w, y, bias=bias, res=y, xcopy=xc, stats_out=st), 'pst gemm 768x768x768 residual + fold producer')
w4 = (torch.randn(768, 3072, device=dev) * 3072 ** -0.5).to(dt); a4 = torch.randn(768, 3072, device=dev).to(dt)
chain(lambda: hip.gemm(a4, w4, y, bias=bias, res=y, xcopy=xc, stats_out=st), 'pst gemm 768x768x3072 residual + fold producer')
w1 = (torch.randn(3072, 768, device=dev) * 768 ** -0.5).to(dt); h = torch.empty(768, 3072, dtype=dt, device=dev); b3 = torch.randn(3072, device=dev)
chain(lambda: hip.gemm(a768, w1, h, bias=b3, act='gelu'), 'pst gemm 768x3072x768 gelu')
g1, b1 = torch.ones(768, device=dev), torch.zeros(768, device=dev)
chain(lambda: hip.layernorm(y, g1, b1, o16, 1e-6), 'pst layernorm 768x768')
