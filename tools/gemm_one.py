#!/usr/bin/env python
"""A few launches of one GEMM shape of the headline scene (driver for tools/sq_profile.sh / timing):
    gemm_one.py M N K [gelu] [res] [f32] [fold] [time]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
M, N, K = map(int, sys.argv[1:4])
fl = sys.argv[4:]
dev, dt = 'cuda:0', torch.float16
a = torch.randn(M, K, device=dev).to(dt)
w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
b = torch.randn(N, device=dev)
out = torch.zeros(M, N, dtype=torch.float32 if ('f32' in fl or ('res' in fl and 'res16' not in fl)) else dt, device=dev)
kern = ([int(x[1:]) for x in fl if x[0] == 'k' and x[1:].isdigit()] + [0])[0]
if 'trans' in fl:
    out = torch.zeros(N, (M + 7) // 8 * 8 + 8, dtype=dt, device=dev)
kw = dict(kernel=kern, trans_out='trans' in fl, bias=b, act='gelu' if 'gelu' in fl else None, res=out if ('res' in fl or 'res16' in fl) else None)
if 'fold' in fl and 'res' in fl:          # producer side of the LayerNorm fold
    kw.update(xcopy=torch.zeros(M, N, dtype=dt, device=dev), stats_out=torch.zeros(M, N // 64, 2, dtype=torch.float32, device=dev))
elif 'fold' in fl:                        # consumer side
    st = torch.zeros(M, K // 64, 2, dtype=torch.float32, device=dev)
    st[..., 1] = 64.0
    kw.update(ln=(st, w.float().sum(1).contiguous(), 1e-6))
if 'rope' in fl:                         # fused RoPE-2D store (q|k projection): 24 x 32 token grid per view
    ys, xs = torch.meshgrid(torch.arange(24), torch.arange(32), indexing='ij')
    pos = torch.stack([ys, xs], -1).reshape(768, 2).to(torch.int32).repeat((M + 767) // 768, 1)[:M].contiguous().to(dev)
    kw.update(rope=(pos, hip.rope_table(32, 64, 100.0, dev)))
f = lambda: hip.gemm(a, w, out, **kw)
if 'time' in fl:
    from tools.kbench import timeit
    t = timeit(f)
    print('%s %s: %.1f us  %.1f TF' % ((M, N, K), fl, t * 1e6, 2.0 * M * N * K / t / 1e12))
else:
    for _ in range(4):
        f()
    torch.cuda.synchronize()
