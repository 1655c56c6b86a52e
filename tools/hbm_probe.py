#!/usr/bin/env python
"""HBM streaming ceiling of the box as plain PyTorch sees it (copy / read-reduce / fill of 1 GiB), next to this library's streaming
kernels (LayerNorm, rowstats, GroupNorm apply) on scene-sized tensors: how far the HBM-bound kernels are from what the memory system gives."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.kbench import timeit
dev = 'cuda:0'
n = 1 << 28
x = torch.randn(n, device=dev); y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x)); print('torch copy fp32 1 GiB -> 1 GiB : %.2f TB/s' % (2 * 4 * n / t / 1e12))
t = timeit(lambda: x.sum()); print('torch sum  fp32 1 GiB          : %.2f TB/s' % (4 * n / t / 1e12))
t = timeit(lambda: y.fill_(1.0)); print('torch fill fp32 1 GiB          : %.2f TB/s' % (4 * n / t / 1e12))
h = x.half(); g = torch.empty_like(h)
t = timeit(lambda: g.copy_(h)); print('torch copy f16 0.5 GiB         : %.2f TB/s' % (2 * 2 * n / t / 1e12))
t = timeit(lambda: torch.add(x, 1.0, out=y)); print('torch add  fp32 (r+w)          : %.2f TB/s' % (2 * 4 * n / t / 1e12))
from panst3r_amd import hip
M, D = 38800, 1024
a = torch.randn(M, D, device=dev); o = torch.empty(M, D, dtype=torch.float16, device=dev)
gm, bt = torch.ones(D, device=dev), torch.zeros(D, device=dev)
t = timeit(lambda: hip.layernorm(a, gm, bt, o, 1e-6)); print('pst layernorm 38800x1024 f32->f16: %.2f TB/s' % (M * D * 6 / t / 1e12))
st = torch.empty(M, D // 64, 2, device=dev)
t = timeit(lambda: hip.rowstats(a, o, st)); print('pst rowstats  38800x1024         : %.2f TB/s' % (M * D * 6 / t / 1e12))
