#!/usr/bin/env python
"""GEMM cases of the scene's big shapes with realistic epilogue arguments (LayerNorm fold consumer / producer outputs, fused RoPE, GELU,
fp32 residual stream, transposed store), shared by the GEMM measurement tools (tools/dispatch_bench.py, pp_bench.py, ...).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip

dev = torch.device('cuda:0')
DT = torch.float16


def burst(fn, reps=10, bursts=5):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(bursts):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e-3)
    return sorted(ts)[len(ts) // 2]


def case(M, N, K, kind):
    """kind: 'fc1' fold consumer + GELU | 'qk' fold consumer + RoPE | 'q' fold consumer plain | 'vt' fold consumer, transposed | 'res' fp32 residual stream +
    producer outputs"""
    g = torch.Generator(device='cpu').manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(DT).to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(DT).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    kw = dict(bias=b)
    if kind in ('fc1', 'qk', 'q', 'vt'):
        st = torch.empty(M, K // 64, 2, device=dev)
        x = a.float()
        xb = torch.empty(M, K, dtype=DT, device=dev)
        hip.rowstats(x, xb, st)
        a = xb
        kw['ln'] = (st, w.float().sum(1).contiguous(), 1e-6)
    if kind == 'plain':                     # 16-bit output, bias only (no LayerNorm fold): the bare main loop
        out = torch.empty(M, N, dtype=DT, device=dev)
    elif kind == 'fc1':
        out = torch.empty(M, N, dtype=DT, device=dev)
        kw['act'] = 'gelu'
    elif kind == 'qk':
        out = torch.empty(M, N, dtype=DT, device=dev)
        T = 768
        ys, xs = torch.meshgrid(torch.arange(24), torch.arange(32), indexing='ij')
        pos = torch.stack([ys, xs], -1).reshape(T, 2).to(torch.int32).repeat(M // T + 1, 1)[:M].contiguous().to(dev)
        kw['rope'] = (pos, hip.rope_table(32, 64, 100.0, dev))
        kw['gamma'] = torch.ones(N, device=dev)
    elif kind == 'q':
        out = torch.empty(M, N, dtype=DT, device=dev)
    elif kind == 'vt':
        out = torch.zeros(N, M + 8, dtype=DT, device=dev)
        kw['trans_out'] = True
    else:
        out = torch.randn(M, N, generator=g).to(dev)
        kw['res'] = out
        kw['xcopy'] = torch.empty(M, N, dtype=DT, device=dev)
        kw['stats_out'] = torch.empty(M, N // 64, 2, device=dev)
    return a, w, out, kw
