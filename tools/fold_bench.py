#!/usr/bin/env python
"""What does the LayerNorm fold cost / save per ViT block?  A/B of the residual GEMM with and without the producer outputs, of the
consumer GEMM with and without the epilogue fold, and the stand-alone LayerNorm they replace (M = views*768 rows, D = 1024)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip

dev = 'cuda:0'
DT = torch.float16


def timeit(fn, n=30, w=5):
    for _ in range(w):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3          # us


def main():
    for V in (16, 34):
        M, D = V * 768, 1024
        x = torch.randn(M, D, device=dev)
        xb = x.to(DT)
        st = torch.empty(M, D // 64, 2, device=dev)
        hip.rowstats(x, xb, st)
        g, bt = torch.ones(D, device=dev), torch.zeros(D, device=dev)
        xn = torch.empty(M, D, dtype=DT, device=dev)
        o = torch.randn(M, D, device=dev).to(DT)
        h = torch.randn(M, 4 * D, device=dev).to(DT)
        wp = (torch.randn(D, D, device=dev) / 32).to(DT)
        w1 = (torch.randn(4 * D, D, device=dev) / 32).to(DT)
        w2 = (torch.randn(D, 4 * D, device=dev) / 64).to(DT)
        wqk = (torch.randn(2 * D, D, device=dev) / 32).to(DT)
        b1, b4, b2k = torch.zeros(D, device=dev), torch.zeros(4 * D, device=dev), torch.zeros(2 * D, device=dev)
        cs1, cs4, cs2k = torch.zeros(D, device=dev), torch.zeros(4 * D, device=dev), torch.zeros(2 * D, device=dev)
        hh = torch.empty(M, 4 * D, dtype=DT, device=dev)
        qk = torch.empty(M, 2 * D, dtype=DT, device=dev)
        print('== %d views (M = %d)' % (V, M))
        for kern in (0, 128, 256):
            t = {}
            t['LN kernel'] = timeit(lambda: hip.layernorm(x, g, bt, xn, 1e-6))
            t['rowstats'] = timeit(lambda: hip.rowstats(x, xb, st))
            t['proj+res plain'] = timeit(lambda: hip.gemm(o, wp, x, bias=b1, res=x, kernel=kern))
            t['proj+res +xcopy+stats'] = timeit(lambda: hip.gemm(o, wp, x, bias=b1, res=x, xcopy=xb, stats_out=st, kernel=kern))
            t['proj+res +xcopy only'] = timeit(lambda: hip.gemm(o, wp, x, bias=b1, res=x, xcopy=xb, kernel=kern))
            t['fc2+res plain'] = timeit(lambda: hip.gemm(h, w2, x, bias=b1, res=x, kernel=kern))
            t['fc2+res +xcopy+stats'] = timeit(lambda: hip.gemm(h, w2, x, bias=b1, res=x, xcopy=xb, stats_out=st, kernel=kern))
            t['fc1 gelu plain'] = timeit(lambda: hip.gemm(xn, w1, hh, bias=b4, act='gelu', kernel=kern))
            t['fc1 gelu fold'] = timeit(lambda: hip.gemm(xb, w1, hh, bias=b4, act='gelu', ln=(st, cs4, 1e-6), kernel=kern))
            t['qk plain'] = timeit(lambda: hip.gemm(xn, wqk, qk, bias=b2k, kernel=kern))
            t['qk fold'] = timeit(lambda: hip.gemm(xb, wqk, qk, bias=b2k, ln=(st, cs2k, 1e-6), kernel=kern))
            print('  kernel=%d: ' % kern + ' | '.join('%s %.1f' % kv for kv in t.items()))
            x.normal_()


main()
