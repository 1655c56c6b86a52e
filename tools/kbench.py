#!/usr/bin/env python
"""Micro-benchmarks of the two MFMA kernels on the shapes of the PanSt3R path (run on the GPU box)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip

dev = 'cuda:0'


def timeit(fn, n=20, w=3):
    for _ in range(w):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


def gemm_case(M, N, K, out_fp32=False, trans=False, act=None, res=False, kernel=0):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    if trans:
        out = torch.zeros(N, M + 8, dtype=torch.bfloat16, device=dev)
    else:
        out = torch.zeros(M, N, dtype=torch.float32 if (out_fp32 or res) else torch.bfloat16, device=dev)
    t = timeit(lambda: hip.gemm(a, w, out, bias=b, act=act, trans_out=trans, res=out if res else None, kernel=kernel))
    return 2.0 * M * N * K / t / 1e12, t * 1e6


def attn_case(B, H, Nq, Nk, hd):
    q = torch.randn(B * Nq, H * hd, device=dev).to(torch.bfloat16)
    k = torch.randn(B * Nk + 8, H * hd, device=dev).to(torch.bfloat16)
    Nkp = (Nk + 7) // 8 * 8
    vt = torch.randn(H * hd, B * Nkp + 8, device=dev).to(torch.bfloat16)
    o = torch.zeros(B * Nq, H * hd, dtype=torch.bfloat16, device=dev)
    D = H * hd
    t = timeit(lambda: hip.attention(q, k, vt, o, B, H, Nq, Nk, hd, (Nq * D, hd, D), (Nk * D, hd, D), (Nkp, hd * vt.stride(0), vt.stride(0)), (Nq * D, hd, D)))
    return 4.0 * B * H * Nq * Nk * hd / t / 1e12, t * 1e6


if __name__ == '__main__':
    V = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    M = V * 768
    print('== GEMM (M=%d rows = %d views)' % (M, V))
    for name, args in [('enc qk', (M, 2048, 1024)), ('enc v^T', (M, 1024, 1024, False, True)), ('enc proj+res', (M, 1024, 1024, False, False, None, True)),
                       ('enc fc1 gelu', (M, 4096, 1024, False, False, 'gelu')), ('enc fc2+res', (M, 1024, 4096, False, False, None, True)),
                       ('dec qk', (M, 1536, 768)), ('dec fc1', (M, 3072, 768, False, False, 'gelu')), ('dec fc2+res', (M, 768, 3072, False, False, None, True)),
                       ('v1 fc1 (8 views)', (8 * 768, 22528, 2816, False, False, 'gelu')), ('v1 p8.fc2', (8 * 768, 2048, 11264)),
                       ('loftup conv-like', (8 * 49152, 384, 3456)), ('loftup q', (8 * 49152, 384, 384)),
                       ('mask head', (200, 49152, 384, True)), ('build qk', (768, 1536, 768)), ('build fc1', (768, 3072, 768, False, False, 'gelu')),
                       ('build fc2', (768, 768, 3072, False, False, None, True)), ('square 4096', (4096, 4096, 4096)), ('square 8192', (8192, 8192, 8192))]:
        tf, us = gemm_case(*args)
        print('%-20s %-40s %8.1f TF %9.1f us' % (name, args[:3], tf, us))
    print('== 128 vs 256 kernel (M=%d and M=38400)' % M)
    for MM in (M, 38400):
        for name, args in [('qk', (MM, 2048, 1024)), ('proj+res', (MM, 1024, 1024, False, False, None, True)), ('fc1 gelu', (MM, 4096, 1024, False, False, 'gelu')),
                           ('fc2+res', (MM, 1024, 4096, False, False, None, True)), ('dec fc1', (MM, 3072, 768, False, False, 'gelu')), ('sq4096', (4096, 4096, 4096)), ('sq8192', (8192, 8192, 8192))]:
            a1 = gemm_case(*args, kernel=128) if len(args) == 7 else gemm_case(*(args + (False, False, None, False)[len(args) - 3:]), kernel=128)
            a2 = gemm_case(*args, kernel=256) if len(args) == 7 else gemm_case(*(args + (False, False, None, False)[len(args) - 3:]), kernel=256)
            print('%-10s %-22s 128: %7.1f TF   256: %7.1f TF' % (name, args[:3], a1[0], a2[0]))
    print('== attention')
    for name, args in [('enc self', (V, 16, 768, 768, 64)), ('dino self', (V, 16, 769, 769, 64)), ('dec cross K=16', (1, 12, M, 12288, 64)),
                       ('dec cross K=32', (1, 12, M, 24576, 64)), ('build cross j=8', (1, 12, 768, 6144, 64)), ('build self', (1, 12, 768, 768, 64)),
                       ('loftup cross', (8, 4, 49152, 768, 96)), ('qdec cross', (1, 8, 200, 12288, 96))]:
        tf, us = attn_case(*args)
        print('%-20s %-40s %8.1f TF %9.1f us' % (name, args, tf, us))
