#!/usr/bin/env python
"""Attention kernel throughput on the shapes of the headline scene (GPU box): plain vs prescaled mode, both 16-bit formats.
    python tools/attn_bench.py [--fmt f16|bf16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from panst3r_amd import hip
from tools.kbench import timeit

SHAPES = [('render cross-attn', 1, 12, 38400, 12288, 64), ('encoder self', 34, 16, 768, 768, 64), ('dino self', 50, 16, 769, 769, 64),
          ('decoder self', 50, 12, 768, 768, 64), ('loftup cross hd96', 16, 4, 49152, 768, 96), ('build self (1 view)', 1, 12, 768, 768, 64),
          ('build cross (8 kf)', 1, 12, 768, 6144, 64)]


def run(B, H, Nq, Nk, hd, dt, pre):
    dev = 'cuda:0'
    D = H * hd
    q = (torch.randn(B * Nq, D, device=dev) * (hd ** -0.5 * hip.LOG2E if pre else 1.0)).to(dt)
    Nkp = (Nk + 7) // 8 * 8
    k = torch.randn(B * Nkp + 8, D, device=dev).to(dt)
    vt = torch.randn(D, B * Nkp + 8, device=dev).to(dt)
    o = torch.zeros(B * Nq, D, dtype=dt, device=dev)
    ws = torch.empty(max(hip.attn_workspace_floats(B, H, Nq, Nk, hd), 1), dtype=torch.float32, device=dev)
    f = lambda: hip.attention(q, k, vt, o, B, H, Nq, Nk, hd, (Nq * D, hd, D), (Nkp * D, hd, D), (Nkp, hd * vt.stride(0), vt.stride(0)), (Nq * D, hd, D), ws=ws, prescaled=pre)
    t = timeit(f)
    return 4.0 * B * H * Nq * Nk * hd / t / 1e12, t * 1e6


if __name__ == '__main__':
    fmt = torch.bfloat16 if '--fmt' in sys.argv and sys.argv[sys.argv.index('--fmt') + 1] == 'bf16' else torch.float16
    for name, *shp in SHAPES:
        a, b = run(*shp, fmt, False), run(*shp, fmt, True)
        print('%-22s %-28s plain %7.1f TF %8.1f us | prescaled %7.1f TF %8.1f us' % (name, shp, a[0], a[1], b[0], b[1]), flush=True)
