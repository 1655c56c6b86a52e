#!/usr/bin/env python
"""Sustained time of the persistent 256 x 256 kernel on the scene's big shapes (single problems and the two towers' pairs), one line per case:
run once per library build to A/B two builds on one box (tools/epi_ab.sh).  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from tools.gemm_cases import case
from tools.dispatch_bench import compare

hip.lib()
CASES = (('fc1+gelu', 4096, 1024, 'fc1'), ('qk+rope', 2048, 1024, 'qk'), ('q', 2048, 1024, 'q'), ('v^T', 1024, 1024, 'vt'), ('proj+res', 1024, 1024, 'res'), ('fc2+res', 1024, 4096, 'res'))
DEC = (('dec fc1', 3072, 768, 'fc1'), ('dec qk', 1536, 768, 'qk'), ('dec q', 768, 768, 'q'), ('dec fc2', 768, 3072, 'res'))
for name, n, k, kind in CASES:
    A = case(26112, n, k, kind)
    B = case(38800, n, k, 'q' if kind == 'qk' else kind)          # DINOv2 carries no RoPE
    fl = 2.0 * (26112 + 38800) * n * k
    t = compare([lambda: hip.gemm(A[0], A[1], A[2], kernel=256, **A[3]), lambda: hip.gemm(B[0], B[1], B[2], kernel=256, **B[3]),
                 lambda: hip.gemm_pair((A[0], A[1], A[2], A[3]), (B[0], B[1], B[2], B[3]))])
    print('%-10s K=%-5d N=%-5d  enc %7.1f us  dino %7.1f us  pair %7.1f us  %6.0f TF (pair)' % (name, k, n, t[0], t[1], t[2], fl / t[2] / 1e6), flush=True)
    del A, B
for name, n, k, kind in DEC:
    A = case(38400, n, k, kind)
    t = compare([lambda: hip.gemm(A[0], A[1], A[2], kernel=256, **A[3])])
    print('%-10s K=%-5d N=%-5d  %7.1f us  %6.0f TF' % (name, k, n, t[0], 2.0 * 38400 * n * k / t[0] / 1e6), flush=True)
    del A
