#!/usr/bin/env python
"""sustained time of one GEMM case on the persistent 256x256 kernel: python tools/pp_time.py M N K kind [pp]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from tools.gemm_cases import case
from tools.dispatch_bench import compare
M, N, K, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
hip.lib()
hip.tune(hip.TUNE_G256_PP, int(sys.argv[5]) if len(sys.argv) > 5 else 1)
a, w, out, kw = case(M, N, K, kind)
t = compare([lambda: hip.gemm(a, w, out, kernel=256, **kw)])[0]
print('%s %s: %.1f us  %.0f TF-equivalent' % ((M, N, K), kind, t, 2.0 * M * N * K / t / 1e6))
