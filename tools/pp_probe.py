#!/usr/bin/env python
"""One GEMM shape, N launches of the persistent 256x256 kernel, for rocprofv3 --pmc runs: python tools/pp_probe.py M N K kind pp [launches]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from tools.gemm_cases import case
M, N, K, kind, pp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 10
hip.lib()
hip.tune(hip.TUNE_G256_PP, pp)
a, w, out, kw = case(M, N, K, kind)
for _ in range(n):
    hip.gemm(a, w, out, kernel=256, **kw)
torch.cuda.synchronize()
