#!/usr/bin/env python
"""sustained time of one GEMM case on the persistent 256x256 kernel: python tools/pp_time.py M N K kind [pp] [workgroups]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from tools.gemm_cases import case
from tools.dispatch_bench import compare
M, N, K, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
hip.lib()
hip.tune(hip.TUNE_G256_PP, int(sys.argv[5]) if len(sys.argv) > 5 else 1)
if len(sys.argv) > 6:
    hip.tune(hip.TUNE_CUS, int(sys.argv[6]))          # grid of the persistent kernel (as on a CU-masked stream)
a, w, out, kw = case(M, N, K, kind)
if os.environ.get('PST_OPERANDS') == 'zeros':      # the same launch on all-zero operands: what the data-dependent power draw of the matrix cores costs (clocks)
    a.zero_(); w.zero_()
t = compare([lambda: hip.gemm(a, w, out, kernel=256, **kw)])[0]
tiles = ((M + 255) // 256) * ((N + 255) // 256)
wgs = int(sys.argv[6]) if len(sys.argv) > 6 else torch.cuda.get_device_properties(0).multi_processor_count
print('%s %s: %.1f us  %.0f TF-equivalent; %.2f us per round of %d workgroups' % ((M, N, K), kind, t, 2.0 * M * N * K / t / 1e6, t / (tiles / wgs), wgs))
