#!/usr/bin/env python
"""Ablation timings of the two-workgroup GEMM's main loop (gemm2g.hip ABL variants; GPU box):  python tools/g2_ablate.py [M N K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from tools.g2bench import burst, case

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (38400, 4096, 1024)
a, w, out, kw = case(M, N, K, 'fc1')
fl = 2.0 * M * N * K
hip.lib()
print('shape %s fc1 + GELU, fold consumer; us (TFLOP/s-equivalent)' % ((M, N, K),))
for tag, kern, mode in [('auto (256p)', 0, None), ('128x128', 128, None), ('2g', 2, 0), ('2g prio', 2, 1), ('2g one workgroup per CU', 2, 8),
                        ('2g no MFMA', 2, 256), ('2g no MFMA, 1 wg/CU', 2, 256 + 8), ('2g loads only', 2, 512), ('2g loads only, 1 wg/CU', 2, 512 + 8),
                        ('2g compute only (no DMA in loop)', 2, 768), ('2g compute only, 1 wg/CU', 2, 768 + 8),
                        ('2g phase priority', 2, 4), ('2g phase priority + delay 8us', 2, 4 + 2 + 16 * 8), ('2g phase priority + static', 2, 5), ('2g (again)', 2, 0), ('2g phase priority (again)', 2, 4)]:
    if mode is not None:
        hip.tune(hip.TUNE_G2_MODE, mode)
    t = burst(lambda: hip.gemm(a, w, out, kernel=kern, **kw))
    print('%-40s %8.1f us  %7.0f' % (tag, t * 1e6, fl / t / 1e12))
