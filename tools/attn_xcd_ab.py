#!/usr/bin/env python
"""A/B of the attention block order (PST_TUNE_ATTN_XCD) on the shapes of the headline scene: plain order vs XCD-contiguous order, same process.
    python tools/attn_xcd_ab.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from panst3r_amd import hip
from tools.attn_bench import SHAPES, run

if __name__ == '__main__':
    extra = [('build cross (15 kf)', 1, 12, 768, 11520, 64), ('decoder self 16 kf', 16, 12, 768, 768, 64), ('mixer self', 16, 16, 768, 768, 64)]
    for name, *shp in SHAPES + extra:
        res = []
        for order in (0, 1, 0, 1):
            hip.tune(hip.TUNE_ATTN_XCD, order)
            res.append(run(*shp, torch.float16, True))
        hip.tune(hip.TUNE_ATTN_XCD, 1)
        print('%-22s %-28s plain order %7.1f / %7.1f TF %8.1f us | XCD order %7.1f / %7.1f TF %8.1f us' %
              (name, shp, res[0][0], res[2][0], res[2][1], res[1][0], res[3][0], res[3][1]), flush=True)
