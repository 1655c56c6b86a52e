#!/usr/bin/env python
"""Self-attention at 768 keys as a function of the number of 128-query blocks (B x 16 heads x 6): where does the time go - block quantisation
(1 024 resident blocks per round?) or a per-block fixed cost?  GPU box: python tools/attn_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.attn_bench import run
for Nk in (768, 1536, 3072):
    for B in (8, 10, 11, 16, 21, 22, 32, 34, 42, 43, 50, 64, 85, 86):
        tf, us = run(B, 16, 768, Nk, 64, torch.float16, True)
        blocks = B * 16 * 6
        print('Nk %5d  B %3d  blocks %5d (%.2f x 1024)  %8.1f us  %6.1f TF  %.3f us per block-tile per CU' % (Nk, B, blocks, blocks / 1024, us, tf, us / (blocks * (Nk // 64) / 256)), flush=True)
