#!/usr/bin/env python
"""A few launches of the attention kernel on one shape (driver for tools/sq_profile.sh): attn_one.py B H Nq Nk hd [pre]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
B, H, Nq, Nk, hd = map(int, sys.argv[1:6])
pre = len(sys.argv) > 6 and sys.argv[6] == 'pre'
dev, D, dt = 'cuda:0', H * hd, torch.float16
q = (torch.randn(B * Nq, D, device=dev) * (hd ** -0.5 * hip.LOG2E if pre else 1.0)).to(dt)
k = torch.randn(B * Nk + 8, D, device=dev).to(dt)
vt = torch.randn(D, B * Nk + 8, device=dev).to(dt)
o = torch.zeros(B * Nq, D, dtype=dt, device=dev)
for _ in range(4):
    hip.attention(q, k, vt, o, B, H, Nq, Nk, hd, (Nq * D, hd, D), (Nk * D, hd, D), (Nk, hd * vt.stride(0), vt.stride(0)), (Nq * D, hd, D), prescaled=pre)
torch.cuda.synchronize()
