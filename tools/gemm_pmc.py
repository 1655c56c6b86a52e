#!/usr/bin/env python
"""One launch each of a few GEMM shapes (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
dev = 'cuda:0'
for (M, N, K) in [(12288, 2048, 1024), (12288, 1024, 4096), (4096, 4096, 4096)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        hip.gemm(a, w, out)
    torch.cuda.synchronize()
