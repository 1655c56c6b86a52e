#!/usr/bin/env python
"""Latency of the sequential keyframe-memory build alone (the Amdahl term of the multi-GPU plan), graph-replayed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd.panst3r import CONFIG_V2, build_from_config
from panst3r_amd.synthetic import fill_module_

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
model = build_from_config(CONFIG_V2).eval()
fill_module_(model.must3r_decoder, seed=1, prefix='must3r_decoder.')
model.must3r_decoder.to(dev)
dec = model.must3r_decoder
h, w = 24, 32
T = h * w
from panst3r_amd.model.common import adt
enc = (torch.randn(K * T, 1024, device=dev) * 0.5).to(adt())
def run():
    return model.build_memory(enc, K, h, w)
run(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    bank = run()
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print('K=%d: build %.2f ms (%.2f ms per step)' % (K, (time.perf_counter() - t0) / 5 * 1e3, (time.perf_counter() - t0) / 5 * 1e3 / (K - 1)))
