#!/usr/bin/env python
"""Memory build (K = 16, graph-replayed) with the 64x64-tile GEMM's LDS ring at 4 / 6 / 8 slabs for launches of at most one tile per CU (PST_TUNE_DEEP_RING)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from panst3r_amd.panst3r import CONFIG_V2, build_from_config
from panst3r_amd.synthetic import fill_module_
from panst3r_amd.model.common import adt
K = 16
dev = torch.device('cuda:0')
model = build_from_config(CONFIG_V2).eval()
fill_module_(model.must3r_decoder, seed=1, prefix='must3r_decoder.')
model.must3r_decoder.to(dev)
h, w = 24, 32
enc = (torch.randn(K * h * w, 1024, device=dev) * 0.5).to(adt())
ref = None
for ring in (4, 6, 8, 4, 6, 8):
    hip.tune(hip.TUNE_DEEP_RING, ring)
    bank = model.build_memory(enc, K, h, w); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        bank = model.build_memory(enc, K, h, w)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8 * 1e3
    kk = bank.K_all.clone()
    same = ref is None or torch.equal(kk, ref)
    ref = kk if ref is None else ref
    print('ring %d: build %.2f ms  (bank bit-identical to ring 4: %s)' % (ring, dt, same), flush=True)
