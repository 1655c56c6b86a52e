#!/usr/bin/env python
"""The sequential memory build (K = 16, graph-replayed) on a CU-masked stream, for a sweep of CU counts - with and without telling the kernels their CU
budget (PST_TUNE_CUS) - next to the unmasked default stream.  What the masked two-queue stage 2 can hide."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from panst3r_amd import hip
from panst3r_amd.panst3r import CONFIG_V2, build_from_config
from panst3r_amd.synthetic import fill_module_
from panst3r_amd.scene import HipBackend
from panst3r_amd.model.common import adt

K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
model = build_from_config(CONFIG_V2).eval()
fill_module_(model.must3r_decoder, seed=1, prefix='must3r_decoder.')
model.must3r_decoder.to(dev)
h, w = 24, 32
T = h * w
enc = (torch.randn(K * T, 1024, device=dev) * 0.5).to(adt())
b = HipBackend.__new__(HipBackend)


def run():
    return model.build_memory(enc, K, h, w)


def bench(stream, cus):
    hip.tune(hip.TUNE_CUS, cus)
    try:
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            bank = run()
    finally:
        hip.tune(hip.TUNE_CUS, 0)
    cur = torch.cuda.current_stream()

    def go():
        if stream is None:
            g.replay()
        else:
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                g.replay()
            cur.wait_stream(stream)
    for _ in range(2):
        go()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        go()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3


print('K = %d' % K)
print('default stream (256 CUs)                  %7.2f ms' % bench(None, 0))
for c in (256 - 8, 192, 128, 96, 72, 64, 48):
    sa, sb, ca, cb = b.masked_streams(dev, c)
    print('masked stream, %3d CUs: budget told %7.2f ms   not told %7.2f ms' % (ca, bench(sa, ca), bench(sa, 0)), flush=True)
