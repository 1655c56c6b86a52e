#!/usr/bin/env python
"""Pure-PyTorch attempt at the two-queue lost-write effect of DESIGN.md section 4 (no kernel of this repository involved).

A captured graph with two parallel branches: the main branch loops a small torch.mm (rocBLAS / hipBLASLt) on private buffers; the side
branch recycles big temporaries (like an encoder pass would), then writes a fresh fp32 buffer with an elementwise kernel and reads it
back with another one.  Every replay is compared with the serial result.  Prints how many replays deviate and the size of the damage."""
import os, sys
import torch
import torch.nn.functional as F

dev = torch.device('cuda:0')
R = int(os.environ.get('PST_R', '25'))
MM = int(os.environ.get('PST_MM', '3000'))
torch.manual_seed(0)
img = torch.rand(13, 3, 384, 512, device=dev) * 2 - 1
a = torch.randn(768, 1024, device=dev).bfloat16(); b = torch.randn(1024, 1024, device=dev).bfloat16(); c = torch.empty(768, 1024, device=dev, dtype=torch.bfloat16)
A = torch.randn(6912, 1024, device=dev).bfloat16(); W1 = torch.randn(1024, 4096, device=dev).bfloat16(); W2 = torch.randn(4096, 1024, device=dev).bfloat16()
mean = torch.tensor([0.485, 0.456, 0.406], device=dev).view(1, 3, 1, 1); std = torch.tensor([0.229, 0.224, 0.225], device=dev).view(1, 3, 1, 1)

def main_branch():
    for _ in range(MM):
        torch.mm(a, b, out=c)

def side_branch(outs):
    x = A
    for _ in range(6):                                   # big recycled temporaries: every h / x below is freed and its block reused
        h = F.gelu(torch.mm(x, W1))
        x = torch.mm(h, W2) * 0.01
    del h
    for _ in range(8):                                   # producer / consumer pairs on fresh buffers (the first DINOv2 kernels in the real scene)
        pre = ((F.interpolate(img, size=(336, 448), mode='bilinear', align_corners=False) * 0.5 + 0.5) - mean) / std
        outs.append(pre.clone())
        del pre
        t = torch.mm(x, W1); del t

def run(two_streams, outs):
    if two_streams:
        main = torch.cuda.current_stream()
        side = SIDE
        side.wait_stream(main)
        with torch.cuda.stream(side):
            side_branch(outs)
        main_branch()
        main.wait_stream(side)
    else:
        side_branch(outs)
        main_branch()

SIDE = torch.cuda.Stream()
ref = []
run(False, ref); torch.cuda.synchronize()                # warm-up (library initialisation) and the serial reference
ref = [r.clone() for r in ref]
for mode in ('one stream', 'two streams'):
    outs = []
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        run(mode == 'two streams', outs)
    bad, worst, sizes = 0, 0.0, []
    for rep in range(R):
        g.replay(); torch.cuda.synchronize()
        hit = False
        for o, r in zip(outs, ref):
            if not torch.equal(o, r):
                d = (o - r).abs()
                hit = True
                worst = max(worst, float(d.max()))
                sizes.append(int((d > 0).sum()))
        bad += hit
    print('%-11s: %d of %d replays deviate from the serial reference; worst |diff| %.3g; differing elements per damaged buffer %s' % (mode, bad, R, worst, sizes[:8]))
