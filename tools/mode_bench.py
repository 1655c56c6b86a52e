#!/usr/bin/env python
"""frames/s of the bench scene (v2, 50 views / 16 keyframes, 384x512) in the precision modes of the API:  python tools/mode_bench.py [mode ...]
   modes: fp16 | bf16 | fp32 (amp=False: 3 x f16 split operands) | fp32_exact | fp16+reference | bf16+reference | fp16+reference_exact"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from panst3r_amd import hip                                                   # noqa: E402
from panst3r_amd.panst3r import CONFIG_V2, build_from_config                  # noqa: E402
from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings   # noqa: E402

V, K, H, W = int(os.environ.get('PST_V', 50)), int(os.environ.get('PST_K', 16)), 384, 512
dev = torch.device('cuda:0')
model = build_from_config(CONFIG_V2).eval()
fill_module_(model, seed=1)
names, emb = synth_class_embeddings(100)
model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
model.to(dev)
images = {i: synth_image(i, H, W).to(dev) for i in range(V)}
MODES = {'fp16': ('fp16', None), 'bf16': ('bf16', None), 'fp32': (False, None), 'fp32_exact': ('fp32_exact', None), 'fp16+reference': ('fp16', 'reference'),
         'bf16+reference': ('bf16', 'reference')}
SHAPES = '--shapes' in sys.argv           # also: one eager scene with every MFMA launch bracketed by HIP events, summed per (kernel, shape tag)
argv = [a for a in sys.argv[1:] if a != '--shapes']
for mode in (argv or ['fp16', 'fp32', 'fp16+reference']):
    exact = mode.endswith('_exact') and '+' in mode
    amp, pp = MODES[mode[:-6] if exact else mode]
    if exact:                                     # the fp32 segments of the reference placement on the fp32-input-MFMA kernels (round-4 behaviour)
        import panst3r_amd.panst3r as P
        orig = P.pan_amp_of
        P.pan_amp_of = lambda a, p: ('fp32_exact', 'reference') if p == 'reference' else orig(a, p)
    runner = model.scene_runner(images, V, H, W, names, num_keyframes=K, use_graphs=True, amp=amp, panoptic_precision=pp)
    runner.run(copy=False)
    runner.run(copy=False)
    steps = 3 if 'exact' in mode else 8
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.run(copy=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print('%-22s %8.2f ms per scene  %7.2f frames/s' % (mode, 1e3 * dt, V / dt), flush=True)
    if SHAPES:
        eager = model.scene_runner(images, V, H, W, names, num_keyframes=K, use_graphs=False, amp=amp, panoptic_precision=pp)
        eager.run(copy=False)
        hip.TIMER = timer = hip.KernelTimer()
        eager.run(copy=False)
        hip.TIMER = None
        rows = sorted(timer.by_tag().items(), key=lambda kv: -kv[1]['ms'])
        tot = sum(d['ms'] for _, d in rows)
        print('   instrumented launches: %.1f ms in %d (kernel, shape) rows' % (tot, len(rows)))
        for (name, tag), d in rows[:40]:
            print('   %-22s %-70s x%-4d %8.2f ms %7.1f TF' % (name, tag, d['launches'], d['ms'], d['flops'] / max(d['ms'], 1e-9) * 1e-9), flush=True)
        eager.release()
        del eager
    if exact:
        P.pan_amp_of = orig
    runner.release()
    del runner
    torch.cuda.empty_cache()
