#!/usr/bin/env python
"""f16 range safety on the "outlier" synthetic weight set (panst3r_amd.synthetic OUTLIER_CHANNELS: a few residual-writing rows of every backbone block x S, what
trained ViT-L / DINOv2 checkpoints look like): full-size v2, V views / K keyframes, each operand format against the fp32 CPU oracle with the same weights, plus
the per-stage max |x| of every 16-bit tensor (hip.maxabs_telemetry).   python tools/range_probe.py [S ...] [--views V --keyframes K]   -> profiles/r6_range_probe.txt"""
import argparse
import json
import os
import sys
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from panst3r_amd import hip
from panst3r_amd.panst3r import CONFIG_V2, build_from_config
from panst3r_amd.synthetic import fill_module_, synth_class_embeddings

ap = argparse.ArgumentParser()
ap.add_argument('scales', nargs='*', type=float, default=[1e3, 3e4])
ap.add_argument('--views', type=int, default=2)
ap.add_argument('--keyframes', type=int, default=2)
args = ap.parse_args()
dev = torch.device('cuda:0')
hip.lib()
names, emb = synth_class_embeddings(100)
for S in args.scales:
    model = build_from_config(CONFIG_V2).eval()
    fill_module_(model, seed=1, outlier=S)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
    model.to(dev)
    rec, ref, imgs, ts = bench.cpu_baseline('v2', 384, 512, state, names, emb, bench.usable_cores(), V=args.views, K=args.keyframes)
    print('== outlier scale %g, %d views / %d keyframes; oracle %s' % (S, args.views, args.keyframes, rec['sample']), flush=True)
    for amp, pp in (('fp16', None), ('bf16', None), ('bf16', 'amp')):
        with warnings.catch_warnings(record=True) as wlog:
            warnings.simplefilter('always')
            with hip.maxabs_telemetry() as log:
                try:
                    par = bench.full_size_parity(model, dev, ref, imgs, ts, names, amp=amp, K=args.keyframes, panoptic_precision=pp)
                except Exception as e:
                    par = {'error': repr(e)}
        fell = [str(w.message)[:160] for w in wlog if 'repeating the call' in str(w.message)]
        keep = {k: par.get(k) for k in ('pointmaps_rel_l2', 'mask_logits_rel_l2', 'mask_sign_agreement', 'class_logits_max_abs', 'out_queries_rel_l2', 'within_tolerance', 'error') if k in par}
        print('amp=%s panoptic_precision=%s -> ran as %s%s: %s' % (amp, pp, getattr(model, 'last_precision', None), (' after ' + ' | '.join(fell)) if fell else '', json.dumps(keep)), flush=True)
        for stage, what, v, frac in hip.maxabs_report(log, top=6):
            print('      max |x| %10.4g (%.3f of the f16 range)  %-40s %s' % (v, frac, stage, what))
    del model
