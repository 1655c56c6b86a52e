#!/usr/bin/env python
"""Full-size oracle fixtures for BASELINE configs[3] and configs[4] ("full-size goldens stored as strided samples + norms").

Run (build container; tens of minutes to hours of host time, once):
    python tests/golden/make_fullsize_golden.py c4        # v2, 50 views / 16 keyframes, 384 x 512
    python tests/golden/make_fullsize_golden.py c5        # v2, 200 views / 32 keyframes, 384 x 512

What runs is the fp32 CPU oracle (oracle/pipeline.py, the restatement that tests/test_oracle_golden.py pins against reference-generated vectors for
the reference-owned half) on the synthetic scene and weights of panst3r_amd.synthetic - exactly the call bench.cpu_baseline makes and
tests/test_hip_fullsize.py used to make at test time.  Only DATA is written (tests/golden/fullsize_<tag>.npz):
  pm_idx [n_pm]            flat pixel indices into H*W (one seeded draw, shared by all views)
  pm [V, n_pm, 7]          pointmap rows at those pixels           pm_norm [V]  L2 norm of each view's whole pointmap (float64)
  mk_idx [n_mk]            flat pixel indices into (H/2)*(W/2)
  mk [V, Q, n_mk]          mask logits of every query at those pixels
  mk_norm [V], mk_pos [V]  L2 norm of each view's whole [Q, H/2, W/2] block, number of positive logits in it (full-coverage statistics)
  sg_idx [n_sg], sg_bits   SIGNS of the mask logits at n_sg further pixels (np.packbits of [V, Q, n_sg] "logit > 0"): the sign-agreement criterion per view
                           needs thousands of samples per view (the flips of a view cluster on few pixels), the values do not
  pred_logits, out_queries whole
  attn_bits                (c4 only) the oracle's attention-mask decisions of every query-decoder layer, np.packbits of [L, Q, K*T]
  keyframes                the keyframe view ids the oracle used
Weights and images are not stored: both sides regenerate them (fill_module_(seed=1), synth_image(view))."""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

CASES = {'c4': dict(variant='v2', V=50, K=16, n_pm=128, n_mk=16, n_sg=256, bits=True),
         'c5': dict(variant='v2', V=200, K=32, n_pm=48, n_mk=5, n_sg=128, bits=False),
         'tiny': dict(variant='v2', V=3, K=2, n_pm=128, n_mk=16, n_sg=256, bits=True)}      # (self-test of this script and of the comparison helper)
H, W = 384, 512


def sample_indices(tag, n_pm, n_mk, n_sg):
    g = np.random.Generator(np.random.PCG64(20260930))
    pm, mk = np.sort(g.choice(H * W, n_pm, replace=False)).astype(np.int64), np.sort(g.choice((H // 2) * (W // 2), n_mk, replace=False)).astype(np.int64)
    return pm, mk, np.sort(g.choice((H // 2) * (W // 2), n_sg, replace=False)).astype(np.int64)


def main(tag):
    import bench
    from panst3r_amd.panst3r import CONFIG_V1, CONFIG_V2, build_from_config
    from panst3r_amd.synthetic import fill_module_, synth_class_embeddings
    c = CASES[tag]
    model = build_from_config(CONFIG_V2 if c['variant'] == 'v2' else CONFIG_V1).eval()
    fill_module_(model, seed=1)
    names, emb = synth_class_embeddings(100)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    del model
    threads = bench.usable_cores()
    t0 = time.time()
    rec, (pm_o, pan_o), imgs, ts = bench.cpu_baseline(c['variant'], H, W, state, names, emb, threads, V=c['V'], K=c['K'])
    print(tag, rec, 'wall %.0f s' % (time.time() - t0), flush=True)
    pm_idx, mk_idx, sg_idx = sample_indices(tag, c['n_pm'], c['n_mk'], c['n_sg'])
    V = c['V']
    pm = np.stack([pm_o[v].reshape(-1, pm_o[v].shape[-1])[pm_idx].numpy() for v in range(V)]).astype(np.float32)
    pm_norm = np.array([float(pm_o[v].double().norm()) for v in range(V)])
    masks = pan_o['pred_masks']
    Q = masks[0].shape[1]
    mk = np.stack([masks[v].reshape(Q, -1)[:, mk_idx].numpy() for v in range(V)]).astype(np.float32)
    mk_norm = np.array([float(masks[v].double().norm()) for v in range(V)])
    mk_pos = np.array([int((masks[v] > 0).sum()) for v in range(V)], dtype=np.int64)
    sg = np.stack([(masks[v].reshape(Q, -1)[:, sg_idx] > 0).numpy() for v in range(V)]).astype(np.uint8)
    out = dict(pm_idx=pm_idx, pm=pm, pm_norm=pm_norm, mk_idx=mk_idx, mk=mk, mk_norm=mk_norm, mk_pos=mk_pos, sg_idx=sg_idx, sg_bits=np.packbits(sg, axis=-1),
               pred_logits=pan_o['pred_logits'].numpy().astype(np.float32), out_queries=pan_o['out_queries'].numpy().astype(np.float32),
               shape=np.array([V, c['K'], H, W]), oracle_frames_per_s=np.array(rec['value']), oracle_threads=np.array(threads))
    if c['bits']:
        bits = torch.stack(pan_o['attn_masks']).numpy().astype(np.uint8)              # [L, Q, K*T]
        out['attn_bits'] = np.packbits(bits, axis=-1)
        out['attn_bits_shape'] = np.array(bits.shape)
    path = os.path.join(HERE, 'fullsize_%s.npz' % tag)
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes', flush=True)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'c4')
