"""Comparison of a HIP scene with a full-size oracle FIXTURE (tests/golden/fullsize_<tag>.npz, written once in the build container by
tests/golden/make_fullsize_golden.py: the fp32 CPU oracle's outputs as samples + whole-tensor statistics).  Everything is gathered on the device: the 200-view
scene's 7.8 GB of mask logits never travel to the host.  Returns the record bench._scene_errors returns (same keys, same criteria - on the sampled pixels),
plus full-coverage statistics: per-view L2-norm ratios of pointmaps / mask logits and the fraction of positive logits against the oracle's."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load(tag):
    path = os.path.join(HERE, 'golden', 'fullsize_%s.npz' % tag)
    if not os.path.exists(path):
        return None
    return dict(np.load(path))


def attention_bits(g, dev):
    """the oracle's attention-mask decisions per query-decoder layer ([Q, K*T] uint8 each), or None when the fixture does not carry them"""
    if 'attn_bits' not in g:
        return None
    L, Q, NK = (int(x) for x in g['attn_bits_shape'])
    bits = np.unpackbits(g['attn_bits'], axis=-1)[..., :NK]
    return [torch.from_numpy(bits[l]).to(dev) for l in range(L)]


def sig(x, digits=3):
    return float('%.*g' % (digits, x))


def scene_errors(pm_h, pan_h, g):
    dev = pm_h[0].device
    V = int(g['shape'][0])
    assert len(pm_h) == V and len(pan_h['pred_masks']) == V, (len(pm_h), V)
    pm_idx = torch.from_numpy(g['pm_idx']).to(dev)
    mk_idx = torch.from_numpy(g['mk_idx']).to(dev)
    pm_o = torch.from_numpy(g['pm']).to(dev).double()
    mk_o = torch.from_numpy(g['mk']).to(dev)
    sg_idx = sg_o = None
    if 'sg_bits' in g:                                   # sign bits at further pixels: the sample the sign-agreement criterion is evaluated on
        sg_idx = torch.from_numpy(g['sg_idx']).to(dev)
        sg_o = torch.from_numpy(np.unpackbits(g['sg_bits'], axis=-1)[..., :sg_idx.numel()]).to(dev).bool()
    pm_rel, pm_ratio, mk_rel, mk_sign, mk_ratio, mk_pos = [], [], [], [], [], []
    num = den = agree = cnt = 0.0
    for v in range(V):
        a = pm_h[v].reshape(-1, pm_h[v].shape[-1])[pm_idx].double()
        pm_rel.append(float((a - pm_o[v]).norm() / pm_o[v].norm().clamp_min(1e-30)))
        pm_ratio.append(float(pm_h[v].double().norm()) / float(g['pm_norm'][v]))
        m = pan_h['pred_masks'][v]
        Q = m.shape[1]
        a = m.reshape(Q, -1)[:, mk_idx]
        b = mk_o[v]
        d2, b2 = float((a.double() - b.double()).pow(2).sum()), float(b.double().pow(2).sum())
        if sg_o is not None:
            ag, n_sg = float(((m.reshape(Q, -1)[:, sg_idx] > 0) == sg_o[v]).sum()), sg_o[v].numel()
        else:
            ag, n_sg = float(((a > 0) == (b > 0)).sum()), b.numel()
        num, den, agree, cnt = num + d2, den + b2, agree + ag, cnt + n_sg
        mk_rel.append((d2 / max(b2, 1e-300)) ** 0.5)
        mk_sign.append(ag / n_sg)
        mk_ratio.append(float(m.double().norm()) / float(g['mk_norm'][v]))
        mk_pos.append((float((m > 0).sum()) - float(g['mk_pos'][v])) / m.numel())
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    lo, oq = torch.from_numpy(g['pred_logits']), torch.from_numpy(g['out_queries'])
    return {'pointmaps_rel_l2': sig(max(pm_rel)), 'pointmaps_rel_l2_per_view': [sig(x) for x in pm_rel],
            'mask_logits_rel_l2_per_view': [sig(x) for x in mk_rel],
            'mask_logits_rel_l2': sig((num / max(den, 1e-300)) ** 0.5), 'mask_sign_agreement': round(agree / cnt, 5),
            'worst_view': {'mask_logits_rel_l2': sig(max(mk_rel)), 'mask_sign_agreement': round(min(mk_sign), 5)},
            'class_logits_max_abs': sig(float((pan_h['pred_logits'].cpu() - lo).abs().max())),
            'out_queries_rel_l2': sig(rel(pan_h['out_queries'], oq)),
            # whole tensors, not samples: L2 norm of every view's pointmap / mask block against the oracle's, and the change of its share of positive logits
            'full_coverage': {'pointmap_norm_ratio_max_dev': sig(max(abs(r - 1.0) for r in pm_ratio)), 'mask_norm_ratio_max_dev': sig(max(abs(r - 1.0) for r in mk_ratio)),
                              'mask_positive_share_max_dev': sig(max(abs(x) for x in mk_pos))},
            'samples': {'pointmap_pixels_per_view': int(pm_idx.numel()), 'mask_pixels_per_view': int(mk_idx.numel()),
                        'sign_pixels_per_view': int(sg_idx.numel()) if sg_idx is not None else int(mk_idx.numel()), 'views': V}}
