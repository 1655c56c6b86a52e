"""Oracle restatement of `panoptic_inference_v2` (TEST INFRASTRUCTURE; CPU, fp32, plain torch).

Follows reference engine/postprocess.py:14-130, multi_ar=True calling convention of the demo
(tools/demo_panst3r.py:236-242): per-view mask logits -> sigmoid -> bilinear resize to true_shape (:19-21), padded stack
(:22-23), per-query score/label/keep (:40-52), then `niters` rounds of {score-weighted argmax over the surviving
queries (:77), per-query area test mask_area / original_area >= overlap_threshold (:87-93), segment ids in query order
(:104-106)}; the last round's panoptic_seg / conf are returned, cropped per view (:121-123).

Pinned by tests/golden/postprocess_v2*.npz, generated from the reference's own function (tests/golden/make_golden.py G6).
The selection of a query in one round does not depend on the other queries' selection in that round (the argmax is
taken once per round, before the loop), so the per-query Python loop of the reference is restated as vector ops.
Both label modes are restated: 'sigmoid' (the released configs, configs/base.yaml:24) and 'softmax' (:48-51,59-60: the last class column is
"no object"; pinned by tests/golden/postprocess_v2_softmax.npz).
"""
import numpy as np
import torch
import torch.nn.functional as F


def query_scores(mask_cls, cls_threshold=0.1, temperature=None, label_mode='sigmoid'):
    """postprocess.py:40-51: scores, labels, keep of one scene's class logits [Q, Ncls] (softmax mode: column Ncls-1 = "no object")."""
    if label_mode == 'softmax':                                                                   # :48-51 (the temperature is not read in this mode)
        scores, labels = F.softmax(mask_cls, dim=-1).max(-1)
        return scores, labels, labels.ne(mask_cls.shape[-1] - 1) & (scores > cls_threshold)
    probs = mask_cls.sigmoid()
    scores, labels = probs.max(-1)
    keep = scores > cls_threshold
    if temperature is not None:
        scores, labels = F.softmax(probs / temperature, dim=-1).max(-1)
    return scores, labels, keep


@torch.no_grad()
def panoptic_inference_v2(mask_cls, mask_pred, true_shape, label_mode='sigmoid', cls_threshold=0.1, temperature=None,
                          mask_threshold=0.25, overlap_threshold=0.5, niters=2, void_confidence=0.1, device=None, multi_ar=True):
    """mask_cls [1,Q,Ncls]; mask_pred list[V] of [1,Q,h,w] logits; true_shape [V,2].
    Returns [{'pan': list[V] int32 [H,W], 'segments_info': [...], 'conf': list[V] fp32 [H,W]}]."""
    assert label_mode in ('sigmoid', 'softmax') and multi_ar and mask_cls.shape[0] == 1
    shapes = [tuple(int(v) for v in s) for s in true_shape]
    V = len(mask_pred)
    Hm, Wm = max(s[0] for s in shapes), max(s[1] for s in shapes)
    Q = mask_cls.shape[1]
    probs = torch.zeros(Q, V, Hm, Wm)                                                             # :22-23 (zero padded), :37
    for i, m in enumerate(mask_pred):
        up = F.interpolate(m.float().sigmoid(), size=list(shapes[i]), mode='bilinear', align_corners=False)   # :20-21
        probs[:, i, :shapes[i][0], :shapes[i][1]] = up[0]
    scores, labels, keep = query_scores(mask_cls[0].float(), cls_threshold, temperature, label_mode)
    cur_idx = torch.nonzero(keep)[:, 0]
    cur_scores, cur_classes, cur_masks = scores[cur_idx], labels[cur_idx], probs[cur_idx]         # :54-58
    cur_prob = cur_scores.view(-1, 1, 1, 1) * cur_masks                                           # :64
    pan = torch.zeros(V, Hm, Wm, dtype=torch.int32)
    conf = torch.full((V, Hm, Wm), float(void_confidence))
    segments = []
    for _ in range(niters):                                                                       # :67
        pan = torch.zeros(V, Hm, Wm, dtype=torch.int32)
        conf = torch.full((V, Hm, Wm), float(void_confidence))
        segments = []
        if cur_masks.shape[0] == 0:                                                               # :71-73
            break
        ids = cur_prob.argmax(0)                                                                  # :78
        n = cur_masks.shape[0]
        original_area = (cur_masks >= 0.5).flatten(1).sum(1)                                      # :86
        owned = (ids[None] == torch.arange(n).view(-1, 1, 1, 1)) & (cur_masks >= mask_threshold)  # :87
        mask_area = owned.flatten(1).sum(1)
        ok = (mask_area > 0) & (original_area > 0)
        ratio = mask_area.double() / original_area.clamp(min=1).double()                          # python float division (:91)
        sel = ok & ~(ratio < overlap_threshold)
        seg_id = torch.cumsum(sel.int(), 0)                                                       # :104 (everything is a "thing", :85)
        for k in torch.nonzero(sel)[:, 0].tolist():
            pan[owned[k]] = int(seg_id[k])                                                        # :105
            conf[owned[k]] = cur_masks[k][owned[k]]                                               # :106
            segments.append({'id': int(seg_id[k]), 'query_id': int(cur_idx[k]), 'category_id': int(cur_classes[k])})
        cur_prob, cur_classes, cur_idx, cur_masks = cur_prob[sel], cur_classes[sel], cur_idx[sel], cur_masks[sel]   # :115-119
    return [{'pan': [pan[i, :h, :w].contiguous() for i, (h, w) in enumerate(shapes)], 'segments_info': segments,
             'conf': [conf[i, :h, :w].contiguous() for i, (h, w) in enumerate(shapes)]}]


def panoptic_inference_v1(*args, mask_threshold=0.5, overlap_threshold=0.8, **kwargs):
    """reference engine/postprocess.py:9-11: one round with the Mask2Former thresholds"""
    return panoptic_inference_v2(*args, mask_threshold=mask_threshold, overlap_threshold=overlap_threshold, niters=1, **kwargs)


# ------------------------------------------------------------------------------------------------ QUBO (engine/postprocess.py:135-336)
def qubo_weights(mask_pred, true_shape, penalty=1):
    """-W of `weight_from_masks` (:229-259) for a list of mask logits [1,Q,h,w]: sigmoid, bilinear to the true shapes, zero-padded to the
    largest; W_ii = mask area, W_ij = -(1 + penalty) sum min(m_i, m_j) / 2, normalised by padded image size and view count (float32)."""
    up = [F.interpolate(m.float().sigmoid(), size=[int(v) for v in ts], mode='bilinear', align_corners=False)[0] for m, ts in zip(mask_pred, true_shape)]
    Hm, Wm = max(u.shape[-2] for u in up), max(u.shape[-1] for u in up)
    Q = up[0].shape[0]
    S = torch.zeros(Q, Q, dtype=torch.float64)
    for u in up:
        f = u.flatten(1).double()
        for i in range(Q):
            S[i] += torch.minimum(f[i][None], f).sum(1)
    W = -(1 + penalty) * S / 2
    W[torch.arange(Q), torch.arange(Q)] = torch.diagonal(S)
    W = W / (Hm * Wm) / len(up)
    return (-W).float().numpy(), up, (Hm, Wm)


def panoptic_inference_qubo(mask_cls, mask_pred, true_shape, num_redo=20, prob_threshold=0.01, temperature=None):
    """CPU restatement of the reference's QUBO post-processing (multi_ar call, sigmoid labels): weights, the product's annealer (host code,
    shared: it IS the reference's algorithm step for step), per-pixel arg-max among the selected queries, per-instance filtering."""
    from panst3r_amd.engine.postprocess import solve_qubo_simulated_annealing
    Wneg, up, (Hm, Wm) = qubo_weights(mask_pred, true_shape)
    cls = mask_cls[0].float().sigmoid()
    if temperature is not None:
        cls = torch.softmax(cls.sigmoid() / temperature, dim=-1)
    sol, _ = solve_qubo_simulated_annealing(Wneg, redo=num_redo, silent=True)
    sel = torch.from_numpy(np.flatnonzero(np.asarray(sol).astype(bool)))
    cls_probs, cls_ids = cls[sel].max(dim=1)
    pad = torch.zeros(len(up), len(sel), Hm, Wm)
    for v, u in enumerate(up):
        pad[v, :, :u.shape[-2], :u.shape[-1]] = u[sel]
    conf, inst = pad.max(dim=1)                         # [V,Hm,Wm]
    pan = torch.zeros_like(inst)
    segs, new_id = [], 1
    for k in torch.unique(inst).tolist():
        m = inst == k
        mask_conf = float(conf[m].mean())
        if float(cls_probs[k]) * mask_conf < prob_threshold:
            continue
        pan[m] = new_id
        segs.append({'id': new_id, 'query_id': int(k), 'class_prob': float(cls_probs[k]), 'mask_conf': mask_conf, 'category_id': int(cls_ids[k]), 'area': int(m.sum())})
        new_id += 1
    shapes = [u.shape[-2:] for u in up]
    return [{'pan': [pan[i, :h, :w] for i, (h, w) in enumerate(shapes)], 'segments_info': segs, 'conf': [conf[i, :h, :w] for i, (h, w) in enumerate(shapes)]}], Wneg
