"""Panoptic post-processing on the GPU -- host mirror of reference engine/postprocess.py.

`panoptic_inference_v2` keeps the reference's signature and result structure (postprocess.py:14-16,125-130) but runs on
the device the masks already live on (the demo moves 39 MB of mask logits per view to the CPU and post-processes there,
tools/demo_panst3r.py:233-242).  The kernels are in csrc/postprocess.hip behind `pst_pp_*` (include/panst3r_hip.h); there
is no CPU fallback: CPU inputs are uploaded to `device`, which must be a GPU.
"""
import torch

from .. import hip


def panoptic_inference_v1(*args, mask_threshold=0.5, overlap_threshold=0.8, **kwargs):
    """reference engine/postprocess.py:9-11: the Mask2Former-style single round (niters=1) with its own thresholds"""
    return panoptic_inference_v2(*args, mask_threshold=mask_threshold, overlap_threshold=overlap_threshold, niters=1, **kwargs)


@torch.no_grad()
def panoptic_inference_v2(mask_cls, mask_pred, true_shape, label_mode='sigmoid', cls_threshold=0.1, temperature=None,
                          mask_threshold=0.25, overlap_threshold=0.5, niters=2, void_confidence=0.1, device=None, multi_ar=False):
    """mask_cls [1,Q,Ncls] class logits; mask_pred: list[V] of [1,Q,h,w] mask logits (multi_ar=True, the demo's call) or one
    [V,Q,h,w] / [1,V,Q,h,w] tensor; true_shape [V,2] (H, W) per view, or one (H, W) for a same-shape stack.
    Returns [{'pan': int32 maps, 'segments_info': [{'id','query_id','category_id'}], 'conf': fp32 maps}] with per-view
    lists for multi_ar=True (postprocess.py:121-123) and stacked [V,H,W] tensors otherwise; maps stay on the device."""
    if label_mode not in ('sigmoid', 'softmax'):
        raise ValueError("label_mode must be 'sigmoid' or 'softmax' (engine/postprocess.py:40-51), got %r" % (label_mode,))
    if isinstance(mask_pred, torch.Tensor):
        mp = mask_pred[0] if mask_pred.dim() == 5 else mask_pred
        views = [mp[i] for i in range(mp.shape[0])]
    else:
        views = [m[0] if m.dim() == 4 else m for m in mask_pred]
    V = len(views)
    if device is None:
        device = views[0].device
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('panst3r_amd.postprocess runs on the GPU only (got device=%s); there is no CPU fallback' % device)
    ts = torch.as_tensor(true_shape).reshape(-1, 2).tolist()
    shapes = [tuple(int(v) for v in (ts[i] if len(ts) == V else ts[0])) for i in range(V)]
    if mask_cls.shape[0] != 1:
        raise NotImplementedError('one scene per call (B == 1)')
    logits = mask_cls[0].to(device=device, dtype=torch.float32).contiguous()
    Q = logits.shape[0]
    views = [m.to(device=device, dtype=torch.float32).contiguous() for m in views]
    i32 = dict(dtype=torch.int32, device=device)
    scores = torch.empty(Q, dtype=torch.float32, device=device)
    labels, keep = torch.empty(Q, **i32), torch.empty(Q, **i32)
    cnt_orig, cnt_mask, seg_id = torch.zeros(Q, **i32), torch.zeros(Q, **i32), torch.zeros(Q, **i32)
    if label_mode == 'softmax':         # the last class column is "no object" (:48-51); the reference does not read the temperature in this mode
        hip.pp_scores_softmax(logits, cls_threshold, scores, labels, keep)
    else:
        hip.pp_scores(logits, cls_threshold, temperature, scores, labels, keep)
    fused = [hip.pp_fused_fits(Q, m.shape[-2], m.shape[-1], shapes[i][0], shapes[i][1]) for i, m in enumerate(views)]
    probs = None
    if not all(fused):     # strong down-sampling: the tile footprint does not fit in LDS -> probability scratch, reused per view
        probs = torch.empty(Q * max(m.shape[-2] * m.shape[-1] for m in views), dtype=torch.float32, device=device)
    best_q = [torch.empty(h * w, **i32) for h, w in shapes]
    best_m = [torch.empty(h * w, dtype=torch.float32, device=device) for h, w in shapes]
    for _ in range(max(int(niters), 1)):
        for i, m in enumerate(views):
            hm, wm = m.shape[-2:]
            if fused[i]:
                hip.pp_argmax_logits(m, scores, keep, Q, hm, wm, shapes[i][0], shapes[i][1], mask_threshold, best_q[i], best_m[i], cnt_orig,
                                     cnt_mask)
            else:
                hip.pp_sigmoid(m, keep, probs, Q, hm * wm)
                hip.pp_argmax(probs, scores, keep, Q, hm, wm, shapes[i][0], shapes[i][1], mask_threshold, best_q[i], best_m[i], cnt_orig,
                              cnt_mask)
        hip.pp_select(keep, cnt_orig, cnt_mask, Q, overlap_threshold, keep, seg_id)        # keep <- this round's selection
    pan, conf = [], []
    for i, (h, w) in enumerate(shapes):
        p, c = torch.empty(h, w, **i32), torch.empty(h, w, dtype=torch.float32, device=device)
        hip.pp_finalize(best_q[i], best_m[i], seg_id, h * w, mask_threshold, void_confidence, p, c)
        pan.append(p)
        conf.append(c)
    ids, lab = seg_id.cpu().tolist(), labels.cpu().tolist()                                  # the only host sync
    segments = [{'id': ids[q], 'query_id': q, 'category_id': lab[q]} for q in range(Q) if ids[q] > 0]
    if not multi_ar:
        pan, conf = torch.stack(pan), torch.stack(conf)
    return [{'pan': pan, 'segments_info': segments, 'conf': conf}]


# ------------------------------------------------------------------------------------------------ QUBO (reference engine/postprocess.py:135-336)
def qubo_energy(alpha, W, lambda_reg):
    """E = alpha^T W alpha + lambda * mean(alpha)  (:262-264)"""
    return alpha.dot(W).dot(alpha) + lambda_reg * alpha.mean()


def solve_qubo_simulated_annealing(W, num_iters=10000, T0=0.5, T_end=1e-4, lambda_reg=1e-3, redo=20, random_init=True, silent=False):
    """The reference's annealer (:266-336), host numpy like the reference ("Optimization done on CPU", :176): same moves, same acceptance
    rule and the same sequence of np.random draws (randint(0, 2, N) per restart; per iteration randint(N), and rand() only when the move
    does not lower the energy), so a seeded run reproduces the reference's trajectory.  Incremental energy (2 x row dot instead of the
    reference's full re-evaluation, its commented-out variant :311-316) is NOT used: bit-compatible acceptance decisions matter more here
    than the 200 x 200 matvec."""
    import numpy as np
    cooling_rate = (T_end / T0) ** (1 / num_iters)
    N = W.shape[0]
    best_x2, best_energy2 = None, float('inf')
    for _ in range(redo):
        x = np.random.randint(0, 2, size=N) if random_init else np.zeros(N)
        best_x = np.copy(x)
        best_energy = current_energy = qubo_energy(x, W, lambda_reg)
        T = T0
        for _i in range(num_iters):
            j = np.random.randint(N)
            new_x = np.copy(x)
            new_x[j] = 1 - new_x[j]
            new_energy = qubo_energy(new_x, W, lambda_reg)
            delta = new_energy - current_energy
            if delta < 0 or np.random.rand() < np.exp(-delta / T):
                x = new_x
                current_energy = new_energy
                if current_energy < best_energy:
                    best_energy = current_energy
                    best_x = np.copy(x)
            T *= cooling_rate
        if best_energy < best_energy2:
            best_energy2, best_x2 = best_energy, best_x
    return best_x2, best_energy2


@torch.no_grad()
def qubo_weights(views, shapes, device, penalty=1):
    """`weight_from_masks` (:229-259) with the O(Q^2 x pixels) sums on the GPU: views = list of mask logits [Q,h,w], shapes = true (H, W)
    per view.  Returns -W as float32 numpy [Q, Q] (what the reference hands to the annealer): diagonal = mask areas, off-diagonal =
    -(1 + penalty) overlap / 2, normalised by the padded image size and the number of views."""
    import numpy as np
    Q = views[0].shape[0]
    Wacc = torch.zeros(Q, Q, dtype=torch.float64, device=device)
    for m, (H, W) in zip(views, shapes):
        probs = torch.empty(Q, H * W, dtype=torch.float32, device=device)
        hip.qubo_upsample(m, probs, Q, m.shape[-2], m.shape[-1], H, W)
        hip.qubo_overlap(probs, Q, H * W, Wacc)
        del probs
    S = Wacc.cpu().numpy()
    Hm, Wm = max(s[0] for s in shapes), max(s[1] for s in shapes)          # the reference pads every view to the largest shape with zeros (:141)
    Wt = -(1 + penalty) * S / 2
    np.fill_diagonal(Wt, np.diag(S))
    Wt = Wt / (Hm * Wm) / len(views)
    return (-Wt).astype(np.float32)


@torch.no_grad()
def panoptic_inference_qubo(mask_cls, mask_pred, true_shape, label_mode='sigmoid', temperature=None, device='cuda', num_redo=20, prob_threshold=0.01,
                            silent=False, multi_ar=False):
    """Reference signature (engine/postprocess.py:135): QUBO selection of a maximal set of non-overlapping masks, then per-pixel arg-max among
    the selected queries.  Pixel-sized work runs on `device` (a GPU); the annealer and the per-instance bookkeeping are host code as in the
    reference.  Result structure as the reference's (:206-217): 'pan' / 'conf' per view for multi_ar, stacked otherwise; `query_id` is the
    index among the SELECTED queries, exactly as the reference reports it (:202)."""
    import numpy as np
    if label_mode != 'sigmoid':         # the reference itself cannot run this combination: :166-167 reads `cur_mask_cls` before any assignment (NameError)
        raise NotImplementedError("panoptic_inference_qubo with label_mode='softmax' fails in the reference too (engine/postprocess.py:166-167); "
                                  "use panoptic_inference_v2 / v1 for softmax-label models")
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError('panst3r_amd.postprocess runs on the GPU only (got device=%s); there is no CPU fallback' % device)
    if isinstance(mask_pred, torch.Tensor):
        mp = mask_pred[0] if mask_pred.dim() == 5 else mask_pred
        views = [mp[i] for i in range(mp.shape[0])]
    else:
        views = [m[0] if m.dim() == 4 else m for m in mask_pred]
    V = len(views)
    ts = torch.as_tensor(np.asarray(true_shape)).reshape(-1, 2).tolist()
    shapes = [tuple(int(v) for v in (ts[i] if len(ts) == V else ts[0])) for i in range(V)]
    if mask_cls.shape[0] != 1:
        raise NotImplementedError('one scene per call (B == 1)')
    views = [m.to(device=device, dtype=torch.float32).contiguous() for m in views]
    Q = views[0].shape[0]
    cls = mask_cls[0].float().cpu().sigmoid()
    if temperature is not None:
        cls = torch.softmax(cls.sigmoid() / temperature, dim=-1)              # as written in the reference (:158-160)
    Wneg = qubo_weights(views, shapes, device)
    solution, _ = solve_qubo_simulated_annealing(Wneg, redo=num_redo, silent=silent)
    sel = np.flatnonzero(np.asarray(solution).astype(bool))
    cls_probs, cls_ids = cls[torch.from_numpy(sel)].max(dim=1)
    sel_dev = torch.from_numpy(sel.astype(np.int32)).to(device)
    nsel = len(sel)
    Hm, Wm = max(s[0] for s in shapes), max(s[1] for s in shapes)
    confs, insts = [], []
    cnt = torch.zeros(nsel, dtype=torch.float64, device=device)
    csum = torch.zeros(nsel, dtype=torch.float64, device=device)
    for m, (H, W) in zip(views, shapes):
        probs = torch.empty(Q, H * W, dtype=torch.float32, device=device)
        hip.qubo_upsample(m, probs, Q, m.shape[-2], m.shape[-1], H, W)
        conf, inst = torch.empty(H * W, dtype=torch.float32, device=device), torch.empty(H * W, dtype=torch.int32, device=device)
        hip.qubo_argmax(probs, sel_dev, H * W, conf, inst)
        idx = inst.long()
        cnt += torch.bincount(idx, minlength=nsel).double()
        csum += torch.bincount(idx, weights=conf.double(), minlength=nsel)
        confs.append(conf.view(H, W)); insts.append(inst.view(H, W))
        del probs
    cnt[0] += sum(Hm * Wm - H * W for H, W in shapes)          # the reference's zero padding (:141): all masks 0 there -> arg-max index 0, confidence 0
    cnt_h, csum_h = cnt.cpu().numpy(), csum.cpu().numpy()
    remap = np.zeros(nsel, dtype=np.int32)
    segments_info, new_id = [], 1
    for k in range(nsel):
        if cnt_h[k] == 0:
            continue                                         # not in torch.unique(instance_ids)
        mask_conf = float(csum_h[k] / cnt_h[k])
        if float(cls_probs[k]) * mask_conf < prob_threshold:
            continue
        remap[k] = new_id
        segments_info.append({'id': new_id, 'query_id': int(k), 'class_prob': float(cls_probs[k]), 'mask_conf': mask_conf,
                              'category_id': cls_ids[k], 'area': int(cnt_h[k])})
        new_id += 1
    remap_dev = torch.from_numpy(remap).to(device)
    pan = [remap_dev[i.long()].to(torch.int64) for i in insts]
    if not multi_ar:
        pan, confs = torch.stack(pan), torch.stack(confs)
    return [{'pan': pan, 'segments_info': segments_info, 'conf': confs}]
