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
    if label_mode != 'sigmoid':
        raise NotImplementedError("released configs use label_mode='sigmoid' (configs/base.yaml:24)")
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
