"""Pointmap post-processing on the device: the camera recovery of the reference demo (tools/demo_panst3r.py:220-221,
246-277) - activation of the raw pointmaps, focal length per view (`estimate_focal_knowing_depth(..., focal_mode='weiszfeld')`) and
camera-to-world pose per view (`roma.rigid_points_registration(pts3d_local, pts3d, weights=conf - 1)`).

The reference moves every pointmap to the CPU first (`outdevice='cpu'`, 5.5 MB per view), post-processes it there, sends it back to
the GPU and runs ~25 small torch ops per view; here the V views of a scene are three launches (one block per view) and only
V x (1 + 16) numbers come back for the 3x3 Procrustes step.  `must3r.engine.inference.postprocess`, `dust3r.post_process` and `roma`
are un-vendored third-party code: the formulas are the published ones, restated (parity unpinned; oracle/pointmaps.py restates them in
torch and tests/test_pointmaps.py pins both against known-answer cameras).
"""
import numpy as np
import torch

from .. import hip

ACTIVATIONS = {'norm_exp': 0, 'linear': 1}


def postprocess(pointmap, pointmaps_activation='norm_exp'):
    """must3r.engine.inference.postprocess restated: raw [..., H, W, 7] (device, fp32) -> dict(pts3d, pts3d_local [..., H, W, 3], conf [..., H, W])."""
    raw = pointmap.float().contiguous()
    lead = raw.shape[:-1]
    pts, loc = torch.empty(*lead, 3, dtype=torch.float32, device=raw.device), torch.empty(*lead, 3, dtype=torch.float32, device=raw.device)
    conf = torch.empty(*lead, dtype=torch.float32, device=raw.device)
    hip.pointmap_activate(raw, pts, loc, conf, ACTIVATIONS[pointmaps_activation])
    return {'pts3d': pts, 'pts3d_local': loc, 'conf': conf}


def estimate_focal_knowing_depth(pts3d, pp, focal_mode='weiszfeld', min_focal=0.0, max_focal=np.inf, iters=10):
    """dust3r.post_process.estimate_focal_knowing_depth restated: pts3d [B, H, W, 3] (camera frame), pp [B, 2] or [2] (x, y) -> focal [B]."""
    if focal_mode != 'weiszfeld':
        raise NotImplementedError("the demo's mode is 'weiszfeld' (tools/demo_panst3r.py:259)")
    B, H, W, _ = pts3d.shape
    pts = pts3d.float().contiguous()
    ppv = pp.to(device=pts.device, dtype=torch.float32).reshape(-1, 2).expand(B, 2).contiguous()
    focal = hip.focal_weiszfeld(pts, ppv, torch.empty(B, dtype=torch.float32, device=pts.device), H, W, iters)
    base = max(H, W) / (2 * np.tan(np.deg2rad(60) / 2))
    return focal.clip(min=min_focal * base, max=max_focal * base)


def rigid_points_registration(x, y, weights=None, compute_scaling=False):
    """roma.rigid_points_registration restated for batches of point sets: x, y [B, N, 3] (device), weights [B, N] -> (R [B,3,3], t [B,3])
    with y ~ R x + t (weighted Kabsch; the 16 moments per set are reduced on the device, the 3x3 SVD runs on the host in float64)."""
    if compute_scaling:
        raise NotImplementedError('the demo calls it with compute_scaling=False (tools/demo_panst3r.py:265)')
    B, N = x.shape[0], x.shape[1]
    w = torch.ones(B, N, dtype=torch.float32, device=x.device) if weights is None else weights.float().reshape(B, N).contiguous()
    mom = hip.rigid_moments(x.float().contiguous(), y.float().contiguous(), w, torch.empty(B, 16, dtype=torch.float64, device=x.device), 0.0).cpu().numpy()
    R, t = np.zeros((B, 3, 3)), np.zeros((B, 3))
    for b in range(B):
        sw, sx, sy, syx = mom[b, 0], mom[b, 1:4], mom[b, 4:7], mom[b, 7:16].reshape(3, 3)
        xm, ym = sx / sw, sy / sw
        M = syx - sw * np.outer(ym, xm)                      # sum w (y - ym)(x - xm)^T
        U, _, Vt = np.linalg.svd(M)
        D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt)) or 1.0])
        R[b] = U @ D @ Vt                                     # special Procrustes: nearest rotation, det +1
        t[b] = ym - R[b] @ xm
    return torch.from_numpy(R).float(), torch.from_numpy(t).float()


@torch.no_grad()
def cameras_from_pointmaps(pointmaps, true_shape, pointmaps_activation='norm_exp'):
    """The demo's loop (tools/demo_panst3r.py:246-277) for a whole scene: pointmaps = list of raw [1, H, W, 7] (same shape per call group
    is not required); returns (x_out list of dicts, focals list[float], cams2world list of [4,4])."""
    x_out = [postprocess(p[0] if p.dim() == 4 else p, pointmaps_activation) for p in pointmaps]
    focals, cams = [None] * len(x_out), [None] * len(x_out)
    groups = {}
    for i, xo in enumerate(x_out):
        groups.setdefault(tuple(xo['conf'].shape), []).append(i)
    for (H, W), idx in groups.items():
        loc = torch.stack([x_out[i]['pts3d_local'] for i in idx])
        pts = torch.stack([x_out[i]['pts3d'] for i in idx])
        conf = torch.stack([x_out[i]['conf'] for i in idx])
        pp = torch.tensor([[W / 2, H / 2]] * len(idx), dtype=torch.float32, device=loc.device)
        f = estimate_focal_knowing_depth(loc, pp, 'weiszfeld').cpu()
        R, t = rigid_points_registration(loc.reshape(len(idx), -1, 3), pts.reshape(len(idx), -1, 3), weights=conf.reshape(len(idx), -1) - 1.0)
        for j, i in enumerate(idx):
            c2w = torch.eye(4)
            c2w[:3, :3], c2w[:3, 3] = R[j], t[j]
            focals[i], cams[i] = float(f[j]), c2w
    return x_out, focals, cams
