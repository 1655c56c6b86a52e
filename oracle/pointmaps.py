"""TEST INFRASTRUCTURE (oracle/): CPU restatement of the pointmap post-processing the reference demo calls (tools/demo_panst3r.py:220-221,
246-277).  All three functions live in un-vendored third-party packages (must3r, dust3r, roma): restated from their published code,
parity unpinned; pinned here by known-answer cameras (tests/test_pointmaps.py)."""
import numpy as np
import torch


def postprocess(pointmap, pointmaps_activation='norm_exp'):
    """must3r.engine.inference.postprocess: raw [..., 7] -> pts3d, pts3d_local (norm_exp: xyz / d * expm1(d)), conf = 1 + exp(c)."""
    def act(xyz):
        if pointmaps_activation == 'linear':
            return xyz
        d = xyz.norm(dim=-1, keepdim=True)
        return xyz / d.clip(min=1e-8) * torch.expm1(d)
    return {'pts3d': act(pointmap[..., 0:3]), 'pts3d_local': act(pointmap[..., 3:6]), 'conf': 1.0 + pointmap[..., 6].exp()}


def estimate_focal_knowing_depth(pts3d, pp, focal_mode='weiszfeld', min_focal=0.0, max_focal=np.inf):
    """dust3r.post_process.estimate_focal_knowing_depth (weiszfeld): focal = argmin sum |pixel - focal (x, y) / z|."""
    B, H, W, _ = pts3d.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    pixels = torch.stack([xs, ys], -1).view(1, -1, 2) - pp.view(-1, 1, 2)
    pts = pts3d.flatten(1, 2)
    xy_over_z = (pts[..., :2] / pts[..., 2:3]).nan_to_num(posinf=0, neginf=0)
    dot_xy_px = (xy_over_z * pixels).sum(dim=-1)
    dot_xy_xy = xy_over_z.square().sum(dim=-1)
    focal = dot_xy_px.mean(dim=1) / dot_xy_xy.mean(dim=1)
    for _ in range(10):
        dis = (pixels - focal.view(-1, 1, 1) * xy_over_z).norm(dim=-1)
        w = dis.clip(min=1e-8).reciprocal()
        focal = (w * dot_xy_px).mean(dim=1) / (w * dot_xy_xy).mean(dim=1)
    base = max(H, W) / (2 * np.tan(np.deg2rad(60) / 2))
    return focal.clip(min=min_focal * base, max=max_focal * base)


def rigid_points_registration(x, y, weights=None, compute_scaling=False):
    """roma.rigid_points_registration: weighted Kabsch, y ~ R x + t; x, y [N, 3]."""
    assert not compute_scaling
    x, y = x.double(), y.double()
    w = torch.ones(x.shape[0], dtype=torch.float64) if weights is None else weights.double()
    w = w[:, None]
    xm, ym = (w * x).sum(0) / w.sum(), (w * y).sum(0) / w.sum()
    M = ((y - ym) * w).T @ (x - xm)
    U, _, Vt = torch.linalg.svd(M)
    D = torch.diag(torch.tensor([1.0, 1.0, float(torch.sign(torch.linalg.det(U @ Vt)))], dtype=torch.float64))
    R = U @ D @ Vt
    return R.float(), (ym - R @ xm).float()
