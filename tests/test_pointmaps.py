"""pointmap post-processing.  The oracle (oracle/pointmaps.py, restated third-party formulas) is pinned by known-answer
cameras on the CPU; the HIP path (panst3r_amd.engine.pointmaps) is compared with the oracle on the GPU."""
import numpy as np
import pytest
import torch


def synth_view(H, W, focal, seed, R=None, t=None, noise=0.0):
    """a pinhole camera looking at a random depth map: local points, their world positions under (R, t), confidences"""
    g = np.random.Generator(np.random.PCG64(seed))
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    z = (1.5 + g.random((H, W)) * 2.0).astype(np.float32)
    loc = np.stack([(xs - W / 2) / focal * z, (ys - H / 2) / focal * z, z], -1)
    if R is None:
        a = g.standard_normal(3); a /= np.linalg.norm(a); th = 0.7
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
        t = g.standard_normal(3)
    pts = loc @ R.T.astype(np.float32) + t.astype(np.float32) + noise * g.standard_normal((H, W, 3)).astype(np.float32)
    conf = (1.0 + g.random((H, W)) * 5).astype(np.float32)
    return torch.from_numpy(loc), torch.from_numpy(pts.astype(np.float32)), torch.from_numpy(conf), torch.from_numpy(R).float(), torch.from_numpy(t).float()


def test_oracle_known_answers():
    from oracle import pointmaps as O
    H, W, f = 48, 64, 55.0
    loc, pts, conf, R, t = synth_view(H, W, f, 1)
    est = O.estimate_focal_knowing_depth(loc[None], torch.tensor([W / 2, H / 2]))
    assert abs(float(est) - f) / f < 1e-4
    Rr, tr = O.rigid_points_registration(loc.reshape(-1, 3), pts.reshape(-1, 3), weights=conf.ravel() - 1.0)
    assert float((Rr - R).abs().max()) < 1e-5 and float((tr - t).abs().max()) < 1e-4
    raw = torch.cat([pts, loc, conf[..., None]], -1)
    out = O.postprocess(raw)
    d = pts.norm(dim=-1, keepdim=True)
    assert torch.allclose(out['pts3d'], pts / d * torch.expm1(d), rtol=1e-6) and torch.allclose(out['conf'], 1 + conf.exp())
    # outliers: the robust (L1) focal resists 10 % corrupted points, the closed-form L2 start does not
    bad = loc.clone()
    bad.view(-1, 3)[::10, :2] *= 4.0
    assert abs(float(O.estimate_focal_knowing_depth(bad[None], torch.tensor([W / 2, H / 2]))) - f) / f < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize('H,W', [(48, 64), (384, 512), (160, 96)])
def test_hip_pointmap_postprocess_matches_oracle(H, W):
    from oracle import pointmaps as O
    from panst3r_amd.engine import pointmaps as Pm
    dev = 'cuda:0'
    views = [synth_view(H, W, 40.0 + 7 * i, 10 + i, noise=1e-3) for i in range(3)]
    raw = torch.stack([torch.cat([v[1], v[0], (v[2] - 1).log()[..., None]], -1) for v in views])      # conf = 1 + exp(c)
    lin = Pm.postprocess(raw.to(dev), 'linear')
    assert torch.equal(lin['pts3d'].cpu(), raw[..., :3]) and torch.equal(lin['pts3d_local'].cpu(), raw[..., 3:6])
    small = raw.clone()
    small[..., :6] *= 0.15                                   # norm_exp: expm1(|xyz|) -- keep the raw vectors in the range the decoder emits
    ne, ne_o = Pm.postprocess(small.to(dev), 'norm_exp'), O.postprocess(small, 'norm_exp')
    for k in ('pts3d', 'pts3d_local', 'conf'):
        assert torch.allclose(ne[k].cpu(), ne_o[k], rtol=1e-5, atol=1e-6), k
    loc = torch.stack([v[0] for v in views]); pts = torch.stack([v[1] for v in views]); conf = torch.stack([v[2] for v in views])
    pp = torch.tensor([W / 2, H / 2])
    f_h = Pm.estimate_focal_knowing_depth(loc.to(dev), pp).cpu()
    f_o = O.estimate_focal_knowing_depth(loc, pp)
    assert torch.allclose(f_h, f_o, rtol=1e-5), (f_h, f_o)
    R_h, t_h = Pm.rigid_points_registration(loc.reshape(3, -1, 3).to(dev), pts.reshape(3, -1, 3).to(dev), weights=conf.reshape(3, -1).to(dev) - 1.0)
    for i, v in enumerate(views):
        R_o, t_o = O.rigid_points_registration(loc[i].reshape(-1, 3), pts[i].reshape(-1, 3), weights=conf[i].ravel() - 1.0)
        assert float((R_h[i] - R_o).abs().max()) < 2e-5 and float((t_h[i] - t_o).abs().max()) < 2e-4
        assert float((R_h[i] - v[3]).abs().max()) < 1e-3
    # scene-level helper = the demo's loop (tools/demo_panst3r.py:246-277)
    x_out, focals, cams = Pm.cameras_from_pointmaps([r[None].to(dev) for r in lin_raw(raw)], None, 'linear')
    assert len(focals) == 3 and abs(focals[1] - 47.0) / 47.0 < 1e-2 and cams[0].shape == (4, 4)


def lin_raw(raw):
    return [raw[i] for i in range(raw.shape[0])]
