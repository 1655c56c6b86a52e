"""Keyframe schedule of the scene pass (reference panst3r.py:65-70, :183-196).

Pure host arithmetic; pinned by tests/golden/keyframes.npz.
"""
import numpy as np


def select_keyframes(n_views, num_keyframes):
    """`np.linspace(0, N-1, K, dtype=int)` (panst3r.py:186); all views when K is None or K > N (:183-184).
    The harness clamps 2 <= K (the demo does max(K, 2), tools/demo_panst3r.py:230)."""
    if num_keyframes is None or num_keyframes > n_views:
        return list(range(n_views))
    return np.linspace(0, n_views - 1, num_keyframes, dtype=int).tolist()


def view_order(n_views, keyframes):
    """keyframes first, then the remaining views in ascending order (panst3r.py:188-192); returns (order, inverse)."""
    rest = sorted(set(range(n_views)) - set(keyframes))
    order = list(keyframes) + rest
    assert len(order) == n_views
    return order, np.argsort(order).tolist()


def mem_batches(n_imgs, init_num_views=2, batch_num_views=1):
    """[2,1,1,...] (panst3r.py:35-39,65-70).  n_imgs < 2 is rejected (the reference loops forever / goes negative)."""
    if n_imgs < init_num_views:
        raise ValueError('need at least %d keyframes, got %d' % (init_num_views, n_imgs))
    out = [init_num_views]
    while sum(out) != n_imgs:
        out.append(min(batch_num_views, n_imgs - sum(out)))
    return out
