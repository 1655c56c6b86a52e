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


def farthest_point_sampling(dist, N=None, dist_thresh=None, start=None, rng=None):
    """Greedy farthest-point sampling on a distance matrix: returns (indices, distances).
    [3P-recalled, parity unpinned] `must3r.demo.inference.farthest_point_sampling` is imported by the reference (panst3r.py:10, call
    :104 with `1 - sim`, N=K, dist_thresh=None) but not vendored; restated from the published mast3r/must3r retrieval code as recalled:
    first index drawn at random, then repeatedly the point whose distance to the chosen set (min over chosen rows) is largest, stopping
    early when that distance drops below `dist_thresh`.  `start` fixes the first index (the upstream draw makes the reference's result
    depend on numpy's global RNG state); with start=None it is drawn from `rng` (default: numpy's global state, like upstream)."""
    dist = np.asarray(dist)
    if N is None and dist_thresh is None:
        raise ValueError('either N or dist_thresh must be given')
    n = dist.shape[0]
    N = n if N is None else min(int(N), n)
    if start is None:
        start = int((rng or np.random).choice(n))
    indices, distances = [int(start)], [0.0]
    for _ in range(1, N):
        d = dist[indices].min(axis=0)
        best = int(d.argmax())
        if dist_thresh is not None and d[best] < dist_thresh:
            break
        indices.append(best)
        distances.append(float(d[best]))
    return np.array(indices), np.array(distances)


def order_keyframes_by_overlap(sim, anchor_idx):
    """Greedy ordering of the sampled keyframes (reference panst3r.py:105-123; pinned by tests/golden/keyframes_retrieval.npz):
    restrict `sim` to the anchors, zero the diagonal, start with the anchor of highest total similarity, then repeatedly append
    the not-yet-chosen anchor with the highest similarity to ANY chosen one (first maximum in row-major order of the
    chosen-rows x all-columns block, chosen columns zeroed).  Returns view indices in memory-build order."""
    anchor_idx = [int(a) for a in anchor_idx]
    K = len(anchor_idx)
    s = np.array(sim, dtype=np.asarray(sim).dtype)[anchor_idx, :][:, anchor_idx]
    s[np.arange(K), np.arange(K)] = 0
    chosen = [int(np.argmax(s.sum(axis=-1)))]
    s[:, chosen[0]] = 0
    while len(chosen) != K:
        block = s[np.array(chosen)]
        nxt = int(np.unravel_index(np.argmax(block), block.shape)[1])
        chosen.append(nxt)
        s[:, nxt] = 0
    return [anchor_idx[k] for k in chosen]


def keyframes_from_similarity(sim, num_keyframes, start=None, rng=None):
    """Keyframes by retrieval given the V x V image-similarity matrix (reference `PanSt3R._get_keyframes_retrieval`,
    panst3r.py:88-125, minus the ASMK retriever that produces `sim`: asmk / faiss are outside this build).
    Farthest-point sampling on `1 - sim` picks K spread-out views, `order_keyframes_by_overlap` orders them so that each new
    keyframe overlaps the memory built so far.  NOTE (reference quirk): when `sim` has all-zero rows among the anchors the
    argmax falls back to column 0, which can repeat an anchor; the reference then loops with a duplicate - we raise instead."""
    sim = np.asarray(sim)
    if sim.ndim != 2 or sim.shape[0] != sim.shape[1]:
        raise ValueError('similarity matrix must be square, got %s' % (sim.shape,))
    K = sim.shape[0] if (num_keyframes is None or num_keyframes > sim.shape[0]) else int(num_keyframes)
    anchors, _ = farthest_point_sampling(1 - sim, N=K, dist_thresh=None, start=start, rng=rng)
    if len(set(anchors.tolist())) != len(anchors):
        raise ValueError('farthest-point sampling repeated a view (%s): every view must be most similar to itself (diagonal = row maximum)'
                         % (anchors.tolist(),))
    out = order_keyframes_by_overlap(sim, anchors)
    if len(set(out)) != len(out):
        raise ValueError('degenerate similarity matrix: the greedy ordering repeated a keyframe (%s)' % (out,))
    return out
