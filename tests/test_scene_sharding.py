"""N>1 path on CPU: the view-sharded scene plan (panst3r_amd/scene.py) over a world_size-2 gloo group, driven by the
oracle backend.  Checks (1) the plan's algebra (mean4-then-dot attention masks, heads once per scene) against the
reference formulation, (2) sharded == unsharded, (3) the collective plumbing for uneven keyframe counts."""
import os
import socket
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import rel_l2
import tiny
from oracle_backend import OracleBackend
from panst3r_amd.scene import run_scene, assign_views, gather_keyframe_rows

H, W = 64, 96


def _scene(variant, V, K, rank=0, world=1, group=None, plan='replicated', minmax_bs=1):
    torch.set_num_threads(2)
    model = tiny.build(tiny.OracleNS, variant)
    imgs = tiny.images(V, H, W)
    if plan == 'broadcast_whole':          # the bank in ONE broadcast behind the whole build (stream_bank=False) instead of one per memory update
        plan, stream = 'broadcast', False
    else:
        stream = True
    with torch.no_grad():
        return run_scene(OracleBackend(model), lambda i: imgs[i], V, H, W, K, tiny.NAMES, rank, world, group, plan=plan, minmax_bs=minmax_bs, stream_bank=stream)


def test_assign_views():
    kf, order, owner = assign_views(50, 16, 8)
    assert kf == [0, 3, 6, 9, 13, 16, 19, 22, 26, 29, 32, 35, 39, 42, 45, 49]
    assert sorted(order) == list(range(50)) and order[:16] == kf
    assert max(owner.count(r) for r in range(8)) - min(owner.count(r) for r in range(8)) <= 1
    assert [owner[i] for i in range(16)] == [i % 8 for i in range(16)]
    # plan='broadcast' (SURVEY 8(e) option 2): rank 0 builds the memory and owns nothing but its keyframes
    kf2, order2, owner2 = assign_views(50, 16, 8, plan='broadcast')
    assert kf2 == kf and order2 == order and owner2[:16] == owner[:16]
    assert owner2.count(0) == 2 and all(r != 0 for r in owner2[16:])
    assert max(owner2.count(r) for r in range(1, 8)) - min(owner2.count(r) for r in range(1, 8)) <= 1
    assert assign_views(50, 16, 1, plan='broadcast')[2] == [0] * 50
    with pytest.raises(ValueError):
        assign_views(50, 16, 8, plan='ring')


def test_auto_plan_never_leaves_a_rank_without_views():
    """plan='auto' resolves BEFORE the K >= world guard: with fewer keyframes than ranks it falls back to 'replicated' (every rank
    owns >= 1 view), an explicit 'broadcast' in that situation raises on every rank, and no constructor returns a runner whose rank is idle."""
    from panst3r_amd.scene import SceneRunner, resolve_plan
    assert resolve_plan('auto', 8, 16) == 'broadcast' and resolve_plan('auto', 8, 4) == 'replicated' and resolve_plan('auto', 2, 16) == 'replicated'
    model = tiny.build(tiny.OracleNS, 'v1')
    imgs = {i: im for i, im in enumerate(tiny.images(9, H, W))}
    for rank in range(8):
        r = SceneRunner(OracleBackend(model), imgs, 9, H, W, 4, tiny.NAMES, rank=rank, world=8, plan='auto')
        assert r.plan == 'replicated' and r.n_local >= 1
        with pytest.raises(ValueError):
            SceneRunner(OracleBackend(model), imgs, 9, H, W, 4, tiny.NAMES, rank=rank, world=8, plan='broadcast')


@pytest.mark.parametrize('variant', ['v1', 'v2'])
def test_plan_matches_reference_formulation(variant):
    """run_scene(world=1) with the oracle backend == the oracle pipeline that follows reference panst3r.py:169-284."""
    V, K = 4, 3
    model = tiny.build(tiny.OracleNS, variant)
    imgs = tiny.images(V, H, W)
    pm_ref, pan_ref = model.forward_inference_multi_ar(imgs, torch.tensor([[H, W]] * V), tiny.NAMES, num_keyframes=K, max_bs=1)
    res, scene = _scene(variant, V, K)
    assert rel_l2(scene['out_queries'], pan_ref['out_queries']) < 1e-4
    assert rel_l2(scene['pred_logits'], pan_ref['pred_logits']) < 1e-4
    for i in range(V):
        assert rel_l2(res[i][0], pm_ref[i]) < 1e-5
        assert rel_l2(res[i][1], pan_ref['pred_masks'][i]) < 1e-4


@pytest.mark.parametrize('max_bs', [None, 2])
def test_minmax_scope_follows_max_bs(max_bs):
    """LoftUp's MinMaxScaler pools min / max over the chunk of views the reference hands it (loftup.py:14-19; stack_views + batched_map chunk by
    max_bs, panst3r.py:212-216,257-261): run_scene(minmax_bs=...) == the oracle pipeline with the same max_bs, on a multi-aspect-ratio scene
    (scopes never cross shapes, nor the keyframe / other-view boundary) - and it is NOT what per-view scaling gives."""
    shapes = [(64, 96), (32, 96), (64, 96), (64, 96), (32, 96), (64, 96), (64, 96)]
    V, K = len(shapes), 4
    model = tiny.build(tiny.OracleNS, 'v2')
    imgs = [tiny.synth_image(i, a, b, 7) for i, (a, b) in enumerate(shapes)]
    pm_ref, pan_ref = model.forward_inference_multi_ar(imgs, torch.tensor(shapes), tiny.NAMES, num_keyframes=K, max_bs=max_bs)
    per_view = model.forward_inference_multi_ar(imgs, torch.tensor(shapes), tiny.NAMES, num_keyframes=K, max_bs=1)[1]
    with torch.no_grad():
        res, scene = run_scene(OracleBackend(model), lambda i: imgs[i], V, None, None, K, tiny.NAMES, shapes=shapes, minmax_bs=max_bs)
    assert rel_l2(scene['out_queries'], pan_ref['out_queries']) < 1e-4
    for i in range(V):
        assert rel_l2(res[i][1], pan_ref['pred_masks'][i]) < 1e-4, i
    assert max(rel_l2(res[i][1], per_view['pred_masks'][i]) for i in range(V)) > 1e-2          # the scope matters


@pytest.mark.parametrize('variant', ['v1', 'v2'])
def test_explicit_keyframe_order_matches_reference_formulation(variant):
    """Keyframes as the reference's retrieval mode hands them over (panst3r.py:179-180): an unsorted list in memory-build order."""
    V, kf = 5, [3, 0, 4]
    model = tiny.build(tiny.OracleNS, variant)
    imgs = tiny.images(V, H, W)
    pm_ref, pan_ref = model.forward_inference_multi_ar(imgs, torch.tensor([[H, W]] * V), tiny.NAMES, num_keyframes=3, use_retrieval=True, keyframes=kf, max_bs=1)
    with torch.no_grad():
        res, scene = run_scene(OracleBackend(model), lambda i: imgs[i], V, H, W, 3, tiny.NAMES, keyframes=kf)
    assert rel_l2(scene['out_queries'], pan_ref['out_queries']) < 1e-4
    for i in range(V):
        assert rel_l2(res[i][0], pm_ref[i]) < 1e-5
        assert rel_l2(res[i][1], pan_ref['pred_masks'][i]) < 1e-4
    lin = model.forward_inference_multi_ar(imgs, torch.tensor([[H, W]] * V), tiny.NAMES, num_keyframes=3, max_bs=1)[1]
    assert rel_l2(lin['out_queries'], pan_ref['out_queries']) > 1e-3            # a different memory than the linspace choice [0, 2, 4]
    kfs, order, owner = assign_views(V, 3, 2, kf)
    assert kfs == kf and order == [3, 0, 4, 1, 2] and owner == [0, 1, 0, 1, 0]
    for bad in ([0], [0, 0], [0, 7]):
        with pytest.raises(ValueError):
            assign_views(V, 3, 1, bad)


MULTI_AR = [(64, 96), (32, 96), (64, 96), (64, 64), (32, 96)]


def _multi_ar_images():
    return [tiny.synth_image(i, h, w, 7) for i, (h, w) in enumerate(MULTI_AR)]


@pytest.mark.parametrize('variant,K', [('v1', 3), ('v2', 5)])
def test_multi_aspect_ratio_scene_matches_reference_formulation(variant, K):
    """Views of different (landscape) shapes: the grouped scene plan == the oracle pipeline that walks the views one by
    one like reference panst3r.py:169-284 (K=5: the first two keyframes already differ in shape)."""
    V = len(MULTI_AR)
    model = tiny.build(tiny.OracleNS, variant)
    imgs = _multi_ar_images()
    pm_ref, pan_ref = model.forward_inference_multi_ar(imgs, torch.tensor(MULTI_AR), tiny.NAMES, num_keyframes=K, max_bs=1)
    with torch.no_grad():
        res, scene = run_scene(OracleBackend(model), lambda i: imgs[i], V, None, None, K, tiny.NAMES, shapes=MULTI_AR)
    assert rel_l2(scene['out_queries'], pan_ref['out_queries']) < 1e-4
    for i in range(V):
        assert res[i][0].shape == pm_ref[i].shape and res[i][1].shape == pan_ref['pred_masks'][i].shape
        assert rel_l2(res[i][0], pm_ref[i]) < 1e-5
        assert rel_l2(res[i][1], pan_ref['pred_masks'][i]) < 1e-4


PORTRAIT_AR = [(96, 64), (64, 96), (96, 64), (96, 32), (64, 96)]


@pytest.mark.parametrize('variant,K', [('v1', 3), ('v2', 5), ('v2', 2)])
def test_portrait_scene_matches_reference_formulation(variant, K):
    """Portrait views (native orientation, true_shape = tensor shape as in tools/demo_panst3r.py:107) mixed with
    landscape ones: DINOv2 runs on the transposed image (model/dino.py:15-47), the upscaler results are handed back
    transposed (utils.py:47-49), the key PE is that of the transposed grid (mask_transformer.py:106-119) and, for LoftUp,
    the attention-mask resize of the native masks to the transposed key grid is anisotropic (:283-287)."""
    V = len(PORTRAIT_AR)
    model = tiny.build(tiny.OracleNS, variant)
    imgs = [tiny.synth_image(i, h, w, 11) for i, (h, w) in enumerate(PORTRAIT_AR)]
    pm_ref, pan_ref = model.forward_inference_multi_ar(imgs, torch.tensor(PORTRAIT_AR), tiny.NAMES, num_keyframes=K, max_bs=1)
    with torch.no_grad():
        res, scene = run_scene(OracleBackend(model), lambda i: imgs[i], V, None, None, K, tiny.NAMES, shapes=PORTRAIT_AR)
    assert rel_l2(scene['out_queries'], pan_ref['out_queries']) < 1e-4
    assert rel_l2(scene['pred_logits'], pan_ref['pred_logits']) < 1e-4
    for i in range(V):
        assert res[i][0].shape == pm_ref[i].shape and res[i][1].shape == pan_ref['pred_masks'][i].shape
        assert rel_l2(res[i][0], pm_ref[i]) < 1e-5
        assert rel_l2(res[i][1], pan_ref['pred_masks'][i]) < 1e-4


def _scene_kind(kind):
    """(shapes, images) of the mixed-shape scenes used by the 2-rank tests."""
    if kind == 'multi_ar':
        return MULTI_AR, _multi_ar_images()
    return PORTRAIT_AR, [tiny.synth_image(i, h, w, 11) for i, (h, w) in enumerate(PORTRAIT_AR)]


def _worker(rank, world, port, variant, V, K, q, plan='replicated', minmax_bs=1):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        if V in ('multi_ar', 'portrait'):
            torch.set_num_threads(2)
            model = tiny.build(tiny.OracleNS, variant)
            shapes, imgs = _scene_kind(V)
            with torch.no_grad():
                res, scene = run_scene(OracleBackend(model), lambda i: imgs[i], len(shapes), None, None, K, tiny.NAMES, rank, world, None,
                                       shapes=shapes, plan=plan, minmax_bs=minmax_bs)
        else:
            res, scene = _scene(variant, V, K, rank, world, None, plan, minmax_bs)
        t = torch.arange(6, dtype=torch.bfloat16).reshape(3, 2) + 10 * rank if rank == 0 else torch.arange(4, dtype=torch.bfloat16).reshape(2, 2) + 10
        g = gather_keyframe_rows(t, 5, 1, rank, world, None)                # K=5 dealt 3/2 over two ranks, bf16 payload
        # numpy payloads are pickled by value: torch tensors would travel as shared-memory fds that die with this process
        q.put((rank, {k: (v[0].numpy().copy(), v[1].numpy().copy()) for k, v in res.items()}, scene['out_queries'].numpy().copy(),
               g.float().numpy().copy()))
    except Exception as e:      # fail fast instead of letting the parent wait for the queue timeout
        q.put((rank, repr(e), None, None))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('variant,V,K,plan,minmax_bs', [('v1', 5, 3, 'replicated', 1), ('v2', 4, 2, 'replicated', 1), ('v1', 'multi_ar', 4, 'replicated', 1),
                                                        ('v2', 'portrait', 3, 'replicated', 1), ('v2', 5, 3, 'broadcast', 1), ('v1', 'multi_ar', 4, 'broadcast', 1),
                                                        ('v2', 5, 3, 'replicated', None), ('v2', 'portrait', 3, 'broadcast', 2), ('v1', 6, 4, 'broadcast_whole', 1)])
def test_two_rank_gloo_equals_single(variant, V, K, plan, minmax_bs):
    """both multi-GPU plans over a world_size-2 gloo group == the unsharded scene, bit for bit ('broadcast': rank 0 builds the memory and
    broadcasts the banks - per memory update, behind each append, or ('broadcast_whole') in one piece after the build -, rank 1 owns every non-keyframe view).  minmax_bs != 1: LoftUp's MinMaxScaler scope spans views of BOTH ranks - the per-view
    (min, max) tables travel with the first all-gather and every rank pools the same table (VERDICT r4 missing 3)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, variant, V, K, q, plan, minmax_bs)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    assert all(not isinstance(g[1], str) for g in got), got
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if V in ('multi_ar', 'portrait'):
        model = tiny.build(tiny.OracleNS, variant)
        shapes, imgs = _scene_kind(V)
        with torch.no_grad():
            ref, ref_scene = run_scene(OracleBackend(model), lambda i: imgs[i], len(shapes), None, None, K, tiny.NAMES, shapes=shapes, minmax_bs=minmax_bs)
        V = len(shapes)
    else:
        ref, ref_scene = _scene(variant, V, K, minmax_bs=minmax_bs)
    merged = {}
    for rank, res, outq, g in got:
        assert torch.equal(torch.from_numpy(outq), ref_scene['out_queries'])                  # identical frozen queries on every rank
        assert torch.equal(torch.from_numpy(g), torch.tensor([[0., 1.], [10., 11.], [2., 3.], [12., 13.], [4., 5.]]))
        merged.update({k: (torch.from_numpy(a), torch.from_numpy(b)) for k, (a, b) in res.items()})
    assert sorted(merged) == list(range(V))
    for i in range(V):
        assert torch.equal(merged[i][0], ref[i][0]) and torch.equal(merged[i][1], ref[i][1])


def _rerun_worker(rank, world, port, q):
    """a streamed-bank receiver over two scenes: the bank object must survive set_images() (captured graphs hold its addresses)"""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from panst3r_amd.scene import SceneRunner
        torch.set_num_threads(2)
        V, K = 5, 3
        model = tiny.build(tiny.OracleNS, 'v1')
        first, second = tiny.images(V, H, W), [tiny.synth_image(100 + i, H, W, 7) for i in range(V)]
        _, order, owner = assign_views(V, K, world, plan='broadcast')
        mine = lambda imgs: {order[i]: imgs[order[i]] for i in range(V) if owner[i] == rank}
        with torch.no_grad():
            rn = SceneRunner(OracleBackend(model), mine(first), V, H, W, K, tiny.NAMES, rank, world, None, plan='broadcast', stream_bank=True)
            assert rn.stream_bank
            rn.run()
            bank_id = id(rn.bank)
            rn.set_images(mine(second))
            res, scene = rn.run()
        q.put((rank, {k: (v[0].numpy().copy(), v[1].numpy().copy()) for k, v in res.items()}, scene['out_queries'].numpy().copy(), id(rn.bank) == bank_id))
    except Exception as e:
        q.put((rank, repr(e), None, None))
        raise
    finally:
        dist.destroy_process_group()


def test_streamed_bank_receiver_keeps_its_bank_across_scenes():
    """with the bank streamed per memory update, the rank that RECEIVES it re-posted its receives into a freshly allocated bank on every
    run() - while a captured stage renders from the bank it saw at capture time.  The bank is now allocated once per runner: after set_images() with a
    different scene the receiver still holds the same bank object, and both ranks' outputs equal the unsharded run of the NEW scene bit for bit."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_rerun_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=600) for _ in range(2)]
    assert all(not isinstance(g[1], str) for g in got), got
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    V, K = 5, 3
    torch.set_num_threads(2)          # as the workers: the CPU GEMMs' blocking (and with it the last bit) follows the thread count
    model = tiny.build(tiny.OracleNS, 'v1')
    second = [tiny.synth_image(100 + i, H, W, 7) for i in range(V)]
    with torch.no_grad():
        ref, ref_scene = run_scene(OracleBackend(model), lambda i: second[i], V, H, W, K, tiny.NAMES)
    merged = {}
    for rank, res, outq, same_bank in got:
        if rank != 0:        # (rank 0 BUILDS its bank inside a stage - eagerly a new one per run, under capture part of the graph)
            assert same_bank, 'receiving rank %d allocated a new bank for the second scene' % rank
        assert torch.equal(torch.from_numpy(outq), ref_scene['out_queries'])
        merged.update({k: (torch.from_numpy(a), torch.from_numpy(b)) for k, (a, b) in res.items()})
    assert sorted(merged) == list(range(V))
    for i in range(V):
        assert torch.equal(merged[i][0], ref[i][0]) and torch.equal(merged[i][1], ref[i][1])


def test_bank_streaming_is_opt_in():
    """The per-update (asynchronous) bank transfer runs the transport's queue beside compute kernels - the co-running-queues situation of DESIGN.md
    section 4; until an N > 1 RCCL run has shown it bit-identical it is opt-in: the default sends the bank in one event-ordered broadcast."""
    import inspect
    import panst3r_amd.scene as S
    from panst3r_amd.panst3r import PanSt3R
    assert inspect.signature(S.SceneRunner.__init__).parameters['stream_bank'].default is False
    assert inspect.signature(S.run_scene).parameters['stream_bank'].default is False
    assert inspect.signature(PanSt3R.scene_runner).parameters['stream_bank'].default is False


def test_single_view_scene_is_refused_clearly():
    """V = 1: the memory build needs a pair (reference quirk 4: get_must3r_mem_batches(n < 2) is broken, the demo duplicates
    the image, tools/demo_panst3r.py:111-112) -> explicit ValueError instead of a failure deep inside; the duplicate works."""
    model = tiny.build(tiny.OracleNS, 'v1')
    img = tiny.images(1, H, W)[0]
    with pytest.raises(ValueError, match='at least 2 views'):
        run_scene(OracleBackend(model), lambda i: img, 1, H, W, None, tiny.NAMES)
    with torch.no_grad():
        res, _ = run_scene(OracleBackend(model), lambda i: img, 2, H, W, None, tiny.NAMES)
    assert torch.equal(res[0][1], res[1][1]) or rel_l2(res[0][1], res[1][1]) < 1.0      # both copies are rendered
    assert sorted(res) == [0, 1]


def test_eight_rank_bookkeeping_of_the_bench_scene(monkeypatch):
    """The driver's scaling run: 50 views, 16 keyframes over 8 ranks.  Every rank's SceneRunner is built in this one process
    (no process group) and the only collective, _all_gather_rows, is replaced by a fake that hands out what the other ranks
    would send: ownership, per-rank keyframe counts and the reorder into keyframe-schedule order are checked for all 8 ranks."""
    import panst3r_amd.scene as S
    V, K, world, Hh, Ww, p = 50, 16, 8, 384, 512, 16
    T = (Hh // p) * (Ww // p)

    class Dummy:
        patch_size, mask_dim = p, 8
        def fpn_grid(self, h, w):
            return (h, w), False

    keyframes, order, owner = assign_views(V, K, world)
    runners = []
    for r in range(world):
        mine = {order[i]: torch.zeros(3, 8, 8) for i in range(V) if owner[i] == r}
        runners.append(S.SceneRunner(Dummy(), mine, V, Hh, Ww, K, tiny.NAMES, rank=r, world=world))
    owned = sorted(v for rn in runners for v in (rn.order[i] for i in rn.mine))
    assert owned == list(range(V))                                           # every view exactly once
    assert [rn.n_local for rn in runners] == [7, 7, 6, 6, 6, 6, 6, 6] and sum(rn.k_local for rn in runners) == K
    assert all(rn.k_local == 2 for rn in runners) and all(rn.kf_T == [T] * K for rn in runners)
    # rank r's local keyframe rows carry the id of the keyframe (schedule position) they belong to
    sends = []
    for r, rn in enumerate(runners):
        ids = [i for i in rn.mine if i < K]                                  # positions in `order` == keyframe schedule positions
        assert ids == list(range(r, K, world))
        sends.append(torch.cat([torch.full((T, 2), float(k)) for k in ids]))
    monkeypatch.setattr(S, '_all_gather_rows', lambda t, counts, w, g: [s[:c] for s, c in zip(sends, counts)])
    monkeypatch.setattr(S.dist, 'is_initialized', lambda: True)
    for r in range(world):
        out = gather_keyframe_rows(sends[r], K, [T] * K, r, world, None)
        assert out.shape == (K * T, 2)
        assert torch.equal(out[::T, 0], torch.arange(K, dtype=torch.float32))    # schedule order 0..K-1 on every rank


def test_two_stream_stage2_is_opt_in():
    """The memory build beside the bulk branch on a second stream is not reproducible on MI355X: every
    runner executes the two branches back to back unless `overlap=True` is passed explicitly."""
    import panst3r_amd.scene as S

    class Dummy:
        patch_size, mask_dim = 16, 8
        def fpn_grid(self, h, w):
            return (h, w), False

    imgs = {i: torch.zeros(3, 8, 8) for i in range(4)}
    assert S.OVERLAP_DEFAULT is False
    assert S.SceneRunner(Dummy(), imgs, 4, 64, 64, 2, tiny.NAMES).serial is True
    assert S.SceneRunner(Dummy(), imgs, 4, 64, 64, 2, tiny.NAMES, overlap=True).serial is False


def test_tower_pass_shares_and_masked_layer_rule(monkeypatch):
    """host logic of scenes longer than one tower pass (PanSt3R.paired_shares) and of the masked stage 2 (scene.mask_layers): every view in exactly one
    pass, no pass above the pass size, near-equal shares; the layer count beside the build grows with the build (K) and shrinks with the views of the pass"""
    import panst3r_amd.panst3r as P
    import panst3r_amd.scene as S
    for Ve, Vd in [(34, 50), (168, 200), (1, 200), (64, 64), (65, 64), (0, 3), (200, 1), (129, 130)]:
        sh = P.PanSt3R.paired_shares(Ve, Vd)
        assert len(sh) == max(1, -(-max(Ve, Vd) // P.ENC_CHUNK))
        for tower, V in ((0, Ve), (1, Vd)):
            spans = [s[tower] for s in sh]
            assert spans[0][0] == 0 and spans[-1][1] == V and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) <= P.ENC_CHUNK and max(sizes) - min(sizes) <= 1
    monkeypatch.setattr(S, 'MASK_LAYERS', 0)
    assert S.mask_layers(16, 84, 24) == 11                    # the bench scene: the measured optimum (profiles/r5_overlap_bench.txt)
    assert S.mask_layers(32, 92, 24) == 23                    # C5: first of four passes of 42 + 50 views beside the 54 ms build
    assert S.mask_layers(2, 200, 24) == 1 and S.mask_layers(32, 4, 24) == 24
    assert [S.mask_layers(K, 84, 24) for K in (4, 8, 16, 32)] == sorted(S.mask_layers(K, 84, 24) for K in (4, 8, 16, 32))
    monkeypatch.setattr(S, 'MASK_LAYERS', 7)
    assert S.mask_layers(16, 84, 24) == 7 and S.mask_layers(16, 84, 5) == 5


def test_to_outdevice_off_gpu_is_a_plain_move():
    import panst3r_amd.scene as S
    a = torch.arange(12.).reshape(3, 4)
    out = S.to_outdevice([a[0], a[1:]], 'cpu')
    assert torch.equal(out[0], a[0]) and torch.equal(out[1], a[1:])
