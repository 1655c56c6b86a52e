"""One PanSt3R scene, view-sharded over the ranks of a torch.distributed group (option 1).

Plan (one process per GPU; RCCL over xGMI when the backend is "nccl", gloo in the CPU tests):
  A. each rank encodes ITS views (CroCo encoder + DINOv2), keyframes dealt round-robin first     -- no collective
  B. all_gather of the keyframes' encoder tokens ([K,T,1024] bf16 = 1.5 MiB x K); every rank then replays the
     sequential memory build redundantly -> identical memory bank everywhere (bit-exact, no broadcast of 13.5 MiB x K)
  C. each rank renders + upscales its views against the memory                                   -- no collective
  D. all_gather of the keyframes' FPN tokens and 2x2-centre mask features ([K,T,768+C] bf16); every rank runs the
     (tiny) panoptic query decoder redundantly -> identical frozen queries everywhere
  E. each rank computes the query x pixel masks of its views                                       -- no collective
The math per view is independent of the sharding, so results equal the 1-GPU run of the same build.

Second plan, `plan='broadcast'` (SURVEY 8(e) option 2): the build is a chain of ~3 300 tiny kernels (25.7 ms at K = 16) that every rank of the
replicated plan repeats.  Here ONLY rank 0 builds: it owns no view but its keyframes, builds right after gather B and broadcasts the projected
K / V^T banks (27 MiB x K), while ranks 1.. spend that time on their (larger) share of the encoder / DINOv2 work; then everyone renders.  Same
results (the bank is built once, by the same kernels, and copied bit for bit).

`run_scene` is written against a small stage backend so that the same orchestration is exercised on the GPU
(HipBackend: the HIP kernels) and in the world_size-2 gloo tests on CPU (tests/ provide an oracle-driven backend).
"""
import os

import torch
import torch.distributed as dist

from .schedule import select_keyframes, view_order


PLANS = ('replicated', 'broadcast')


def resolve_plan(plan, world, K=None):
    """'auto' -> the plan the one-GPU projection favours (tools/shard_estimate.py, profiles/r3_shard_estimate.txt: 50 views / 16 keyframes, critical path
    replicated vs broadcast: 2 ranks 104.7 vs 123.4 ms, 4 ranks 71.3 vs 57.5 ms, 8 ranks 58.1 vs 44.5 ms with the 453 MB of banks at an ASSUMED 100 GB/s):
    'broadcast' from 4 ranks on, 'replicated' below - and 'replicated' whenever there are fewer keyframes than ranks (`K` given), which the
    broadcast plan's deal cannot serve."""
    if plan in (None, 'auto'):
        return 'broadcast' if (world >= 4 and (K is None or K >= world)) else 'replicated'
    if plan not in PLANS:
        raise ValueError("plan must be 'auto' or one of %s (got %r)" % (PLANS, plan))
    return plan


def assign_views(V, K, world, keyframes=None, plan='replicated'):
    """keyframes (in schedule order) dealt round-robin to ranks, then the remaining views continue the deal.
    Returns (keyframes, order, owner) with owner[i] = rank of order[i].  `keyframes`: an explicit list of distinct view ids in
    memory-build order (e.g. schedule.keyframes_from_similarity, the reference's retrieval mode) instead of the linspace schedule.
    plan='broadcast': rank 0 (the only rank that builds the memory) gets no view beyond its keyframes; the other views are dealt over
    ranks 1 .. world-1."""
    if plan not in PLANS:
        raise ValueError('plan must be one of %s (got %r)' % (PLANS, plan))
    if keyframes is None:
        keyframes = select_keyframes(V, K)
    else:
        keyframes = [int(k) for k in keyframes]
        if len(set(keyframes)) != len(keyframes) or not all(0 <= k < V for k in keyframes) or len(keyframes) < 2:
            raise ValueError('keyframes must be >= 2 distinct view ids in [0, %d): %s' % (V, keyframes))
    order, _ = view_order(V, keyframes)
    Kn = len(keyframes)
    if plan == 'broadcast' and world > 1:
        owner = [i % world for i in range(Kn)] + [1 + (i % (world - 1)) for i in range(V - Kn)]
    else:
        owner = [i % world for i in range(V)]
    return keyframes, order, owner


def _all_gather_rows(t, counts, world, group):
    """all_gather of row blocks with uneven row counts (padded to the maximum); returns the list of per-rank blocks."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return [t]
    mx = max(counts)
    pad = torch.zeros(mx, t.shape[1], dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    raw = pad.view(torch.uint8)          # bit-preserving byte view: every backend (RCCL, gloo) moves uint8
    outs = [torch.empty_like(raw) for _ in range(world)]
    dist.all_gather(outs, raw.contiguous(), group=group)
    return [o.view(t.dtype)[:c] for o, c in zip(outs, counts)]


def gather_keyframe_rows(local_rows, K, T, rank, world, group):
    """local_rows: [sum of this rank's keyframe token counts, C] (keyframes in deal order) -> all K keyframes' rows in
    keyframe-schedule order.  T: tokens per keyframe, an int or a per-keyframe list (multi-aspect-ratio scenes)."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local_rows                    # one rank owns every keyframe, already in schedule order
    Ts = [T] * K if isinstance(T, int) else list(T)
    counts = [sum(Ts[kf] for kf in range(r, K, world)) for r in range(world)]
    blocks = _all_gather_rows(local_rows, counts, world, group)
    offs = [0]
    for t in Ts:
        offs.append(offs[-1] + t)
    out = torch.empty(offs[-1], local_rows.shape[1], dtype=local_rows.dtype, device=local_rows.device)
    for r, blk in enumerate(blocks):
        o = 0
        for kf in range(r, K, world):
            out[offs[kf]:offs[kf + 1]] = blk[o:o + Ts[kf]]
            o += Ts[kf]
    return out


def gather_view_rows(local_rows, owner, rank, world, group):
    """local_rows [n_local, ...] (this rank's views in local order = ascending position in `order`) -> the rows of ALL views by position in `order`"""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local_rows
    shape = local_rows.shape[1:]
    flat = local_rows.reshape(local_rows.shape[0], -1)
    counts = [sum(1 for o in owner if o == r) for r in range(world)]
    blocks = _all_gather_rows(flat, counts, world, group)
    out = torch.empty(len(owner), flat.shape[1], dtype=flat.dtype, device=flat.device)
    for r, blk in enumerate(blocks):
        pos = [i for i, o in enumerate(owner) if o == r]
        out[torch.tensor(pos, dtype=torch.long, device=out.device)] = blk
    return out.reshape(len(owner), *shape)


def broadcast_tensors(tensors, src, world, group):
    """broadcast a list of equally-shaped-on-every-rank tensors from `src` (byte views: every backend moves uint8)."""
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return
    for t in tensors:
        dist.broadcast(t.view(torch.uint8) if t.is_contiguous() else t, src, group=group)


# Run the two independent branches of stage 2 (sequential memory build || bulk encoder + DINOv2) on two streams.  OFF by default:
# with two HIP queues active, the output of the first DINOv2 kernel intermittently lost 64-byte half-lines (seen as DINOv2 tokens of
# whole views deviating), in captured graphs as well as eager launches, depending on the scene shape, and triggered just as well by a
# rocBLAS GEMM loop on private buffers in place of our memory build.  After a rewrite of that kernel no deviation has been seen, but
# probe variants show the rewrite is not a causal fix (the same instructions fail in one library and pass in another:
# tests/diag/dino_taps.py); the mechanism is not understood, so results must not depend on it.
# One stream is reproducible in every trial.  `overlap=True` (bench.py --overlap) re-enables the two streams.
OVERLAP_DEFAULT = False
# Round 5: the effect needs waves of BOTH queues on the SAME CU.  With the two queues on disjoint CU sets (hipExtStreamCreateWithCUMask) the stand-alone
# reproducer deviates 0 of 25 times on three boxes where plain (and full-mask) stream pairs deviate 25 of 25 (tests/diag/cu_mask_two_queue.py,
# profiles/r5_cu_mask_two_queue_box{1,2,3}.txt), a captured graph runs on the CUs of the stream it is launched on and two graphs on two masked streams run
# side by side (tests/diag/cu_mask_graph.py).  `overlap='masked'`: the sequential memory build on MASK_BUILD_CUS CUs beside the first MASK_LAYERS layers of
# the two ViT-L towers on the rest; everything behind the join on all CUs again.  The kernels keep the grids they have on the whole device (a masked queue maps a persistent
# kernel's 256 workgroups onto its CUs, two per CU); which kernel variant runs never changes a bit, so the masked scene equals the serial one bit for bit
# (tests/test_hip_fullsize.py).  Telling the kernels the CU count of their stream (pst_tune PST_TUNE_CUS -> grids of 112 / 144 workgroups; PST_CU_BUDGET=1 here)
# was measured and is OFF: the masked scene got slower (171 vs 154 ms, serial 160) and 2 of 36 replays deviated from the serial scene in the masks of the last
# view that is not a keyframe - 0 of 130+ without it (profiles/r5_overlap_soak_budget.txt, r5_overlap_soak_nobudget.txt).  Not understood; not used.
# Measured (tools/overlap_bench.py, profiles/r5_overlap_bench.txt; 50 views / 16 keyframes, healthy streams): the optimum is flat between 96 and 128 CUs for the
# build with 10 - 12 of the 24 tower layers beside it: 148.5 - 151.4 ms against 158.3 - 160.5 serial (+5 ... 7 %); 64 or 160 CUs for the build: no gain.  Some masked
# queues come up in a slow state on this platform (every kernel 3 - 6 x slower; HipBackend.masked_streams calibrates and re-creates them), so `pick_overlap`
# below still keeps the masked form only where it is measured faster on the box at hand.
MASK_BUILD_CUS = int(os.environ.get('PST_MASK_BUILD_CUS', '112'))
MASK_LAYERS = int(os.environ.get('PST_MASK_LAYERS', '0'))      # 0 = from the balance rule below (11 at 50 views / 16 keyframes); tools/overlap_bench.py sets it


def mask_layers(K, views_in_pass, n_layers):
    """how many layers of the first lock-step tower pass run beside the memory build: as many as take the build's time.  Measured at 384 x 512 (768 tokens per
    view): the build takes 23.2 ms at K = 16 and 53.7 ms at K = 32 (profiles/r5_build_bench.txt; 1.22 K + 0.0143 K^2), x 1.15 on its masked stream; one tower
    layer of one view takes 0.0286 ms on the other 144 CUs (11 layers of 84 views beside the K = 16 build).  Both sides scale with the token count alike.
    A mis-estimate costs idle time on one of the two streams, never a bit; `--overlap auto` keeps the masked form only where it is measured faster."""
    if MASK_LAYERS > 0:
        return min(MASK_LAYERS, n_layers)
    build_ms = 1.15 * (1.222 * K + 0.01426 * K * K)
    return max(1, min(n_layers, int(round(build_ms / (0.0286 * max(views_in_pass, 1))))))
DIAG_CONCURRENT = None     # diagnostics only (tests/diag/dino_taps.py): a callable run on the main stream beside the side branch


def output_digest(res, scene):
    """one int64 over every bit of a scene's outputs (pointmaps, mask logits, queries, class logits): equal digests = equal bits, up to collisions of a sum of
    the 32-bit words; ~1 ms at 50 views.  What bench.py records so that the timed form of the scene is a CHECKED one."""
    tot = None
    for t in [scene['out_queries'], scene['pred_logits']] + [x for i in sorted(res) for x in res[i]]:
        d = t.contiguous().view(torch.int32).to(torch.int64).sum()
        tot = d if tot is None else tot * 1000003 + d           # (order-sensitive across tensors; wraps in int64)
    return int(tot)


def pick_overlap(make_runner, steps=3):
    """overlap='auto': build the serial and the masked runner of a scene (make_runner(overlap) -> SceneRunner with captured graphs), time `steps` replays of
    each and keep the faster one; the other is released.  Returns (runner, {'chosen', 'serial_ms', 'masked_ms', 'serial_digest', 'masked_identical'}).
    Both produce the same bits - and that is CHECKED here on every replay of the trial: a masked form whose outputs deviate from the serial scene's is
    never chosen ('masked_identical': false in the record)."""
    import time
    res = {}
    runners = {}
    for mode in (False, 'masked'):
        r = make_runner(mode)
        if mode == 'masked' and not r.masked:          # the scene cannot take the masked form (plan, shapes): serial
            r.release()
            continue
        r.run(copy=False)
        digests = {output_digest(*r.run(copy=False))}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.run(copy=False)
        torch.cuda.synchronize()
        res['masked_ms' if mode else 'serial_ms'] = 1e3 * (time.perf_counter() - t0) / steps
        digests.add(output_digest(*r.results(copy=False)))
        if mode:
            res['masked_identical'] = digests == {res['serial_digest']}
        else:
            res['serial_digest'] = digests.pop() if len(digests) == 1 else None       # (None: the serial scene itself is not reproducible - never seen)
        runners[mode] = r
    best = 'masked' if (res.get('masked_identical') and res['masked_ms'] < 0.99 * res['serial_ms']) else False
    for mode, r in runners.items():
        if mode != best:
            r.release()
    res['chosen'] = 'masked' if best else 'serial'
    return runners[best], res


def adt_of(rows):
    return next(r.dtype for r in rows if r is not None)


def to_outdevice(tensors, outdevice):
    """Copies of `tensors` (device tensors, typically per-view VIEWS of a few large blocks) on `outdevice`, same shapes.  Towards the host the copies go
    through PINNED memory: one staging allocation and one DMA per distinct device storage that the tensors cover (almost) completely - the per-view mask
    and pointmap tensors of a scene are views of one block per shape group -, everything asynchronous on the current stream, ONE synchronisation at the end.
    (The reference's `.to(outdevice)` per tensor lands in pageable memory: ~7 GB/s and a synchronisation per tensor; 2.2 GB of outputs at 50 views.)
    The returned host tensors are views of the pinned blocks (torch's caching host allocator recycles them once every view is gone)."""
    dev = torch.device(outdevice)
    if dev.type != 'cpu' or not tensors or not tensors[0].is_cuda:
        return [t.to(dev) for t in tensors]
    try:
        return _to_host_pinned(tensors)
    except RuntimeError as e:                  # the host cannot pin that much memory (hipHostMalloc failed): pageable copies, as the reference does
        if 'out of memory' not in str(e).lower() and 'hipHostMalloc' not in str(e) and 'pin' not in str(e).lower():
            raise
        return [t.to(dev) for t in tensors]


def _to_host_pinned(tensors):
    by_store = {}
    for k, t in enumerate(tensors):
        by_store.setdefault((t.untyped_storage().data_ptr(), t.dtype), []).append(k)
    out = [None] * len(tensors)
    for (_, dt), ks in by_store.items():
        ts = [tensors[k] for k in ks]
        lo = min(t.storage_offset() for t in ts)
        hi = max(t.storage_offset() + (sum((n - 1) * st for n, st in zip(t.shape, t.stride())) + 1 if t.numel() else 0) for t in ts)
        covered = sum(t.numel() for t in ts)
        if len(ts) > 1 and covered >= 0.9 * (hi - lo):            # views that tile one span of the storage: one DMA of the span
            span = torch.empty(0, dtype=dt, device=ts[0].device).set_(ts[0].untyped_storage(), lo, (hi - lo,), (1,))
            host = torch.empty(hi - lo, dtype=dt, pin_memory=True)
            host.copy_(span, non_blocking=True)
            for k, t in zip(ks, ts):
                out[k] = host.as_strided(t.shape, t.stride(), t.storage_offset() - lo)
        else:
            for k, t in zip(ks, ts):
                h = torch.empty(t.shape, dtype=dt, pin_memory=True)
                h.copy_(t, non_blocking=True)
                out[k] = h
    torch.cuda.current_stream().synchronize()
    return out


class _Group:
    """The views of one image shape owned by this rank (keyframes first)."""
    __slots__ = ('H', 'W', 'h', 'w', 'T', 'idx', 'k', 'imgs', 'cat', 'pointmaps', 'fpn', 'mf', 'guid', 'mm', 'enc', 'pos')


class SceneRunner:
    """One scene as three stages separated by the two all-gathers:
         stage1  CroCo encoder on own keyframes                     -> enc_send
         gather  keyframe encoder tokens                            -> enc_kf
         stage2  encoder of the other views + DINOv2; memory build (replayed on every rank); render; upscale
         gather  keyframe FPN tokens + attention-mask features      -> both_kf
         stage3  query decoding (replayed on every rank) + query x pixel masks of own views
    With `use_graphs=True` (GPU only) each stage is captured once into a HIP graph and replayed: a scene is ~4 700
    kernel launches, most of them 5-20 us kernels of the sequential memory build, so eager launching is host-bound.
    The collectives stay eager between the graph replays.  Shapes, keyframe schedule and class list are static.
    Views may have different shapes (landscape or portrait, native orientation): they are batched per shape group
    (multi-aspect-ratio scenes); `backend.fpn_grid(h, w)` gives the key grid / orientation flag the query decoder sees."""

    def __init__(self, backend, images, V, H, W, K, classes, rank=0, world=1, group=None, use_graphs=False, shapes=None, overlap=None, keyframes=None,
                 amp=None, plan='replicated', minmax_bs=1, pan_amp=None, mm_override=None, pan_scope='reference', stream_bank=False):
        self.b, self.V, self.classes = backend, V, classes
        # reference AMP placement (panst3r.py:174-175,204-245,268): `amp` names the format of the encoder, the memory build and the keyframes' render +
        # DINOv2; `pan_amp` (None = the same format) that of the panoptic decoder AND of the render + DINOv2 of the views that are not keyframes.
        # With two formats in play (`mixed`) the feature concat is kept in the panoptic format and the encoder tokens additionally in the scene's.
        # `pan_scope` says how far the second format reaches: 'reference' = as just described (the reference's autocast boundary); 'decoder' = the panoptic
        # decoder ONLY (InputMixer, upscaler, query decoder, mask head) - every view's encoder, DINOv2 and render stay in the scene's format.  That is the
        # default placement of amp='bf16' (PanSt3R.forward_inference_multi_ar: bf16 where the reference autocasts, f16 - 3 more mantissa bits at the same MFMA
        # rate - where it computes in fp32 and every operand sits behind a normalisation).
        self.pan_amp = amp if pan_amp is None else pan_amp
        self.mixed = pan_amp is not None and not backend.same_format(amp, pan_amp)
        if pan_scope not in ('reference', 'decoder'):
            raise ValueError("pan_scope must be 'reference' or 'decoder' (got %r)" % (pan_scope,))
        self.ref_split = self.mixed and pan_scope == 'reference'        # the other views' render + DINOv2 run in the panoptic format too
        # LoftUp's MinMaxScaler scope (loftup.py:14-19 pools min / max over the chunk of views it is handed; the reference chunks by max_bs):
        # 1 = per view (the demo's max_bs=1, tools/demo_panst3r.py:201 - the default here and what bench.py times); k = same-shape keyframes /
        # same-shape other views in chunks of k, None = all of them together (the reference with max_bs=None: stack_views + batched_map,
        # panst3r.py:212-216,244,257-270).  A scope may span ranks: every rank takes the per-view (min, max) table of ITS views in stage 1, the
        # tables travel with the first all-gather (6 floats per view) and every rank pools the whole table over the same scope ids (exact:
        # min / max do not depend on the order).  `mm_override`: {view id: fp32 [3, 2] (min, max)} = tables pooled by the caller over a
        # scope the scene does not see (PanSt3R.forward pools over ALL scenes of a batch, panst3r.py:294 / panoptic_decoder.py:56-62).
        self.minmax_bs = minmax_bs
        self.mm_override = mm_override
        self.amp = amp                # False | 'bf16' | 'fp16' (reference utils.py:206-215): 16-bit format of this runner, fixed for its lifetime
        self._refs = None             # packed weights / tables the captured graphs point into (kept alive with the runner)
        self.rank, self.world, self.group = rank, world, group
        if V < 2:
            # the memory build starts from a PAIR of views (schedule [2,1,1,...], panst3r.py:35-39,65-70; the reference's helper
            # yields a negative batch for n < 2); the demo duplicates a lone image instead (tools/demo_panst3r.py:111-112)
            raise ValueError('a scene needs at least 2 views (got %d): duplicate a single image as the reference demo does' % V)
        self.K = K = len(keyframes) if keyframes is not None else (V if (K is None or K > V) else max(int(K), 2))
        if V < world:
            raise ValueError('need at least one view per rank (V=%d, world=%d)' % (V, world))
        self.shapes = [tuple(sh) for sh in shapes] if shapes is not None else [(H, W)] * V      # per view id
        self.plan = plan = resolve_plan(plan, world, K)
        if plan == 'broadcast' and world > 1 and K < world:
            raise ValueError("plan='broadcast' deals the keyframes over all ranks and the other views over ranks 1..: it needs K >= world (K=%d, world=%d)" % (K, world))
        self.keyframes, self.order, owner = assign_views(V, K, world, keyframes, plan)
        empty = [r for r in range(world) if r not in owner]
        if empty:           # every rank computes this from the same arguments: all of them raise, none is left waiting in a collective
            raise ValueError('plan %r leaves rank(s) %s without a view (V=%d, K=%d, world=%d)' % (plan, empty, V, K, world))
        self.builder = plan == 'replicated' or rank == 0        # this rank runs the sequential memory build
        # the broadcast plan's split stage 2 also runs on a 1-rank process group (PST_FORCE_DIST=1: the collectives execute on RCCL at world = 1)
        self.split = plan == 'broadcast' and (world > 1 or (dist.is_available() and dist.is_initialized()))
        # broadcast plan, stream_bank=True: the bank travels PER MEMORY UPDATE (the entries of update u go out while update u + 1 computes: K - 1 async broadcasts of
        # 13.5 MiB x keyframes each instead of one of 27 MiB x K after the whole build - off the critical path).  OPT-IN since round 6: the async
        # broadcasts run on the transport's own queue BESIDE this rank's compute kernels, the co-running-queues situation of DESIGN.md section 4 (two-queue effect),
        # and no N > 1 RCCL run has shown it bit-identical yet.  The default (False) is EVENT-ORDERED: one synchronous broadcast behind the whole build - the
        # collective's queue waits for the compute stream and the compute stream for the collective, so the two never share a CU.
        self.stream_bank = bool(stream_bank) and self.split and hasattr(backend, 'bank_update_payload')
        self._works, self._staged = [], []
        self.mine = [i for i in range(V) if owner[i] == rank]       # positions in `order`; keyframe positions first
        self.n_local = len(self.mine)
        self.k_local = sum(1 for i in self.mine if i < K)
        p = backend.patch_size
        self.kf_grids = [(self.shapes[v][0] // p, self.shapes[v][1] // p) for v in self.keyframes]     # schedule order
        self.kf_T = [a * c for a, c in self.kf_grids]
        kf_fpn = [backend.fpn_grid(a, c) for a, c in self.kf_grids]        # key grid + portrait flag per keyframe
        self.kf_fpn_grids, self.kf_portrait = [g for g, _ in kf_fpn], [pt for _, pt in kf_fpn]
        # shape groups of the local views (local index j = position in `mine`)
        self.groups = []
        by_shape = {}
        for j, i in enumerate(self.mine):
            sh = self.shapes[self.order[i]]
            if sh not in by_shape:
                g = _Group()
                g.H, g.W = sh
                g.h, g.w = sh[0] // p, sh[1] // p
                g.T = g.h * g.w
                g.idx, g.k = [], 0
                by_shape[sh] = g
                self.groups.append(g)
            by_shape[sh].idx.append(j)
            if i < K:
                by_shape[sh].k += 1
        for g in self.groups:
            g.imgs = torch.stack([images[self.order[self.mine[j]]] for j in g.idx]).float().contiguous()   # static input buffers
        self.where = {}                       # local index j -> (group, row within group)
        for g in self.groups:
            for r, j in enumerate(g.idx):
                self.where[j] = (g, r)
        for g in self.groups:                 # position in `order` of every row of the group
            g.pos = [self.mine[j] for j in g.idx]
        # scope id of every view, by position in `order` (identical on every rank): views with equal ids share one min-max scale.  The reference
        # stacks same-shape keyframes and same-shape other views separately (stack_views, panst3r.py:212-216,257-261), in chunks of max_bs.
        self.mm_scope = self.mm_all = self.mm_send = None
        self.owner = owner
        if getattr(backend, 'minmax_scaled', lambda: False)():
            if mm_override is not None:
                self.mm_all = backend.table_of([mm_override[self.order[i]] for i in range(V)], self.groups[0].imgs.device)
            elif minmax_bs != 1:
                ids, nxt = [0] * V, 0
                for lo, hi in ((0, K), (K, V)):
                    by_shape = {}
                    for i in range(lo, hi):
                        by_shape.setdefault(self.shapes[self.order[i]], []).append(i)
                    for members in by_shape.values():
                        bs = len(members) if minmax_bs is None else int(minmax_bs)
                        for c, i in enumerate(members):
                            ids[i] = nxt + c // bs
                        nxt = ids[members[-1]] + 1
                if len(set(ids)) < V:
                    self.mm_scope = backend.scope_ids(ids, self.groups[0].imgs.device)
        self.use_graphs = use_graphs
        ov = OVERLAP_DEFAULT if overlap is None else overlap
        # 'masked': the build and the first tower layers on disjoint CU sets (above); needs the lock-step tower pass for every shape group and a rank that builds
        self.masked = ov == 'masked' and not self.split and self.builder and not self.ref_split and hasattr(backend, 'masked_streams') and \
            all(backend.rest_pairable(g.imgs[g.k:], g.imgs) for g in self.groups) and backend.masked_streams(self.groups[0].imgs.device, MASK_BUILD_CUS) is not None
        self.serial = self.masked or not ov      # True: no plain two-stream stage 2 (the two branches back-to-back, or the masked form)
        self.graphs = None
        self.coll_events = None       # bench.py: [] -> every eager collective is bracketed by HIP events (collective_ms)
        self.enc_kf = self.both_kf = None
        self.out = self.bank = self._rest = None

    def _kf_rows(self, per_group_rows):
        """concatenate this rank's keyframe rows (one [k_g*T_g, C] tensor per group) in deal order."""
        if len(self.groups) == 1:
            return per_group_rows[0].contiguous()
        parts = []
        for j in range(self.k_local):
            g, r = self.where[j]
            parts.append(per_group_rows[self.groups.index(g)][r * g.T:(r + 1) * g.T])
        return torch.cat(parts) if parts else per_group_rows[0][:0].contiguous()

    # ---- stages (every tensor they leave on `self` is read by a later stage)
    def stage1(self):
        """CroCo encoder on this rank's keyframes only: all the memory build needs."""
        b = self.b
        rows = []
        for g in self.groups:
            with b.precision(self.pan_amp):
                g.cat = b.alloc_cat(len(g.idx) * g.T, g.imgs.device)
            g.enc = b.alloc_enc(len(g.idx) * g.T, g.imgs.device) if self.mixed else None       # encoder tokens in the scene's format (mixed: cat is not)
            if g.k:
                b.encode_enc(g.imgs[:g.k], g.cat[:g.k * g.T], None if g.enc is None else g.enc[:g.k * g.T])
            rows.append(b.enc_rows(g.cat if g.enc is None else g.enc, g.k * g.T))
        self.enc_send = self._kf_rows(rows)
        if self.mm_scope is not None:          # per-view (min, max) of this rank's views, in local order (6 floats per view; pooled behind gather 1)
            with b.precision(self.pan_amp):
                tabs = [b.minmax_local(g.imgs) for g in self.groups]
            self.mm_send = tabs[0] if len(self.groups) == 1 else b.rows_in_order(tabs, [g.idx for g in self.groups], self.n_local)

    def _encode_rest(self):
        """Everything the build does not depend on: encoder of the non-keyframe views + DINOv2 of every view."""
        b = self.b
        for g in self.groups:
            if not self.ref_split and len(g.idx) > g.k and hasattr(b, 'encode_rest_paired'):
                # the encoder of the views that are not keyframes and DINOv2 of all views, layer by layer in lock-step (shared launches)
                b.encode_rest_paired(g.imgs[g.k:], g.cat[g.k * g.T:], g.imgs, g.cat, None if g.enc is None else g.enc[g.k * g.T:])
                continue
            if len(g.idx) > g.k:
                b.encode_enc(g.imgs[g.k:], g.cat[g.k * g.T:], None if g.enc is None else g.enc[g.k * g.T:])
            if not self.ref_split:
                b.encode_dino(g.imgs, g.cat)
            else:               # reference placement: DINOv2 of the keyframes under autocast (panst3r.py:229-230), of the other views outside it (:150 via :268)
                if g.k:
                    b.encode_dino(g.imgs[:g.k], g.cat[:g.k * g.T])
                if len(g.idx) > g.k:
                    with b.precision(self.pan_amp):
                        b.encode_dino(g.imgs[g.k:], g.cat[g.k * g.T:])
        self._guidance()

    def _guidance(self):
        b = self.b
        with b.precision(self.pan_amp):
            pooled = self.mm_all
            if self.mm_scope is not None:           # the whole scene's per-view table pooled over the scope ids (every rank: same table, same ids)
                pooled = b.minmax_pool(self.mm_all, self.mm_scope)
            for g in self.groups:                   # image-only part of the upscaler (LoftUp guidance convs, SURVEY 8(e) phase A): memory-independent,
                g.mm = None if pooled is None else b.table_rows(pooled, g.pos)         # so it belongs to this branch (with overlap=True it fills the
                g.guid = b.guidance(g.imgs, g.h, g.w, g.mm)                            # tail of the memory build)

    def gather1(self):
        kf = gather_keyframe_rows(self.enc_send, self.K, self.kf_T, rank=self.rank, world=self.world, group=self.group)
        if self.enc_kf is None or self.enc_kf is kf:
            self.enc_kf = kf
        else:
            self.enc_kf.copy_(kf)
        if self.mm_scope is not None:
            tab = gather_view_rows(self.mm_send, self.owner, self.rank, self.world, self.group)
            if self.mm_all is None or self.mm_all is tab:
                self.mm_all = tab
            else:
                self.mm_all.copy_(tab)

    def stage2(self):
        """replicated plan: everything between the two all-gathers as ONE stage (one captured graph)."""
        self.stage2a()
        self.stage2b()

    def stage2a(self):
        """Up to the point where the memory bank exists on the rank that builds it.
        The sequential memory build is a chain of thousands of tiny kernels that leaves most CUs idle; with `overlap` the
        independent bulk work (_encode_rest) runs concurrently on a second stream (a parallel branch of the captured graph).
        plan='broadcast': rank 0 builds and nothing else (its own encoder / DINOv2 work follows in stage2b, behind the broadcast); the other
        ranks do their bulk work here and allocate the bank they receive."""
        b = self.b
        dev = self.groups[0].imgs.device
        if self.split:
            if self.builder:
                self.bank = b.build_memory(self.enc_kf, self.K, self.kf_grids, self.ref_split)
            else:
                self._encode_rest()
                if not self.stream_bank:          # (streamed: allocated when the receives were posted, before this stage)
                    self.bank = b.bank_alloc(self.K, self.kf_grids, dev, self.ref_split)
            return
        # (measured +5 % frames/s at 50 views, but unsafe on this platform - see OVERLAP_DEFAULT - hence only when asked for)
        side = b.side_stream(dev) if not self.serial else None
        if side is None:
            self._encode_rest()
            bank = b.build_memory(self.enc_kf, self.K, self.kf_grids, self.ref_split)
        else:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._encode_rest()
            if DIAG_CONCURRENT is not None:        # diagnostics only (tests/diag/dino_taps.py): some other workload beside the side branch
                DIAG_CONCURRENT()
                main.wait_stream(side)
                bank = b.build_memory(self.enc_kf, self.K, self.kf_grids, self.ref_split)
            else:
                bank = b.build_memory(self.enc_kf, self.K, self.kf_grids, self.ref_split)
                main.wait_stream(side)
        self.bank = bank

    # ---- broadcast plan with a streamed bank: per-update stages / collectives (callables are cached: _segments is called more than once)
    def _cached(self, key, make):
        c = self.__dict__.setdefault('_seg_cache', {})
        if key not in c:
            c[key] = make()
        return c[key]

    def _build_step(self, u):
        def step():
            b = self.b
            if u == 0:
                self.bank = b.bank_new(self.K, self.kf_grids, self.groups[0].imgs.device, self.ref_split)
            b.build_step(self.bank, self.enc_kf, self.K, self.kf_grids, u)
        step.__name__ = 'build_step_%d' % u
        return self._cached(('step', u), lambda: step)

    def _bank_send(self, u):
        def send():
            if u == 0:
                self._works, self._staged = [], []
            payload = self.b.bank_update_payload(self.bank, self.K, self.kf_grids, u)        # contiguous staging copies of the update's entries
            self._staged.append(payload)                                                    # (alive until the transfer is done)
            for t in payload:
                self._works.append(dist.broadcast(t.view(torch.uint8) if t.is_contiguous() else t, 0, group=self.group, async_op=True))
            if u == len(self.b.update_spans(self.K, self.kf_grids)) - 1:
                for w in self._works:
                    w.wait()
                self._works, self._staged = [], []
                self.bank = self.b.bank_final(self.bank)
        send.__name__ = 'bank_send_%d' % u
        return self._cached(('send', u), lambda: send)

    def _bank_post_recvs(self):
        b = self.b
        # the bank is allocated ONCE per runner (release() drops it): this collective runs eagerly on every run(), while a captured stage2b renders from
        # the addresses it saw at capture time - a fresh bank per run would leave the graphs on a freed one
        if self.bank is None:
            self.bank = b.bank_alloc(self.K, self.kf_grids, self.groups[0].imgs.device, self.ref_split)
        self._works, self._staged = [], []
        for u in range(len(b.update_spans(self.K, self.kf_grids))):
            payload = b.bank_update_buffers(self.bank, self.K, self.kf_grids, u)
            self._staged.append(payload)
            for t in payload:
                self._works.append(dist.broadcast(t.view(torch.uint8) if t.is_contiguous() else t, 0, group=self.group, async_op=True))

    def _bank_finish(self):
        for w in self._works:
            w.wait()
        for u, payload in enumerate(self._staged):
            self.b.bank_update_store(self.bank, self.K, self.kf_grids, u, payload)
        self._works, self._staged = [], []

    def bank_exchange(self):
        """plan='broadcast': the projected K / V^T caches of all layers go from rank 0 to everybody (RCCL broadcast over xGMI; 27 MiB x K)."""
        if self.split:
            broadcast_tensors(self.b.bank_payload(self.bank), 0, self.world, self.group)

    def stage2b(self):
        b = self.b
        if self.split and self.builder:
            self._encode_rest()                    # rank 0's own (keyframe) views: DINOv2 + guidance, after the bank is on its way
        bank = self.bank
        rows = []
        for g in self.groups:
            n = len(g.idx)
            if not self.mixed:
                g.pointmaps = b.render(g.cat, n, g.h, g.w, bank)
            elif not self.ref_split:      # second format for the panoptic decoder only: every view rendered in the scene's format from its copy of the encoder tokens
                g.pointmaps = b.render(g.cat, n, g.h, g.w, bank, g.enc)
            else:               # reference placement: keyframes rendered in the scene's format (panst3r.py:221-227), the other views in the panoptic one (:268)
                kT = g.k * g.T
                pms = [b.render(g.cat[:kT], g.k, g.h, g.w, bank, g.enc[:kT])] if g.k else []
                if n > g.k:
                    with b.precision(self.pan_amp):
                        pms.append(b.render(g.cat[kT:], n - g.k, g.h, g.w, b.bank_f32(bank)))
                g.pointmaps = torch.cat(pms) if len(pms) > 1 else pms[0]
            with b.precision(self.pan_amp):
                g.fpn, g.mf = b.features(g.cat, g.imgs, n, g.h, g.w, g.guid, g.mm)
                g.guid = g.mm = None
                fm = b.attn_feats(g.mf, g.k, b.fpn_grid(g.h, g.w)[0])
            self.d = g.fpn.shape[1]
            rows.append(torch.cat([g.fpn[:g.k * g.T], fm], dim=1) if g.k else g.fpn.new_zeros(0, self.d + b.mask_dim))
        self.both_send = self._kf_rows(rows)

    def gather2(self):
        kf = gather_keyframe_rows(self.both_send, self.K, self.kf_T, rank=self.rank, world=self.world, group=self.group)
        if self.both_kf is None or self.both_kf is kf:
            self.both_kf = kf
        else:
            self.both_kf.copy_(kf)

    def stage3(self):
        b = self.b
        with b.precision(self.pan_amp):
            outq, head = b.decode(self.both_kf[:, :self.d].contiguous(), self.both_kf[:, self.d:].contiguous(), self.K, self.kf_fpn_grids,
                                  self.classes, self.kf_portrait)
            masks = [None] * self.n_local
            for g in self.groups:
                gm = b.masks_group(head, g.mf)          # [n, Q, Hm, Wm]: all views of the shape group in one launch where the backend can
                for r, j in enumerate(g.idx):
                    masks[j] = gm[r]
            self.out = (outq, b.logits(head), masks)

    def _segments(self):
        """[(stage, collective run eagerly behind it | None)]: the replicated plan has three stages, the broadcast plan splits stage 2 around the
        bank broadcast"""
        if self.split and self.stream_bank:
            U = len(self.b.update_spans(self.K, self.kf_grids))
            if self.builder:
                mid = [(self._build_step(u), self._bank_send(u)) for u in range(U)]
            else:         # the receives are posted BEFORE this rank's own stage 2a work (they complete behind it, on the transport's stream)
                mid = [(None, self._bank_post_recvs), (self.stage2a, self._bank_finish)]
            return [(self.stage1, self.gather1)] + mid + [(self.stage2b, self.gather2), (self.stage3, None)]
        if self.split:
            return [(self.stage1, self.gather1), (self.stage2a, self.bank_exchange), (self.stage2b, self.gather2), (self.stage3, None)]
        if self.masked:       # a PAIR of stages = two branches on two CU-masked streams, joined before the next segment
            return [(self.stage1, self.gather1), ((self.stage2_build, self.stage2_head), None), (self.stage2_tail, self.gather2), (self.stage3, None)]
        return [(self.stage1, self.gather1), (self.stage2, self.gather2), (self.stage3, None)]

    # ---- masked overlap: stage 2 as  [build || first tower layers]  ->  [remaining layers, guidance, render, upscale]
    def stage2_build(self):
        self.bank = self.b.build_memory(self.enc_kf, self.K, self.kf_grids, False)

    def stage2_head(self):
        b = self.b
        self._rest = [b.rest_begin(g.imgs[g.k:], g.imgs) for g in self.groups]
        # the same number of layers for every shape group's first pass, together as long as the build
        self._head_layers = mask_layers(self.K, sum(st.get('views', 0) for st in self._rest), max(st['n'] for st in self._rest))
        for st in self._rest:
            b.rest_layers(st, 0, min(self._head_layers, st['n']))

    def stage2_tail(self):
        b = self.b
        for g, st in zip(self.groups, self._rest):
            b.rest_layers(st, min(self._head_layers, st['n']), st['n'])
            b.rest_finish(st, g.cat[g.k * g.T:], g.cat, None if g.enc is None else g.enc[g.k * g.T:])
        self._rest = None
        self._guidance()
        self.stage2b()

    def _par(self, pair, graphs=None):
        """run the two branches of a masked pair: eagerly, or (graphs = the two captured graphs) by replaying one graph per masked stream"""
        b = self.b
        cur = torch.cuda.current_stream()
        sa, sb, ca, cb = b.masked_streams(self.groups[0].imgs.device, MASK_BUILD_CUS)
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        for i, (stage, s, cus) in enumerate(((pair[0], sa, ca), (pair[1], sb, cb))):
            with torch.cuda.stream(s):
                if graphs is not None:
                    graphs[i].replay()
                else:
                    b.cu_budget(cus)
                    try:
                        stage()
                    finally:
                        b.cu_budget(0)
        cur.wait_stream(sa)
        cur.wait_stream(sb)

    def _collective(self, coll):
        """run one eager collective; with `coll_events` set (bench.py) it is bracketed by HIP events on the launch stream"""
        if self.coll_events is None or not torch.cuda.is_available():
            return coll()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        coll()
        b.record()
        self.coll_events.append((coll.__name__, a, b))

    def collective_ms(self):
        """{collective name: [ms per recorded call]} of the calls recorded since `coll_events = []` (synchronises)"""
        torch.cuda.synchronize()
        out = {}
        for name, a, b in self.coll_events or []:
            out.setdefault(name, []).append(a.elapsed_time(b))
        return out

    def _eager(self):
        for stage, coll in self._segments():
            if isinstance(stage, tuple):
                self._par(stage)
            elif stage is not None:
                stage()
            if coll is not None:
                self._collective(coll)

    def release(self):
        """Drop everything the runner holds on the device (stacked inputs, feature / mask-feature buffers, gathered keyframe rows, captured
        graphs).  Outputs already handed out by results() stay valid: they are tensors of their own."""
        for g in self.groups:
            g.imgs = g.cat = g.pointmaps = g.fpn = g.mf = g.guid = g.mm = g.enc = None
        self.enc_kf = self.both_kf = self.enc_send = self.both_send = self.out = self.graphs = self._refs = self.bank = None
        self.mm_all = self.mm_send = self._rest = None
        self._sgraphs = self._sx = None

    def set_images(self, images):
        """Load a new scene of the SAME shapes / schedule into the static input buffers (the captured graphs read them in place).
        images: {view_id: [3,H,W] tensor} or a list indexed by view id; only this rank's views are read."""
        for g in self.groups:
            for r, j in enumerate(g.idx):
                vid = self.order[self.mine[j]]
                im = images[vid]
                if tuple(im.shape[-2:]) != (g.H, g.W):
                    raise ValueError('view %d: shape %s does not match the runner (%d, %d)' % (vid, tuple(im.shape[-2:]), g.H, g.W))
                g.imgs[r].copy_(im, non_blocking=True)

    def _capture(self):
        self._eager()                              # warm-up: packs weights, builds tables, fills allocator pools
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        self.graphs = []
        for stage, gather in self._segments():
            if isinstance(stage, tuple):
                # two branches that REPLAY CONCURRENTLY: each captured on its own masked stream with the CU budget of that stream, each with a memory pool
                # of its own (graphs that share a pool must not run at the same time: their temporaries alias)
                sa, sb, ca, cb = self.b.masked_streams(self.groups[0].imgs.device, MASK_BUILD_CUS)
                pair = []
                for st, s_, cus in ((stage[0], sa, ca), (stage[1], sb, cb)):
                    g = torch.cuda.CUDAGraph()
                    self.b.cu_budget(cus)
                    try:
                        with torch.cuda.graph(g, stream=s_, capture_error_mode='thread_local'):
                            st()
                    finally:
                        self.b.cu_budget(0)
                    pair.append(g)
                self.graphs.append(tuple(pair))
                self._par(stage, pair)
                continue
            if stage is None:                  # a collective-only segment (posting the bank receives)
                self.graphs.append(None)
                gather()
                continue
            g = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of torch.distributed may query events while we capture
            with torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):
                stage()
            self.graphs.append(g)
            g.replay()
            if gather is not None:
                gather()
        torch.cuda.synchronize()
        if hasattr(self.b, 'pack_refs'):
            self._refs = self.b.pack_refs()      # the graphs hold raw pointers into these packs: they must outlive a later load_state_dict

    @torch.no_grad()
    def run(self, outdevice=None, eager=False, serial=None, copy=True):
        """Execute the scene.  copy=True (default) returns outputs the caller owns; copy=False returns VIEWS into the runner's
        graph-pool buffers, valid only until the next run() (bench.py, which consumes nothing, uses that)."""
        if serial is not None:
            assert eager or not self.use_graphs or self.graphs is None, 'overlap mode is fixed once the graphs are captured'
            prev, self.serial = (self.serial, self.masked), serial
            if serial:
                self.masked = False          # one stream, all CUs (bench.py's instrumented step: event durations not inflated by a co-runner)
            try:
                return self.run(outdevice, eager, copy=copy)
            finally:
                self.serial, self.masked = prev
        with self.b.precision(self.amp):
            if self.use_graphs and not eager:
                if self.graphs is None:
                    self._capture()
                else:
                    for g, (stage, coll) in zip(self.graphs, self._segments()):
                        if isinstance(g, tuple):
                            self._par(stage, g)
                        elif g is not None:
                            g.replay()
                        if coll is not None:
                            self._collective(coll)
            else:
                self._eager()
        return self.results(outdevice, copy=copy)

    def streamable(self):
        """run_streamed applies: one rank, one precision placement for the render (not the reference's split placement)"""
        return self.world == 1 and not self.split and not self.ref_split and not self.masked and hasattr(self.b, 'copy_stream')

    def _stream_plan(self, sx):
        """[(stage, what runs eagerly behind it)] of the scene with its outputs leaving as they appear (run_streamed).  Stages only enqueue device work and
        leave their tensors in `sx` / on self (with captured graphs: static addresses); the second element of each pair is host code that runs after the stage
        in every mode - the trivial one-rank gathers and the `send` of finished blocks."""
        from .model.panoptic import view_chunks
        b = self.b
        send, finite = (lambda t: sx['send'](t)), (lambda t: sx['finite'](t))       # (looked up per call: a replayed plan outlives the call that captured it)
        sx.update(pm={}, kf_mf={}, blocks={}, host_pm={}, host_blocks={})

        def front():           # stage 2 up to the keyframes' features: the query decoder needs nothing else
            self.stage2a()
            rows = []
            for gi, g in enumerate(self.groups):
                n = len(g.idx)
                g.pointmaps = b.render(g.cat, n, g.h, g.w, self.bank) if not self.mixed else b.render(g.cat, n, g.h, g.w, self.bank, g.enc)
                sx['pm'][gi] = g.pointmaps
                P = g.guid.shape[0] // n if g.guid is not None else 0
                with b.precision(self.pan_amp):
                    if g.k:
                        fpn, mf = b.features(g.cat[:g.k * g.T], g.imgs[:g.k], g.k, g.h, g.w, None if g.guid is None else g.guid[:g.k * P],
                                             None if g.mm is None else g.mm[:g.k])
                        fm = b.attn_feats(mf, g.k, b.fpn_grid(g.h, g.w)[0])
                        self.d = fpn.shape[1]
                        rows.append(torch.cat([fpn, fm], dim=1))
                        sx['kf_mf'][gi] = mf
                    else:
                        rows.append(None)
            width = next(r.shape[1] for r in rows if r is not None)
            rows = [r if r is not None else torch.zeros(0, width, dtype=adt_of(rows), device=self.groups[0].imgs.device) for r in rows]
            self.both_send = self._kf_rows(rows)

        def after_front():
            self.gather2()
            for gi in range(len(self.groups)):
                sx['host_pm'][gi] = send(sx['pm'][gi])

        def decode():
            with b.precision(self.pan_amp):
                outq, head = b.decode(self.both_kf[:, :self.d].contiguous(), self.both_kf[:, self.d:].contiguous(), self.K, self.kf_fpn_grids,
                                      self.classes, self.kf_portrait)
                sx.update(outq=outq, head=head, logits=b.logits(head))
                for gi, g in enumerate(self.groups):
                    sx['blocks'][gi] = [b.masks_group(head, sx['kf_mf'].pop(gi))] if g.k else []
                sx['nkf'] = {gi: len(v) for gi, v in sx['blocks'].items()}

        def after_decode():
            finite(sx['outq'])
            for gi in range(len(self.groups)):
                sx['host_blocks'][gi] = [send(t) for t in sx['blocks'][gi][:sx['nkf'][gi]]]
            sx['host_logits'] = send(sx['logits'])

        plan = [(self.stage1, self.gather1), (front, after_front), (decode, after_decode)]
        work = [(gi, g.k + v0, c) for gi, g in enumerate(self.groups) for v0, c in view_chunks(len(g.idx) - g.k)]
        for w, (gi, a, c) in enumerate(work):
            def upscale(gi=gi, a=a, c=c, last=(w == len(work) - 1)):          # one upscaler pass of the views that are not keyframes + its mask logits
                g = self.groups[gi]
                P = g.guid.shape[0] // len(g.idx) if g.guid is not None else 0
                with b.precision(self.pan_amp):
                    _, mf = b.features(g.cat[a * g.T:(a + c) * g.T], g.imgs[a:a + c], c, g.h, g.w, None if g.guid is None else g.guid[a * P:(a + c) * P],
                                       None if g.mm is None else g.mm[a:a + c])
                    sx['blocks'][gi].append(b.masks_group(sx['head'], mf))
                if last:
                    for gg in self.groups:
                        gg.guid = gg.mm = None

            def after_upscale(gi=gi):
                sx['host_blocks'][gi].append(send(sx['blocks'][gi][len(sx['host_blocks'][gi])]))
            plan.append((upscale, after_upscale))
        return plan

    @torch.no_grad()
    def run_streamed(self, check_finite=True):
        """The scene with its outputs LEAVING FOR THE HOST WHILE IT STILL COMPUTES (`outdevice='cpu'`, the demo's call): 2.2 GB of pointmaps and mask logits
        at 50 views cost 40 ms behind the scene when copied at the end (to_outdevice).  Here stage 2b / 3 run keyframes first: pointmaps are on their way
        once the render is done; the keyframes' features feed the query decoder; then the mask head runs per pass of the upscaler (keyframes, then the
        other views in `view_chunks` passes) and every pass's block of mask logits is copied to pinned memory on a copy stream while the next pass computes -
        only the last block's copy is exposed.  Every per-view result is independent of the pass it is computed in: same bits as run().  Eager, or
        (use_graphs) one captured graph per stage of _stream_plan with the copies enqueued between the replays.
        Returns (results, scene dict, finite flag | None) like results(): host tensors (views of pinned blocks), class logits on the host, queries on the device."""
        assert self.streamable()
        b = self.b
        dev = self.groups[0].imgs.device
        cur, cp = torch.cuda.current_stream(), b.copy_stream(dev)
        keep, ok = [], [None]

        def finite(t):
            if check_finite:
                f = torch.isfinite(t).all()
                ok[0] = f if ok[0] is None else ok[0] & f

        def send(t):
            """device block -> pinned host block, asynchronously behind everything enqueued so far"""
            finite(t)
            host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            ev = torch.cuda.Event()
            ev.record(cur)
            with torch.cuda.stream(cp):
                cp.wait_event(ev)
                host.copy_(t, non_blocking=True)
            keep.append(t)                       # an eager pass's device block stays allocated until the copy stream is drained below
            return host

        with b.precision(self.amp):
            if not self.use_graphs:
                sx = dict(send=send, finite=finite)
                for stage, after in self._stream_plan(sx):
                    stage()
                    after()
            elif getattr(self, '_sgraphs', None) is None:
                # one eager pass of THIS plan first (its passes have their own shapes: position tables, scale vectors are made on first use), then one graph per stage
                warm = dict(send=lambda t: t, finite=lambda t: None)
                for stage, after in self._stream_plan(warm):
                    stage()
                    after()
                del warm
                torch.cuda.synchronize()
                sx = self._sx = dict(send=send, finite=finite)
                pool = torch.cuda.graph_pool_handle()
                self._sgraphs = []
                for stage, after in self._stream_plan(sx):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):
                        stage()
                    g.replay()
                    after()
                    self._sgraphs.append((g, after))
                if hasattr(b, 'pack_refs'):
                    self._refs = b.pack_refs()
            else:
                sx = self._sx
                sx.update(send=send, finite=finite, host_pm={}, host_blocks={})
                for g, after in self._sgraphs:
                    g.replay()
                    after()
        cp.synchronize()
        keep.clear()
        flag = None if ok[0] is None else bool(ok[0])
        res = {}
        for j, i in enumerate(self.mine):
            g, r = self.where[j]
            gi = self.groups.index(g)
            rr = r
            for blk in sx['host_blocks'][gi]:
                if rr < blk.shape[0]:
                    m = blk[rr]
                    break
                rr -= blk.shape[0]
            res[self.order[i]] = (sx['host_pm'][gi][r][None], m[None])
        outq = sx['outq']
        return res, {'pred_logits': sx['host_logits'][None], 'out_queries': (outq.clone() if self.use_graphs else outq)[:, None]}, flag

    def results(self, outdevice=None, copy=True):
        """({view_id: (pointmap [1,H,W,7], masks [1,Q,H/2,W/2])} of this rank's views, scene dict) after stage 3.
        With captured graphs the outputs live in buffers the next run() overwrites in place: copy=True clones them (moving them to
        `outdevice` is a copy already)."""
        outq, logits, masks = self.out
        own = (lambda t: t.clone()) if (copy and self.use_graphs and outdevice is None) else (lambda t: t)
        ms, pms = [], []
        for j, i in enumerate(self.mine):
            g, r = self.where[j]
            ms.append(masks[j][None])
            pms.append(g.pointmaps[r][None])
        if outdevice is not None:
            moved = to_outdevice(ms + pms, outdevice)
            ms, pms = moved[:len(ms)], moved[len(ms):]
        res = {self.order[i]: (own(pms[j]), own(ms[j])) for j, i in enumerate(self.mine)}
        return res, {'pred_logits': own(logits[None]), 'out_queries': own(outq[:, None])}


@torch.no_grad()
def run_scene(backend, get_image, V, H, W, K, classes, rank=0, world=1, group=None, outdevice=None, shapes=None, keyframes=None, amp=None, plan='replicated',
              minmax_bs=1, stream_bank=False):
    """Run one scene eagerly.  get_image(view_id) -> fp32 [3,H,W] on the rank's device (only called for owned views).
    Returns {view_id: (pointmap [1,H,W,7], masks [1,Q,H/2,W/2])} for the views this rank owns, plus the scene dict
    {'pred_logits' [1,Q,Ncls], 'out_queries' [Q,1,d]} (identical on every rank).  `shapes`: optional per-view (H, W);
    `keyframes`: optional explicit keyframe list in memory-build order (overrides the linspace schedule of K)."""
    Kc = len(keyframes) if keyframes is not None else (V if (K is None or K > V) else max(int(K), 2))
    plan = resolve_plan(plan, world, Kc)
    _, order, owner = assign_views(V, Kc, world, keyframes, plan)
    images = {order[i]: get_image(order[i]) for i in range(V) if owner[i] == rank}
    return SceneRunner(backend, images, V, H, W, K, classes, rank, world, group, use_graphs=False, shapes=shapes, keyframes=keyframes, amp=amp, plan=plan,
                       minmax_bs=minmax_bs, stream_bank=stream_bank).run(outdevice)


def _stage(name):
    """label of the launches of a section for hip.maxabs_telemetry() (a no-op list push / pop otherwise)"""
    from . import hip
    return hip.stage(name)


class HipBackend:
    """Stage backend on the HIP kernels (wraps a panst3r_amd.PanSt3R)."""

    def __init__(self, model):
        self.m = model
        self.patch_size = model.must3r_encoder.patch_size
        self.mask_dim = model.panoptic_decoder.mask_transformer.mask_dim
        self.De = model.must3r_encoder.embed_dim

    def precision(self, amp):
        from .model.common import precision
        return precision(amp)

    def pack_refs(self):
        from .model.common import HipModule
        refs = []
        for m in self.m.children():
            if isinstance(m, HipModule):
                refs.extend(m.pack_refs())
        return refs

    def same_format(self, a, b):
        from .model.common import amp_dtype
        return amp_dtype(a, quiet=True) == amp_dtype(b, quiet=True)

    def alloc_cat(self, rows, device):
        from .model.common import adt
        return torch.empty(rows, self.m._cat_width(), dtype=adt(), device=device)

    def alloc_enc(self, rows, device):
        from .model.common import adt
        return torch.empty(rows, self.De, dtype=adt(), device=device)

    def encode_enc(self, imgs, cat_rows, enc_rows=None):
        """enc_rows: additionally the tokens in the format in effect when `cat_rows` is kept in another one (the final LayerNorm's fp32 result rounded once,
        exactly what the LayerNorm kernel stores when it writes that format itself)"""
        with _stage('CroCo encoder'):
            self.m.encode_views(imgs, cat_rows, dino=False, enc_copy=enc_rows)

    def bank_f32(self, bank):
        return bank.f32

    def encode_rest_paired(self, imgs_enc, cat_enc, imgs_dino, cat_dino, enc_rows=None):
        with _stage('CroCo encoder + DINOv2 (paired towers)'):
            self.m.encode_views_paired(imgs_enc, cat_enc, imgs_dino, cat_dino, enc_rows)

    def encode_dino(self, imgs, cat_rows):
        with _stage('DINOv2'):
            self.m.encode_views(imgs, cat_rows, enc=False)

    def copy_stream(self, device):
        """a stream that only ever carries device -> host copies of finished blocks (SceneRunner.run_streamed): DMA engines, no kernels beside the scene's"""
        if getattr(self, '_copy', None) is None:
            self._copy = torch.cuda.Stream(device=device)
        return self._copy

    def side_stream(self, device):
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=device)
        return self._side

    _MASKED = {}
    _MASKED_LOG = {}       # per (device, split): the calibration timings of the streams that were tried

    def masked_streams(self, device, build_cus):
        """(stream A, stream B, CUs of A, CUs of B): two HIP streams whose queues may only use DISJOINT CU sets - A the first `build_cus` mask bits, B the
        rest (hipExtStreamCreateWithCUMask; the driver deals consecutive mask bits round-robin over the 8 XCDs, so both sets span all of them).
        One pair per (device, split), for the life of the process; None when no healthy pair could be made (the caller runs serially)."""
        key = (str(device), int(build_cus))
        if key not in HipBackend._MASKED:
            import ctypes
            rt = ctypes.CDLL('libamdhip64.so')
            ncu = torch.cuda.get_device_properties(device).multi_processor_count
            build_cus = max(8, min(int(build_cus), ncu - 8))
            words = (ncu + 31) // 32

            def mk(lo, hi):
                mask = (ctypes.c_uint32 * words)()
                for i in range(lo, hi):
                    mask[i // 32] |= 1 << (i % 32)
                h = ctypes.c_void_p()
                rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
                if rc != 0:
                    raise RuntimeError('hipExtStreamCreateWithCUMask failed (%d)' % rc)
                return h

            # Some masked queues come up SLOW on this platform: every kernel on them takes 3 - 6 x as long (profiles/r5_masked_kernel_probe.txt: a 768^3 GEMM 6.1 us
            # on the default stream, 7.7 / 8.8 us on the 192- / 96-CU streams, 34 / 31 us on the 128- / 64-CU streams created between them; which ones
            # alternates with the creation order, not with the mask).  Each stream is therefore calibrated when it is made - a captured chain of small GEMMs
            # against the same chain on the default stream - and re-created (up to 6 times) until it runs at the speed its CU share allows.
            a_ = torch.randn(768, 768, device=device).half()
            c_ = torch.empty(768, 768, device=device, dtype=torch.float16)

            def chain_us(stream):
                def chain():
                    for _ in range(64):
                        torch.mm(a_, a_, out=c_)
                chain()
                torch.cuda.synchronize(device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream):
                    chain()
                import time
                best = 1e9
                for _ in range(3):
                    torch.cuda.synchronize(device)
                    t0 = time.perf_counter()
                    if stream is None:
                        g.replay()
                    else:
                        with torch.cuda.stream(stream):
                            g.replay()
                    torch.cuda.synchronize(device)
                    best = min(best, (time.perf_counter() - t0) * 1e6 / 64)
                return best

            def good(lo, hi, ref_us):
                share = (hi - lo) / float(ncu)
                limit = ref_us * max(2.0, 1.2 / share)             # a 144-tile GEMM on a quarter of the chip may take ~4 x; a slow queue takes that on HALF of it
                tried = []
                for _ in range(6):
                    h = mk(lo, hi)
                    s_ = torch.cuda.ExternalStream(h.value, device=device)
                    us = chain_us(s_)
                    tried.append(round(us, 1))
                    if us <= limit:
                        return s_, tried
                    del s_
                    rt.hipStreamDestroy(h)
                return None, tried
            with torch.cuda.device(device):
                ref_us = chain_us(None)
                sa, ta = good(0, build_cus, ref_us)
                sb, tb = good(build_cus, ncu, ref_us) if sa is not None else (None, [])
                HipBackend._MASKED[key] = (sa, sb, build_cus, ncu - build_cus) if sb is not None else None
                HipBackend._MASKED_LOG[key] = dict(default_stream_us=round(ref_us, 1), build_stream_tries_us=ta, bulk_stream_tries_us=tb)
        return HipBackend._MASKED[key]

    def cu_budget(self, cus):
        from . import hip
        if os.environ.get('PST_CU_BUDGET') != '1':           # measurement switch (tools/overlap_bench.py); off: see the note at MASK_BUILD_CUS
            cus = 0
        hip.tune(hip.TUNE_CUS, int(cus))

    def rest_pairable(self, imgs_enc, imgs_dino):
        return self.m.paired_ok(imgs_enc, imgs_dino)

    def rest_begin(self, imgs_enc, imgs_dino):
        return self.m.paired_begin(imgs_enc, imgs_dino)

    def rest_layers(self, st, lo, hi):
        with _stage('CroCo encoder + DINOv2 (paired towers)'):
            self.m.paired_layers(st, lo, hi)

    def rest_finish(self, st, cat_enc, cat_dino, enc_rows=None):
        self.m.paired_finish(st, cat_enc, cat_dino, enc_rows)

    def enc_rows(self, cat, rows):
        return cat[:rows, :self.De].contiguous()

    def build_memory(self, enc_kf, K, grids, f32_bank=False):
        with _stage('memory build'):
            return self.m.build_memory(enc_kf, K, grids=grids, f32_bank=f32_bank)

    def bank_payload(self, bank):
        return [bank.K_all, bank.Vt_all] + ([bank.f32.K_all, bank.f32.Vt_all] if bank.f32 is not None else [])

    # ---- the bank per memory update (broadcast plan, streamed)
    def update_spans(self, K, grids):
        return self.m.memory_update_spans(K, grids)

    def bank_new(self, K, grids, device, f32_bank=False):
        return self.m.must3r_decoder.new_bank(device, sum(a * c for a, c in grids), f32=f32_bank)

    def build_step(self, bank, enc_kf, K, grids, u):
        with _stage('memory build'):
            self.m.build_memory_step(bank, enc_kf, K, grids, u)

    def bank_final(self, bank):
        return bank

    def _bank_views(self, bank, K, grids, u):
        _, _, tok, n = self.update_spans(K, grids)[u]
        banks = [bank] + ([bank.f32] if bank.f32 is not None else [])
        return [v for bk in banks for v in (bk.K_all[:, tok:tok + n], bk.Vt_all[:, :, tok:tok + n])]

    def bank_update_payload(self, bank, K, grids, u):
        """the entries update `u` appended, as contiguous copies (K rows [L, n, D] and V^T columns [L, D, n] per bank): what goes on the wire"""
        return [v.contiguous() for v in self._bank_views(bank, K, grids, u)]

    def bank_update_buffers(self, bank, K, grids, u):
        return [torch.empty(v.shape, dtype=v.dtype, device=v.device) for v in self._bank_views(bank, K, grids, u)]

    def bank_update_store(self, bank, K, grids, u, payload):
        for dst, src in zip(self._bank_views(bank, K, grids, u), payload):
            dst.copy_(src)

    def bank_alloc(self, K, grids, device, f32_bank=False):
        """an empty memory bank of the shape build_memory leaves behind (plan='broadcast': filled by the broadcast from rank 0)"""
        n = sum(a * c for a, c in grids)
        bank = self.m.must3r_decoder.new_bank(device, n, f32=f32_bank)
        for bk in (bank, bank.f32):
            if bk is not None:
                bk.n, bk.labels, bk.nimgs = n, list(range(K)), K
        return bank

    def render(self, cat, n, h, w, bank, enc=None):
        with _stage('render (MUSt3R decoder vs. memory)'):
            return self.m.render_views(cat, n, h, w, bank, enc=enc)

    def minmax_scaled(self):
        return self.m.panoptic_decoder.minmax_scaled()

    def scope_ids(self, ids, device):
        """scope id per view (by position in the scene's order) -> a static int32 device tensor (made once, outside any graph capture)"""
        return torch.tensor(list(ids), dtype=torch.int32).to(device)

    def table_of(self, rows, device):
        return torch.stack([r.float().reshape(3, 2) for r in rows]).to(device).contiguous()

    def minmax_local(self, imgs):
        """per (view, channel) (min, max) of the x0.5 guidance image: fp32 [n, 3, 2]"""
        from . import hip
        return hip.loftup_minmax(imgs, torch.empty(imgs.shape[0], 3, 2, dtype=torch.float32, device=imgs.device))

    def rows_in_order(self, tabs, idxs, n):
        out = torch.empty(n, *tabs[0].shape[1:], dtype=tabs[0].dtype, device=tabs[0].device)
        for t, idx in zip(tabs, idxs):
            out[self._index(idx, t.device)] = t
        return out

    def _index(self, idx, device):
        cache = self.__dict__.setdefault('_idx', {})          # static index tensors: made once (a host->device copy cannot be captured)
        key = (tuple(idx), str(device))
        if key not in cache:
            cache[key] = torch.tensor(list(idx), dtype=torch.long).to(device)
        return cache[key]

    def minmax_pool(self, table, scope):
        from . import hip
        return hip.minmax_merge(table.contiguous(), scope, torch.empty_like(table))

    def table_rows(self, table, pos):
        return table.index_select(0, self._index(pos, table.device)).contiguous()

    def guidance(self, imgs, h, w, mm=None):
        with _stage('LoftUp guidance'):
            return self.m.panoptic_decoder.guidance_tokens(imgs, h, w, mm=mm)

    def features(self, cat, imgs, n, h, w, guidance=None, mm=None):
        with _stage('InputMixer + upscaler'):
            return self.m.panoptic_decoder.features_tokens(cat, imgs, n, h, w, guidance=guidance, mm=mm)

    def fpn_grid(self, h, w):
        return self.m.panoptic_decoder.fpn_grid(h, w)

    def attn_feats(self, mf, k_local, grid):
        mt = self.m.panoptic_decoder.mask_transformer
        if k_local == 0:
            return torch.zeros(0, mt.mask_dim, dtype=mf.dtype, device=mf.device)
        return mt.attn_feats(mf[:k_local], grid)

    def decode(self, fpn_kf, fm_kf, K, grids, classes, portrait):
        pd = self.m.panoptic_decoder
        cls = pd.class_rows(classes, fpn_kf.device)
        with _stage('query decoder'):
            return pd.mask_transformer.decode_tokens(fpn_kf, fm_kf, list(grids), cls, list(portrait))

    def masks(self, head, mf, j):
        return self.m.panoptic_decoder.mask_transformer.masks_for(head.embed, mf[j])

    def masks_group(self, head, mf):
        return self.m.panoptic_decoder.mask_transformer.masks_for_group(head.embed, mf)

    def logits(self, head):
        return head.logits
