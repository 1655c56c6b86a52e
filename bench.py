#!/usr/bin/env python
"""bench.py -- frames/s of N-view 512-px PanSt3R panoptic inference on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--variant v2] [--views 50] [--keyframes 16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full scene forward (encoder + DINOv2 on every view, sequential keyframe-memory build, render of every
view, upscaler, panoptic query decoder, query x pixel masks of every view) on synthetic images already resident in
HBM, random-init full-size weights.  Default workload = BASELINE.json configs[3] (v2 / LoftUp, 50 views, 16
keyframes, 384x512), which fits one GPU; with --gpus N the SAME scene is view-sharded over N ranks (RCCL all-gathers
of keyframe tokens, panst3r_amd/scene.py), i.e. strong scaling.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0       # dense bf16 MFMA peak per MI355X (MI355X_MICROARCH.md chip table)


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (a 256-thread torch pool
    on a 16-core quota oversubscribes by 16x and runs ~50x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, 64))


TOLERANCE = {'pointmaps_rel_l2': 2e-2, 'mask_logits_rel_l2': 3e-2, 'mask_sign_agreement': 0.995, 'class_logits_max_abs': 0.05,
             'out_queries_rel_l2': 2e-2}        # 16-bit MFMA path vs the fp32 oracle


def cpu_baseline(variant, H, W, state, names, emb, threads, V=2, K=2, sharp=None):
    """Oracle (fp32 torch restatement, the "port" kind) timed on the host cores on a bounded sample: one V-view / K-keyframe scene of
    the same model at the same resolution (default 2 / 2).  Returns (record, oracle outputs, images, true shapes)."""
    from oracle.pipeline import build
    from panst3r_amd.synthetic import synth_image
    torch.set_num_threads(threads)
    model = build(variant)
    model.load_state_dict(state, strict=True)
    model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
    imgs = [synth_image(i, H, W) for i in range(V)]
    ts = torch.tensor([[H, W]] * V)
    # record the attention-mask bits of every query-decoder layer (mask_transformer.py:264-272) for the parity decomposition below
    mt = model.panoptic_decoder.mask_transformer
    amasks, heads = [], mt.forward_prediction_heads

    def recording_heads(*a, **k):
        out = heads(*a, **k)
        if out[2] is not None:
            m = out[2][0].clone()                      # [Q, NK] (identical for every head)
            m[m.all(-1)] = False                       # fully blocked rows attend everywhere (:172)
            amasks.append(m.to(torch.uint8))
        return out
    mt.forward_prediction_heads = recording_heads
    t0 = time.perf_counter()
    with torch.no_grad():
        ref = model.forward_inference_multi_ar(imgs, ts, names, num_keyframes=K, max_bs=1)       # the demo's max_bs=1 (per-view MinMaxScaler), as the timed runner
    dt = time.perf_counter() - t0
    mt.forward_prediction_heads = heads
    ref = (ref[0], dict(ref[1], attn_masks=amasks[:mt.num_layers]))
    return {'value': round(V / dt, 4), 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': '1 scene of %d views / %d keyframes at %dx%d, same %s model and weights, fp32 torch on %d host threads (%.1f s)'
                      % (V, K, H, W, variant, threads, dt)}, ref, imgs, ts


def _scene_errors(pm_h, pan_h, pm_o, pan_o):
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    mk = [(a.cpu(), b) for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks'])]
    num = sum(float((a.double() - b.double()).pow(2).sum()) for a, b in mk)
    den = sum(float(b.double().pow(2).sum()) for _, b in mk)
    agree = sum(float(((a > 0) == (b > 0)).sum()) for a, b in mk) / sum(b.numel() for _, b in mk)
    return {'pointmaps_rel_l2': sig(max(rel(a, b) for a, b in zip(pm_h, pm_o))),
            'pointmaps_rel_l2_per_view': [sig(rel(a, b)) for a, b in zip(pm_h, pm_o)],        # view id order (= keyframe index when V == K)
            'mask_logits_rel_l2_per_view': [sig(rel(a, b)) for a, b in mk],
            'mask_logits_rel_l2': sig((num / max(den, 1e-300)) ** 0.5),
            'mask_sign_agreement': round(agree, 5),
            'worst_view': {'mask_logits_rel_l2': sig(max(rel(a, b) for a, b in mk)),
                           'mask_sign_agreement': round(min(float(((a > 0) == (b > 0)).float().mean()) for a, b in mk), 5)},
            'class_logits_max_abs': sig(float((pan_h['pred_logits'].cpu() - pan_o['pred_logits']).abs().max())),
            'out_queries_rel_l2': sig(rel(pan_h['out_queries'], pan_o['out_queries']))}


def _within(e, worst=False):
    t = TOLERANCE
    m = e['worst_view'] if worst else e
    return bool(e['pointmaps_rel_l2'] <= t['pointmaps_rel_l2'] and m['mask_logits_rel_l2'] <= t['mask_logits_rel_l2'] and
                m['mask_sign_agreement'] >= t['mask_sign_agreement'] and e['class_logits_max_abs'] <= t['class_logits_max_abs'] and
                e['out_queries_rel_l2'] <= t['out_queries_rel_l2'])


def sig(x, digits=3):
    """x to `digits` significant digits (the fp32 mode's errors are ~1e-6: fixed decimals would print 0)"""
    return float('%.*g' % (digits, x))


def full_size_parity(model, dev, ref, imgs, ts, names, amp='fp16', K=2, panoptic_precision=None):
    """The oracle outputs of the cpu_baseline sample double as a FULL-SIZE parity check (the -m gpu tests use tiny configurations for
    most rows): the HIP path runs the same scene with the same weights; deviations against the tolerances SURVEY 8(d) states.
      * top level = the scene as a user runs it.  Mask logits are pooled over the pixels of all views ("sign agreement >= 99.5 % of
        pixels"); `worst_view` is the worst single view.
      * `decisions_matched` = the same scene with the HIP query decoder given the oracle's attention-mask BITS (200 x K*T per layer).
        The masked cross-attention thresholds mask logits at 0 and resets fully blocked rows (mask_transformer.py:172,264-272), so a query
        with few open keys is a discontinuous function of its inputs: a 1e-3 difference flips a bit and moves that query by several %.
        With the decisions matched every stated tolerance must hold for every view: what this measures is the arithmetic.
      * `attention_mask_bit_agreement` = how often the free-running HIP decoder takes the oracle's decision.
    The oracle is only the checker here."""
    pm_o, pan_o = ref
    mt = model.panoptic_decoder.mask_transformer
    inp = [i.to(dev) for i in imgs]
    with torch.no_grad():
        log = []
        with mt.instrument(log=log):       # (forward_inference_multi_ar runs eagerly unless cache_graphs=True: nothing is captured here)
            pm_h, pan_h = model.forward_inference_multi_ar(inp, ts, names, num_keyframes=K, amp=amp, max_bs=1, panoptic_precision=panoptic_precision)
    torch.cuda.synchronize()
    res = {'scene': '%d views / %d keyframes, full-size weights (the cpu_baseline sample)' % (len(imgs), K), 'amp': amp if amp else 'False (fp32 mode)'}
    if panoptic_precision:
        res['panoptic_precision'] = panoptic_precision
    res.update(_scene_errors(pm_h, pan_h, pm_o, pan_o))
    res['tolerance'] = dict(TOLERANCE)
    res['within_tolerance'] = _within(res)
    om = pan_o.get('attn_masks')
    if om:
        res['attention_mask_bit_agreement'] = round(min(float((a.cpu() == b).float().mean()) for a, b in zip(log, om)), 5)
        with torch.no_grad():
            with mt.instrument(forced=[m.to(dev) for m in om]):
                pm_f, pan_f = model.forward_inference_multi_ar(inp, ts, names, num_keyframes=K, amp=amp, max_bs=1, panoptic_precision=panoptic_precision)
        torch.cuda.synchronize()
        dm = _scene_errors(pm_f, pan_f, pm_o, pan_o)
        dm['within_tolerance_every_view'] = _within(dm, worst=True)
        res['decisions_matched'] = dm
    # which bound each number refers to: the top-level keys are the FREE-RUNNING scene (pooled over views), the
    # `decisions_matched` keys the run with the oracle's attention-mask bits; the stated SURVEY 8(d) tolerances are `tolerance`
    res['bounds_met'] = {'free_running_stated_tolerances': res['within_tolerance'],
                         'decisions_matched_stated_tolerances_every_view': res.get('decisions_matched', {}).get('within_tolerance_every_view')}
    return res


REAL_STDOUT = 1
OVERLAP_PICK = {}        # --overlap auto: per measured mode, what the warm-up timing chose


def cpu_c1(threads):
    """BASELINE configs[0] (the reference's own CPU-runnable case): 2 views at 224x224, fp32, v1 model, PanSt3R.forward style."""
    from oracle.pipeline import build
    from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings
    torch.set_num_threads(threads)
    model = fill_module_(build('v1'), seed=1)
    names, emb = synth_class_embeddings(100)
    model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
    imgs = torch.stack([synth_image(i, 224, 224, 7) for i in range(2)])[None]
    ts = torch.tensor([[[224, 224]] * 2])
    t0 = time.perf_counter()
    with torch.no_grad():
        model(imgs, ts, names)
    dt = time.perf_counter() - t0
    return {'config': 'C1: 2 views / 2 keyframes, 224x224, v1, fp32 torch', 'frames_per_s': round(2 / dt, 4), 'seconds': round(dt, 2)}


def hbm_stage_table(timer, V, H, W, variant):
    """Per-stage HBM GB/s of the sub-stages SURVEY 8(d) lists as HBM-bound, from the instrumented step's HIP events: ALGORITHMIC bytes
    (what the stage must move once) / event time, against the 8 TB/s peak."""
    from panst3r_amd import flops as F
    out = {}
    by = timer.by_tag()
    Q, C, P = 200, (384 if variant == 'v2' else 256), (H // 2) * (W // 2)

    def add(key, pred, nbytes):
        ms = sum(d['ms'] for (n, t), d in by.items() if pred(n, t))
        cnt = sum(d['launches'] for (n, t), d in by.items() if pred(n, t))
        if cnt:
            out[key] = {'launches': cnt, 'bytes_per_launch': int(nbytes), 'avg_us': round(1e3 * ms / cnt, 2),
                        'GBps': round(nbytes * cnt / (ms * 1e-3) / 1e9, 1), 'frac_of_8TBps': round(nbytes * cnt / (ms * 1e-3) / 8e12, 4)}
    add('mask_head (query x pixel einsum, per view)', lambda n, t: n.startswith('gemm') and t[:3] == (Q, P, C), C * P * 2 + Q * P * 4)
    T = (H // 16) * (W // 16)
    pm = [(n, t) for (n, t) in by if n.startswith('gemm') and len(t) > 9 and t[9] == 'ps' and t[1] == 1792]
    for n, t in pm:
        add('pointmap head + pixel-shuffle store (M=%d)' % t[0], lambda nn, tt, t=t: nn == n and tt == t, t[0] * 768 * 2 + t[0] * 1792 * 4)
    summ = timer.summary()
    for k in ('mask_head_kernel', 'layernorm', 'rowstats', 'groupnorm_stats', 'groupnorm_apply', 'loftup_guidance_gn', 'mean4', 'patch_rows'):
        d = summ.get(k)
        if d and d['launches']:
            out[k] = {'launches': d['launches'], 'bytes_per_launch': int(d['bytes'] / d['launches']), 'avg_us': round(1e3 * d['ms'] / d['launches'], 2),
                      'GBps': round(d['bytes'] / (d['ms'] * 1e-3) / 1e9, 1), 'frac_of_8TBps': round(d['bytes'] / (d['ms'] * 1e-3) / 8e12, 4)}
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): re-execute this command line under torch.distributed.run, one rank per
    GPU of this node (rendezvous on 127.0.0.1, a free port), pass rank 0's JSON line through and return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env, stdout=REAL_STDOUT, stderr=2).returncode


def launch_selftest(args):
    """--launch-selftest: the launch / rendezvous / one-line-output plumbing of `--gpus N` without any model work - N ranks (RCCL when every rank
    has a GPU, else gloo on the CPU), one barrier and one all_gather; rank 0 prints the JSON line.  tests/test_bench_launch.py runs it with 2
    ranks on gloo in the CPU container."""
    import torch.distributed as dist
    world, rank, local = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    gpu = torch.cuda.is_available() and torch.cuda.device_count() >= world
    if gpu:
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    dev = torch.device('cuda', local) if gpu else torch.device('cpu')
    t = torch.full((4,), float(rank), device=dev)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.barrier()
    dist.all_gather(outs, t)
    ok = all(float(o[0]) == r for r, o in enumerate(outs))
    dist.barrier()
    if rank == 0:
        os.write(REAL_STDOUT, (json.dumps({'launch_selftest': True, 'n_gpus': world, 'world_size_seen': dist.get_world_size(), 'backend': dist.get_backend(),
                                           'all_gather_ok': ok, 'gpus_requested': args.gpus}) + '\n').encode())
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--variant', default='v2', choices=['v1', 'v2'])
    ap.add_argument('--views', type=int, default=50)
    ap.add_argument('--keyframes', type=int, default=16)
    ap.add_argument('--height', type=int, default=384)
    ap.add_argument('--width', type=int, default=512)
    ap.add_argument('--amp', default='fp16', choices=['fp16', 'bf16'], help="16-bit MFMA operand format (reference --amp, tools/demo_panst3r.py:88)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-c2', action='store_true', help='also time the reduced C2 sample (v1, 8 views / 8 keyframes) on the host: ~1 min')
    ap.add_argument('--no-alt-dtype', action='store_true', help='skip the short measurement of the other 16-bit format')
    ap.add_argument('--no-depth-parity', action='store_true', help='skip the K = 16 parity scene (16 views = 16 keyframes = BASELINE configs[2]; ~1.5 min of host time)')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--parity-c4', action='store_true', help='also compare the TIMED configuration itself (v2, 50 views / 16 keyframes, full size) with the oracle on the host: several minutes')
    ap.add_argument('--parity-c5', action='store_true', help='also compare BASELINE configs[4] (v2, 200 views / 32 keyframes, fp16, full size) with the oracle on the host: ~15 min, ~12 GB of host memory')
    ap.add_argument('--c5-views', type=int, default=200, help='views of the --parity-c5 scene (32 keyframes stay: the memory size class of configs[4]; the host oracle costs ~4 s / view)')
    ap.add_argument('--overlap', default='auto', choices=['auto', 'off', 'masked', 'plain'],
                    help="stage 2 of the scene: 'off' = the memory build and the bulk encoder / DINOv2 work back to back on one stream; 'masked' = the build on "
                         "a CU-masked stream beside the first tower layers on the other CUs (disjoint CU sets: panst3r_amd/scene.py); 'auto' (default) = both "
                         "are captured and timed during warm-up, the faster one runs the timed steps (same bits either way); 'plain' = MEASUREMENT ONLY, two "
                         "ordinary streams (loses writes on this platform: tests/diag/cu_mask_two_queue.py)")
    ap.add_argument('--no-overlap', action='store_true', help=argparse.SUPPRESS)       # former switch; serial is the default now
    ap.add_argument('--eager', action='store_true', help='launch every kernel from the host instead of replaying HIP graphs')
    ap.add_argument('--plan', default='auto', choices=['auto', 'replicated', 'broadcast'],
                    help="multi-GPU plan (panst3r_amd/scene.py): every rank repeats the memory build | rank 0 builds and broadcasts the banks; "
                         "auto = broadcast from 4 ranks on (projection: profiles/r3_shard_estimate.txt)")
    ap.add_argument('--stream-bank', action='store_true', help="plan 'broadcast': send the bank per memory update with asynchronous broadcasts beside the build "
                                                               "(opt-in; default = one event-ordered broadcast behind the build, panst3r_amd/scene.py)")
    ap.add_argument('--tune', action='append', default=[], metavar='KNOB=VALUE', help='pst_tune knob for A/B measurements (G256_PP, PAIR, PAIR_RES); results never depend on a knob')
    ap.add_argument('--launch-selftest', action='store_true', help='only exercise the rank launch / rendezvous / output plumbing (no model work; CPU + gloo when there is no GPU per rank)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started the way the driver starts `--gpus 1` (plain `python bench.py --gpus N`): become the launcher of N ranks
        raise SystemExit(spawn_ranks(args.gpus))
    if args.launch_selftest:
        return launch_selftest(args)

    import torch.distributed as dist
    from panst3r_amd import hip
    from panst3r_amd.panst3r import CONFIG_V1, CONFIG_V2, build_from_config
    from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings
    from panst3r_amd import flops as F

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_dist = world > 1 or os.environ.get('PST_FORCE_DIST') == '1'       # PST_FORCE_DIST=1: exercise the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if world == 1 and 'MASTER_ADDR' not in os.environ:      # PST_FORCE_DIST=1 without a launcher: a one-rank rendezvous of its own
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
            os.environ['MASTER_ADDR'] = '127.0.0.1'
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    hip.lib()      # fail loudly if the HIP library is missing
    for kv in args.tune:
        k, v = kv.split('=')
        hip.tune(getattr(hip, 'TUNE_' + k.upper()), int(v))

    V, K, H, W = args.views, args.keyframes, args.height, args.width
    model = build_from_config(CONFIG_V2 if args.variant == 'v2' else CONFIG_V1).eval()
    fill_module_(model, seed=1)
    names, emb = synth_class_embeddings(100)
    host_legs = rank == 0 and world == 1 and not args.no_cpu_baseline
    state = {k: v.clone() for k, v in model.state_dict().items()} if host_legs else None
    model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
    model.to(dev)

    from panst3r_amd.scene import assign_views, resolve_plan
    args.plan = resolve_plan(args.plan, world, min(K, V))
    _, order, owner = assign_views(V, K, world, plan=args.plan)
    mine = {order[i] for i in range(V) if owner[i] == rank}
    images = {i: synth_image(i, H, W).to(dev) for i in sorted(mine)}        # inputs resident in HBM before timing

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(amp, steps, warmup, instrument, panoptic_precision=None):
        """W untimed warm-up steps (the first also captures the three HIP graphs), then EXACTLY `steps` timed steps between two fences."""
        mk = lambda ov: model.scene_runner(images, V, H, W, names, num_keyframes=K, use_graphs=not args.eager, overlap=ov, amp=amp, plan=args.plan,
                                           panoptic_precision=panoptic_precision, stream_bank=args.stream_bank)
        if args.overlap == 'auto' and not args.eager and world == 1:
            from panst3r_amd.scene import pick_overlap
            runner, picked = pick_overlap(mk)
            OVERLAP_PICK.setdefault(str(amp) + '/' + str(panoptic_precision), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in picked.items()})
        else:
            runner = mk({'auto': False, 'off': False, 'masked': 'masked', 'plain': True}[args.overlap])
        for _ in range(max(warmup, 1)):
            runner.run(copy=False)
        timer = None
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        if use_dist:
            runner.coll_events = []          # HIP events around every eager collective of the timed steps (2 events per collective, no sync)
        fence()
        t0 = time.perf_counter()
        for s in range(steps):
            marks[s].record()
            if s == steps - 1 and instrument:
                # the LAST timed step runs eagerly (not as a graph replay) with HIP events around every MFMA-kernel and HBM-stage launch; it
                # runs the two branches of stage 2 back-to-back also under --overlap (event durations not inflated by the other branch)
                timer = hip.KernelTimer()
                hip.TIMER = timer
                runner.run(eager=True, serial=True, copy=False)
            else:
                runner.run(copy=False)
        marks[steps].record()
        hip.TIMER = None
        fence()
        elapsed = time.perf_counter() - t0
        pk = OVERLAP_PICK.get(str(amp) + '/' + str(panoptic_precision))
        if pk is not None and pk.get('serial_digest') is not None and 'after_timed_steps_identical' not in pk:
            # outside the timed region: one more replay of the form that was timed, its outputs against the serial scene's bits of the warm-up
            from panst3r_amd.scene import output_digest
            pk['after_timed_steps_identical'] = output_digest(*runner.run(copy=False)) == pk['serial_digest']
            pk['serial_digest'] = '%016x' % (pk['serial_digest'] & (2 ** 64 - 1))
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t)
        per = sorted(marks[s].elapsed_time(marks[s + 1]) for s in range(steps - (1 if instrument else 0)))      # graph-replay steps only
        coll = None
        if use_dist:
            # per collective: median over the timed steps on this rank, then the max over ranks (a collective ends when its slowest rank does;
            # the time a rank WAITS for a late peer is inside its own measurement, so this is transfer + skew, an upper bound on the transfer)
            med = {k: sorted(v)[len(v) // 2] for k, v in runner.collective_ms().items()}
            cnames = sorted(med)
            t = torch.tensor([med[k] for k in cnames], device=dev, dtype=torch.float64)
            lo = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            coll = {k: {'max_over_ranks_ms': round(float(a), 3), 'min_over_ranks_ms': round(float(b), 3)} for k, a, b in zip(cnames, t, lo)}
        del runner
        return elapsed, (per[len(per) // 2] if per else None), timer, coll

    elapsed, median_ms, timer, coll_ms = measure(args.amp, args.steps, args.warmup, not args.no_kernel_timing)

    if rank == 0:
        fps = V * args.steps / elapsed
        scene_flops = F.scene_flops(H, W, V, K, args.variant)
        out = {
            'metric': 'frames/sec (whole node), N-view 512px panoptic inference', 'value': round(fps, 3), 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f16' if args.amp == 'fp16' else 'bf16 (f16 operands in the panoptic decoder)', 'data': 'synthetic',
            'config': {'workload': 'PanSt3R_%s_512 scene: %d views, %d keyframes, %dx%d, 100 classes, random-init full-size weights'
                                   % (args.variant, V, K, H, W),
                       'variant': args.variant, 'views': V, 'keyframes': K, 'resolution': [H, W],
                       'parallelism': 'views sharded over %d rank(s), plan %s' % (world, args.plan),
                       'operands': "%s MFMA operands, fp32 accumulate / residual streams / softmax / statistics (reference --amp %s, tools/demo_panst3r.py:88)"
                                   % ('f16' if args.amp == 'fp16' else 'bf16', args.amp),
                       'launch': 'eager' if args.eager else 'HIP-graph replay (3 graphs per scene; last timed step eager + HIP-event instrumented)',
                       'overlap_auto': OVERLAP_PICK.get(str(args.amp) + '/None'),
                       'overlap_streams': {k[1]: v for k, v in __import__('panst3r_amd.scene', fromlist=['HipBackend']).HipBackend._MASKED_LOG.items()} or None,
                       'overlap': {'auto': 'auto: the serial and the CU-masked two-queue stage 2 are both captured and timed during warm-up, the faster runs '
                                           '(overlap_auto; bit-identical results, CHECKED in this run: overlap_auto.masked_identical over the trial replays, .after_timed_steps_identical after the timed steps; tests/test_hip_fullsize.py::test_full_size_masked_overlap_equals_serial)',
                                   'off': 'off (one stream)', 'plain': 'memory build || non-keyframe encoder + DINOv2 on two ordinary streams (measurement only)',
                                   'masked': 'memory build on a CU-masked stream || first layers of the two ViT-L towers on the other CUs (panst3r_amd/scene.py)'}[args.overlap],
                       'median_ms_per_graph_step': None if median_ms is None else round(median_ms, 3),
                       'median_frames_per_s': None if median_ms is None else round(V / (median_ms * 1e-3), 2),
                       'scene_algorithmic_tflop': round(scene_flops / 1e12, 2),
                       'scene_mfma_frac': round(scene_flops / (elapsed / args.steps) / world / (PEAK_BF16_TFLOPS * 1e12), 4)},
        }
        if use_dist:
            # what the transport did, measured, next to the one-GPU projection of the same plan (profiles/r3_shard_estimate.txt: every rank's stage
            # graphs replayed on ONE GPU, all-gathers counted as zero, the 453 MB bank broadcast ASSUMED at 100 GB/s)
            proj = {('replicated', 2): 102.2, ('replicated', 4): 69.2, ('replicated', 8): 52.3, ('broadcast', 2): 121.9, ('broadcast', 4): 56.2, ('broadcast', 8): 42.5}
            out['multi_gpu'] = {'backend': dist.get_backend(), 'rccl_world_size': dist.get_world_size(), 'plan': args.plan,
                                'collectives_measured': coll_ms,
                                'projected_ms_per_scene_one_gpu_estimate': proj.get((args.plan, world)) if (V, K, args.variant) == (50, 16, 'v2') else None,
                                'measured_ms_per_scene': round(1e3 * elapsed / args.steps, 3),
                                'bank_bytes': (2 * 12 * K * (H // 16) * (W // 16) * 768 * 2) if args.plan == 'broadcast' else 0,
                                'bank_transfer': ('per memory update, asynchronous broadcasts beside the build (--stream-bank)' if args.stream_bank else
                                                  'one event-ordered broadcast behind the build (default: no collective shares a CU with compute)') if args.plan == 'broadcast' else None}
        if timer is not None and os.environ.get('PST_SHAPE_PROFILE') == '1':      # per-shape table of the instrumented step (stderr)
            rows = sorted(timer.by_tag().items(), key=lambda kv: -kv[1]['ms'])
            for (name, tag), d in rows[:48]:
                print('%-24s %-62s x%-4d %8.2f ms %7.1f TF' % (name, tag, d['launches'], d['ms'], d['flops'] / (d['ms'] * 1e-3) / 1e12), file=sys.stderr)
        if timer is not None:
            summ = {k: v for k, v in timer.summary().items() if v['flops'] > 0}
            dom = max(summ, key=lambda k: summ[k]['ms'])
            d = summ[dom]
            ach = d['flops'] / (d['ms'] * 1e-3) / 1e12
            # HBM bytes per launch of that kernel from PMC passes of THIS kernel source (tools/pmc_profile.sh: FETCH_SIZE and WRITE_SIZE in separate
            # rocprofv3 --pmc runs of this workload, FETCH_SIZE doubled per the gfx950 note); null when the committed summary is of other sources
            traffic, traffic_note = None, 'no PMC summary for these kernel sources'
            try:
                from panst3r_amd.build import source_hash
                for fn in sorted(os.listdir(os.path.join(ROOT, 'profiles')), reverse=True):
                    if not fn.endswith('_pmc_summary.json'):
                        continue
                    pmc = json.load(open(os.path.join(ROOT, 'profiles', fn)))
                    if pmc.get('_source_hash') != source_hash():
                        continue
                    # PMC names carry every template argument ("gemm_kernel<4, 4, false, 2, true>"); the dispatch name is a prefix of it
                    # ("gemm_kernel<4,4,false>") or the bare kernel name ("gemm256p_kernel": both epilogue classes, launch-weighted)
                    key = dom.replace(',', ', ')
                    pref = key[:-1] + ',' if key.endswith('>') else key + '<'
                    es = [v for k, v in pmc.items() if isinstance(v, dict) and (k == key or k.startswith(pref))]
                    if es and V == 50 and K == 16 and args.variant == 'v2':
                        n = sum(e['launches'] for e in es)
                        traffic = int(sum(e['launches'] * (e['hbm_read_bytes_per_launch'] + e['hbm_write_bytes_per_launch']) for e in es) / n)
                        traffic_note = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same kernel sources: %s)' % (fn, pmc['_source_hash'])
                    break
            except Exception:
                pass
            out['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': round(ach, 1), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                               'frac': round(ach / PEAK_BF16_TFLOPS, 4), 'traffic': traffic, 'traffic_source': traffic_note, 'launches': d['launches'],
                               'avg_launch_us': round(1e3 * d['ms'] / d['launches'], 2),
                               'avg_launch_gflop': round(d['flops'] / d['launches'] / 1e9, 3)}
            out['kernels'] = {k: {'launches': v['launches'], 'ms': round(v['ms'], 2),
                                  'tflops': round(v['flops'] / max(v['ms'], 1e-9) / 1e9, 1)} for k, v in sorted(summ.items())}
            out['hbm_stages'] = hbm_stage_table(timer, V, H, W, args.variant)
            # the same stages stand-alone: one captured HIP graph per stage on scene-sized cold operands, replay timed as a whole (the eager numbers
            # above carry 5-10 us of HIP-event overhead per 20-35 us launch)
            try:
                from panst3r_amd.stagebench import standalone_hbm_stages
                out['hbm_stages_standalone'] = standalone_hbm_stages(dev, args.variant, H, W, torch.float16 if args.amp == 'fp16' else torch.bfloat16)
            except Exception as e:          # a measurement leg must not take the bench line down
                out['hbm_stages_standalone'] = {'error': repr(e)}
        if host_legs and not args.no_alt_dtype:
            alt = 'bf16' if args.amp == 'fp16' else 'fp16'
            e2, m2, _, _ = measure(alt, max(3, args.steps // 4), 1, False)
            out['alt_dtype'] = {'dtype': 'bf16 (+ f16 panoptic decoder)' if alt == 'bf16' else 'f16', 'value': round(V * max(3, args.steps // 4) / e2, 3),
                                'note': 'the other 16-bit format of the reference (--amp %s), same scene, %d timed steps' % (alt, max(3, args.steps // 4))}
        if host_legs and not args.no_alt_dtype:
            try:          # the reference's default mode (amp=False: fp32 end to end) on the fp32 kernels: the precision path, not the benchmark
                e3, _, _, _ = measure(False, 2, 1, False)
                out['fp32_mode'] = {'dtype': 'f32 (3 x f16 split operands)', 'value': round(V * 2 / e3, 3), 'unit': 'frames/s',
                                    'note': 'amp=False: float32 activations, every GEMM / attention contraction as three f16 MFMAs on split operands x = hi + lo (22 mantissa bits; '
                                            'csrc/split.hip, attn_x3.hip), same scene, 2 timed steps'}
                e3x, _, _, _ = measure('fp32_exact', 1, 1, False)
                out['fp32_mode']['fp32_exact'] = {'value': round(V * 1 / e3x, 3), 'note': "amp='fp32_exact': the fp32-input-MFMA kernels (exact fp32 products), 1 timed step"}
            except Exception as e:
                out['fp32_mode'] = {'error': repr(e)}
        if host_legs and not args.no_alt_dtype:
            try:          # the reference's own precision placement under --amp: fp32 panoptic decoder + fp32 render / DINOv2 of the views that are not keyframes
                e4, _, _, _ = measure(args.amp, 3, 1, False, panoptic_precision='reference')
                out['reference_amp_placement'] = {'value': round(V * 3 / e4, 3), 'unit': 'frames/s', 'dtype': out['dtype'] + ' + f32',
                                                  'note': "amp=%r, panoptic_precision='reference' (panst3r.py:174-175,204-245,268: autocast covers the encoder, the memory build and the "
                                                          "keyframes' render + DINOv2 only), same scene, 3 timed steps" % args.amp}
            except Exception as e:
                out['reference_amp_placement'] = {'error': repr(e)}
        if host_legs and not args.no_alt_dtype:
            # the API entry exactly as the reference's demo calls it (tools/demo_panst3r.py:232-233: max_bs=1, outdevice='cpu'; --amp defaults to False):
            # one-off eager call and repeated calls with cache_graphs=True (captured HIP graphs replayed; not in the reference), per format.  Includes what the
            # timed runner leaves out: stacking the inputs, the finite check, and the device -> host copy of every pointmap and mask (2.2 GB at 50 views).
            try:
                import torch as _t
                ts_all = _t.tensor([[H, W]] * V)
                imgs_all = [synth_image(i, H, W).to(dev) for i in range(V)]

                def api(amp, graphs, n, outdevice='cpu'):
                    model.clear_runners()
                    for _ in range(2 if graphs else 1):            # warm-up (weights packed; with graphs: first call eager, second captures)
                        model.forward_inference_multi_ar(imgs_all, ts_all, names, num_keyframes=K, max_bs=1, outdevice=outdevice, amp=amp, cache_graphs=graphs)
                    _t.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(n):
                        model.forward_inference_multi_ar(imgs_all, ts_all, names, num_keyframes=K, max_bs=1, outdevice=outdevice, amp=amp, cache_graphs=graphs)
                    _t.cuda.synchronize()
                    return round(V * n / (time.perf_counter() - t0), 2)
                out['api_entry'] = {'call': "forward_inference_multi_ar(imgs, true_shape, classes, num_keyframes=%d, max_bs=1, outdevice='cpu', amp=...) as tools/demo_panst3r.py:232-233; frames/s incl. the device -> host copy of all outputs" % K,
                                    'amp_False_eager': api(False, False, 2), 'amp_False_cache_graphs': api(False, True, 2),
                                    'amp_fp16_eager': api('fp16', False, 3), 'amp_fp16_cache_graphs': api('fp16', True, 3)}
                # the same entry with the outputs LEFT ON THE DEVICE (outdevice=None) and cached graphs: stage 2 in the form the runner-level selection picks
                # (PanSt3R.stage2_overlap='auto' -> scene.pick_overlap, what the timed runner of `value` goes through), incl. input stacking + the finite check
                out['api_entry']['amp_fp16_cache_graphs_on_device'] = api('fp16', True, 5, None)
                out['api_entry']['stage2_form_on_device'] = (getattr(model, 'stage2_pick', None) or {}).get('chosen')
                model.clear_runners()
                del imgs_all
            except Exception as e:
                out['api_entry'] = {'error': repr(e)}
        if host_legs:
            threads = usable_cores()
            out['cpu_baseline'], ref, ref_imgs, ref_ts = cpu_baseline(args.variant, H, W, state, names, emb, threads)
            samples = {'C1': cpu_c1(threads)}
            if args.cpu_baseline_c2:
                m1 = build_from_config(CONFIG_V1).eval()
                fill_module_(m1, seed=1)
                rec, _, _, _ = cpu_baseline('v1', H, W, {k: v.clone() for k, v in m1.state_dict().items()}, names, emb, threads, V=8, K=8)
                samples['C2_reduced'] = {'config': 'C2 reduced: v1, 8 views / 8 keyframes, %dx%d' % (H, W), 'frames_per_s': rec['value']}
                del m1
            # C4 extrapolated from the measured sample by algorithmic FLOPs (labelled as such)
            sample_flops = F.scene_flops(H, W, 2, 2, args.variant)
            cpu_tflops = sample_flops / (2 / out['cpu_baseline']['value']) / 1e12
            samples['C4_extrapolated'] = {'config': 'C4: %s, %d views / %d keyframes (EXTRAPOLATED by algorithmic FLOPs from the measured sample, not run)' % (args.variant, V, K),
                                          'frames_per_s': round(V / (scene_flops / 1e12 / cpu_tflops), 4), 'host_tflops_fp32': round(cpu_tflops, 3)}
            out['cpu_baseline']['other_configs'] = samples
            out['parity'] = full_size_parity(model, dev, ref, ref_imgs, ref_ts, names, args.amp)
            out['parity']['fp32_mode'] = full_size_parity(model, dev, ref, ref_imgs, ref_ts, names, False)
            if not args.no_depth_parity:
                # parity at the benchmark's own memory depth: 16 views = 16 keyframes (BASELINE configs[2] as stated; 15 sequential memory updates
                # as in the timed 50 / 16 scene), oracle on the host (timed: a MEASURED C3 next to the extrapolated C4), HIP path free-running and
                # decisions-matched
                rec16, ref16, imgs16, ts16 = cpu_baseline(args.variant, H, W, state, names, emb, threads, V=16, K=16)
                samples['C3_measured'] = {'config': 'C3: %s, 16 views / 16 keyframes, %dx%d, fp32 torch on %d host threads' % (args.variant, H, W, threads),
                                          'frames_per_s': rec16['value']}
                out['parity']['K16'] = full_size_parity(model, dev, ref16, imgs16, ts16, names, args.amp, K=16)
                out['parity']['K16_fp32_mode'] = full_size_parity(model, dev, ref16, imgs16, ts16, names, False, K=16)      # amp=False (3 x f16 split operands) at the bench's memory depth
                # BASELINE configs[1..2] NAME bf16: what that format meets of the five stated tolerances on configs[2] (v2, 16 = 16), all-16-bit and with the
                # reference's own placement (fp32 panoptic decoder) - said here, not hidden in a relaxed assert
                try:
                    b16 = full_size_parity(model, dev, ref16, imgs16, ts16, names, 'bf16', K=16)
                    b16p = full_size_parity(model, dev, ref16, imgs16, ts16, names, 'bf16', K=16, panoptic_precision='amp')
                    b16r = full_size_parity(model, dev, ref16, imgs16, ts16, names, 'bf16', K=16, panoptic_precision='reference')
                    f16r = full_size_parity(model, dev, ref16, imgs16, ts16, names, 'fp16', K=16, panoptic_precision='reference')

                    def met(e, worst=False):
                        t = TOLERANCE
                        m = e['worst_view'] if worst else e
                        return sum([e['pointmaps_rel_l2'] <= t['pointmaps_rel_l2'], m['mask_logits_rel_l2'] <= t['mask_logits_rel_l2'],
                                    m['mask_sign_agreement'] >= t['mask_sign_agreement'], e['class_logits_max_abs'] <= t['class_logits_max_abs'],
                                    e['out_queries_rel_l2'] <= t['out_queries_rel_l2']])
                    brief = lambda e: {k: e[k] for k in ('pointmaps_rel_l2', 'mask_logits_rel_l2', 'mask_sign_agreement', 'class_logits_max_abs', 'out_queries_rel_l2', 'worst_view')}
                    out['configs_named_bf16'] = {'scene': 'configs[2]: v2, 16 views = 16 keyframes, full size, free-running, vs the fp32 oracle',
                                                 'placement': "amp='bf16' (default): bf16 operands where the reference autocasts (encoder, DINOv2, memory build, render), f16 operands in "
                                                              "the panoptic decoder, which the reference runs in fp32 (panst3r.py:236-245); panoptic_precision='amp' = bf16 there too",
                                                 'bf16_all_16_bit': '%d of 5 stated tolerances' % met(b16), 'bf16_all_16_bit_worst_view': '%d of 5' % met(b16, True),
                                                 'bf16_all_16_bit_errors': brief(b16),
                                                 'bf16_pure': '%d of 5 stated tolerances' % met(b16p), 'bf16_pure_errors': brief(b16p),
                                                 'bf16_reference_placement': '%d of 5 stated tolerances' % met(b16r), 'bf16_reference_placement_errors': brief(b16r),
                                                 'f16_reference_placement_errors': brief(f16r)}
                except Exception as e:
                    out['configs_named_bf16'] = {'error': repr(e)}
                del ref16
            if args.parity_c4:
                recc, refc, imgsc, tsc = cpu_baseline(args.variant, H, W, state, names, emb, threads, V=V, K=K)
                samples['C4_measured'] = {'config': 'C4: %s, %d views / %d keyframes, %dx%d, fp32 torch on %d host threads' % (args.variant, V, K, H, W, threads),
                                          'frames_per_s': recc['value']}
                out['parity']['C4'] = full_size_parity(model, dev, refc, imgsc, tsc, names, args.amp, K=K)
                del refc
            if args.parity_c5:
                rec5, ref5, imgs5, ts5 = cpu_baseline(args.variant, H, W, state, names, emb, threads, V=args.c5_views, K=32)
                samples['C5_measured'] = {'config': 'C5: %s, %d views / 32 keyframes, %dx%d, fp32 torch on %d host threads' % (args.variant, args.c5_views, H, W, threads), 'frames_per_s': rec5['value']}
                out['parity']['C5'] = full_size_parity(model, dev, ref5, imgs5, ts5, names, 'fp16', K=32)
                del ref5
            if not out['parity']['within_tolerance']:
                print('WARNING: full-size parity outside the stated tolerance: %s' % out['parity'], file=sys.stderr)
        sys.stdout.flush()
        os.write(REAL_STDOUT, (json.dumps(out) + '\n').encode())      # the one line on stdout
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    # stdout carries exactly ONE line, the JSON result: everything else any library prints there (RCCL's version banner comes through C
    # stdio at process exit) is routed to stderr by pointing fd 1 at fd 2 and keeping the real stdout aside for the final write.
    sys.stdout.flush()
    REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    main()
