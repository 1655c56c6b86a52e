#!/usr/bin/env python
"""Per-rank critical path of the view-sharded scene, measured on ONE GPU: N SceneRunners stepped in lock-step (the all-gathers replaced
by a fake that hands out the other ranks' rows), each stage of each rank captured into a HIP graph and timed on replay.
    python tools/shard_estimate.py [--views 50 --keyframes 16 --ranks 1 2 4 8]
    python tools/shard_estimate.py --plans replicated broadcast
Projection for N GPUs: the collectives are synchronisation points, so the critical path is  max_r stage1 + [stage 2] + max_r stage3  with
  replicated: [stage 2] = max_r stage2 (every rank repeats the memory build), the two <= 30 MiB all-gathers counted as zero (latency-bound);
  broadcast : rank 0 builds (stage2a) while the others encode; the banks (2 x 12 layers x K*T x 768 x 2 B = 453 MB at K = 16) arrive
              bytes / --bcast-gbps later (default 100 GB/s effective: one xGMI link is 153 GB/s peak; an ASSUMPTION, the driver's SCALE run
              measures the real thing); rank r then runs stage2b from max(own stage2a end, bank arrival).
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variant', default='v2')
    ap.add_argument('--views', type=int, default=50)
    ap.add_argument('--keyframes', type=int, default=16)
    ap.add_argument('--ranks', type=int, nargs='+', default=[1, 2, 4, 8])
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--plans', nargs='+', default=['replicated'], choices=['replicated', 'broadcast'])
    ap.add_argument('--bcast-gbps', type=float, default=100.0)
    args = ap.parse_args()
    from panst3r_amd.panst3r import CONFIG_V1, CONFIG_V2, build_from_config
    from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings
    import panst3r_amd.scene as S
    dev = torch.device('cuda:0')
    V, K, H, W = args.views, args.keyframes, 384, 512
    model = build_from_config(CONFIG_V2 if args.variant == 'v2' else CONFIG_V1).eval()
    fill_module_(model, seed=1)
    names, emb = synth_class_embeddings(100)
    model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
    model.to(dev)
    imgs = {i: synth_image(i, H, W).to(dev) for i in range(V)}
    sends = []
    S._all_gather_rows = lambda t, counts, w, g: [s[:c] for s, c in zip(sends, counts)]
    out = {}
    base = None
    for plan in args.plans:
      for world in args.ranks:
        if plan == 'broadcast' and world == 1:
            continue
        _, order, owner = S.assign_views(V, K, world, plan=plan)
        runners = [S.SceneRunner(S.HipBackend(model), {order[i]: imgs[order[i]] for i in range(V) if owner[i] == r}, V, H, W, K, names,
                                 rank=r, world=world, plan=plan, stream_bank=False) for r in range(world)]
        split = plan == 'broadcast'
        for rn in runners:
            rn.split = split
        nseg = 4 if split else 3

        def seg(rn, k):
            return ([rn.stage1, rn.stage2a, rn.stage2b, rn.stage3] if split else [rn.stage1, rn.stage2, rn.stage3])[k]

        def exchange(k):
            if k == 0:
                sends[:] = [rn.enc_send for rn in runners]
                for rn in runners:
                    rn.gather1()
            elif split and k == 1:
                src = runners[0].b.bank_payload(runners[0].bank)
                for rn in runners[1:]:
                    for d, t in zip(rn.b.bank_payload(rn.bank), src):
                        d.copy_(t)
            elif k == nseg - 2:
                sends[:] = [rn.both_send for rn in runners]
                for rn in runners:
                    rn.gather2()

        with torch.no_grad():
            def scene(run):
                for k in range(nseg):
                    run(k)
                    exchange(k)
            scene(lambda k: [seg(rn, k)() for rn in runners])          # warm-up
            torch.cuda.synchronize()
            graphs = [[None] * nseg for _ in runners]
            pools = [torch.cuda.graph_pool_handle() for _ in runners]

            def capture(k):
                for r, rn in enumerate(runners):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pools[r], capture_error_mode='thread_local'):
                        seg(rn, k)()
                    graphs[r][k] = g
                    g.replay()
            scene(capture)
            torch.cuda.synchronize()
            ms = [[0.0] * nseg for _ in runners]

            def timed(k):
                for r in range(world):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); graphs[r][k].replay(); b.record(); b.synchronize()
                    ms[r][k] += a.elapsed_time(b) / args.reps
            for _ in range(args.reps):
                scene(timed)
        t1 = max(m[0] for m in ms)
        if split:
            bank_bytes = sum(t.numel() * t.element_size() for t in runners[0].b.bank_payload(runners[0].bank))
            bcast = bank_bytes / (args.bcast_gbps * 1e9) * 1e3
            arrive = t1 + ms[0][1] + bcast
            t2 = max(max(t1 + m[1], arrive) + m[2] for m in ms)
            extra = dict(bank_MB=round(bank_bytes / 1e6, 1), assumed_bcast_ms=round(bcast, 2), build_ms_rank0=round(ms[0][1], 2))
        else:
            t2 = t1 + max(m[1] for m in ms)
            extra = {}
        crit = t2 + max(m[-1] for m in ms)
        base = base or crit
        out['%s/%d' % (plan, world)] = dict(plan=plan, ranks=world, views_per_rank=[rn.n_local for rn in runners],
                                            stage_ms_per_rank=[[round(x, 2) for x in m] for m in ms], critical_path_ms=round(crit, 2),
                                            projected_frames_per_s=round(V / crit * 1e3, 1), projected_efficiency=round(base / crit / world, 3), **extra)
        print(plan, world, json.dumps(out['%s/%d' % (plan, world)]), flush=True)
        del runners, graphs, pools
        torch.cuda.empty_cache()
    print(json.dumps({'workload': '%s, %d views, %d keyframes, 384x512; ranks simulated on one GPU, stage graphs timed on replay' % (args.variant, V, K), 'ranks': out}))


if __name__ == '__main__':
    main()
