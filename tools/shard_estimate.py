#!/usr/bin/env python
"""Per-rank critical path of the view-sharded scene, measured on ONE GPU: N SceneRunners stepped in lock-step (the all-gathers replaced
by a fake that hands out the other ranks' rows), each stage of each rank captured into a HIP graph and timed on replay.
    python tools/shard_estimate.py [--views 50 --keyframes 16 --ranks 1 2 4 8]
Projection for N GPUs = max over ranks of (stage1 + stage2 + stage3) + nothing for the two <= 30 MiB all-gathers (latency-bound, tens of
microseconds over xGMI); the driver's SCALE run measures the real thing."""
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
    for world in args.ranks:
        _, order, owner = S.assign_views(V, K, world)
        runners = [S.SceneRunner(S.HipBackend(model), {order[i]: imgs[order[i]] for i in range(V) if owner[i] == r}, V, H, W, K, names,
                                 rank=r, world=world) for r in range(world)]

        def step(fn):
            for rn in runners:
                fn(rn)
        with torch.no_grad():
            def scene(stage):
                stage(0); sends[:] = [rn.enc_send for rn in runners]; step(lambda rn: rn.gather1())
                stage(1); sends[:] = [rn.both_send for rn in runners]; step(lambda rn: rn.gather2())
                stage(2)
            scene(lambda k: step(lambda rn: (rn.stage1, rn.stage2, rn.stage3)[k]()))          # warm-up
            torch.cuda.synchronize()
            graphs = [[None] * 3 for _ in runners]
            pools = [torch.cuda.graph_pool_handle() for _ in runners]

            def capture(k):
                for r, rn in enumerate(runners):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=pools[r], capture_error_mode='thread_local'):
                        (rn.stage1, rn.stage2, rn.stage3)[k]()
                    graphs[r][k] = g
                    g.replay()
            scene(capture)
            torch.cuda.synchronize()
            ms = [[0.0] * 3 for _ in runners]

            def timed(k):
                for r in range(world):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); graphs[r][k].replay(); b.record(); b.synchronize()
                    ms[r][k] += a.elapsed_time(b) / args.reps
            for _ in range(args.reps):
                scene(timed)
        per_rank = [sum(m) for m in ms]
        crit = max(per_rank)
        base = base or crit
        out[world] = dict(views_per_rank=[rn.n_local for rn in runners], stage_ms_of_slowest_rank=[round(x, 2) for x in ms[per_rank.index(crit)]],
                          critical_path_ms=round(crit, 2), projected_frames_per_s=round(V / crit * 1e3, 1),
                          projected_efficiency=round(base / crit / world, 3))
        print(world, json.dumps(out[world]), flush=True)
        del runners, graphs, pools
        torch.cuda.empty_cache()
    print(json.dumps({'workload': '%s, %d views, %d keyframes, 384x512; ranks simulated on one GPU, stage graphs timed on replay' % (args.variant, V, K), 'ranks': out}))


if __name__ == '__main__':
    main()
