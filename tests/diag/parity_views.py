"""Per-view parity of a full-size scene (HIP vs fp32 oracle): which views carry the error?  python tests/diag/parity_views.py v2 5 3 fp16"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import bench
    from panst3r_amd.panst3r import CONFIG_V1, CONFIG_V2, build_from_config
    from panst3r_amd.schedule import select_keyframes
    from panst3r_amd.synthetic import fill_module_, synth_class_embeddings
    variant, V, K, amp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dev = torch.device('cuda:0')
    model = build_from_config(CONFIG_V2 if variant == 'v2' else CONFIG_V1).eval()
    fill_module_(model, seed=1)
    names, emb = synth_class_embeddings(100)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
    model.to(dev)
    _, ref, imgs, ts = bench.cpu_baseline(variant, 384, 512, state, names, emb, bench.usable_cores(), V=V, K=K)
    pm_o, pan_o = ref
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    out = {'keyframes': select_keyframes(V, K)}
    with torch.no_grad():
        pm_h, pan_h = model.forward_inference_multi_ar([i.to(dev) for i in imgs], ts, names, num_keyframes=K, amp=amp)
        out['pointmaps'] = [rel(a, b) for a, b in zip(pm_h, pm_o)]
        out['masks'] = [rel(a, b) for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks'])]
        out['sign'] = [float(((a.cpu() > 0) == (b > 0)).float().mean()) for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks'])]
        out['queries'] = rel(pan_h['out_queries'], pan_o['out_queries'])
        out['logits_abs'] = float((pan_h['pred_logits'].cpu() - pan_o['pred_logits']).abs().max())
        # per-query error of out_queries: a few bad queries (attention-mask bit flips) or uniformly spread?
        dq = (pan_h['out_queries'].cpu().double() - pan_o['out_queries'].double()).reshape(200, -1).norm(dim=-1) / pan_o['out_queries'].double().reshape(200, -1).norm(dim=-1)
        out['query_err_sorted_top'] = [round(float(x), 4) for x in dq.sort(descending=True).values[:12]]
        out['query_err_median'] = float(dq.median())
        # masks with the ORACLE's queries (heads-only path): isolates the mask-feature error from the query error
        hs = model.panoptic_decoder.mask_transformer
        from panst3r_amd.model.common import precision
        with precision(amp):
            pass
    print(json.dumps(out))


main()
