"""Full-size parity of the HIP path vs the fp32 oracle in BOTH 16-bit formats (amp='fp16' / 'bf16'), tolerances of SURVEY 8(d).
    python tests/diag/parity_amp.py [variant] [V] [K] > gpurun_out/parity_amp.json
The oracle is the checker (this file lives under tests/)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import bench
    from panst3r_amd.panst3r import CONFIG_V1, CONFIG_V2, build_from_config
    from panst3r_amd.synthetic import fill_module_, synth_class_embeddings, synth_image
    from oracle.pipeline import build
    variant = sys.argv[1] if len(sys.argv) > 1 else 'v2'
    V = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    sharp = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    H, W = 384, 512
    dev = torch.device('cuda:0')
    model = build_from_config(CONFIG_V2 if variant == 'v2' else CONFIG_V1).eval()
    fill_module_(model, seed=1, sharp=sharp)
    names, emb = synth_class_embeddings(100)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
    torch.set_num_threads(bench.usable_cores())
    orc = build(variant)
    orc.load_state_dict(state, strict=True)
    orc.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
    imgs = [synth_image(i, H, W) for i in range(V)]
    ts = torch.tensor([[H, W]] * V)
    t0 = time.perf_counter()
    with torch.no_grad():
        pm_o, pan_o = orc.forward_inference_multi_ar(imgs, ts, names, num_keyframes=K)
    t_cpu = time.perf_counter() - t0
    model.to(dev)
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    out = {'variant': variant, 'views': V, 'keyframes': K, 'sharp': sharp, 'oracle_seconds': round(t_cpu, 1)}
    for amp in ('fp16', 'bf16'):
        with torch.no_grad():
            pm_h, pan_h = model.forward_inference_multi_ar([i.to(dev) for i in imgs], ts, names, num_keyframes=K, amp=amp)
        torch.cuda.synchronize()
        mk = [(a.cpu(), b) for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks'])]
        out[amp] = {'pointmaps_rel_l2': max(rel(a, b) for a, b in zip(pm_h, pm_o)),
                    'mask_logits_rel_l2': max(rel(a, b) for a, b in mk),
                    'mask_sign_agreement': min(float(((a > 0) == (b > 0)).float().mean()) for a, b in mk),
                    'class_logits_max_abs': float((pan_h['pred_logits'].cpu() - pan_o['pred_logits']).abs().max()),
                    'out_queries_rel_l2': rel(pan_h['out_queries'], pan_o['out_queries']),
                    'max_abs_pointmap': float(max(a.abs().max() for a in pm_h)), 'max_abs_mask': float(max(a.abs().max() for a, _ in mk))}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
