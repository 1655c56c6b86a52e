#!/usr/bin/env python
"""Where does the 16-bit error enter?  (PST_AMP=bf16 selects the bf16 operand format; f16 is the default.)  Full-size v2 model, 2 views / 2 keyframes at 384x512 (the bench's cpu_baseline sample):
the HIP path and the fp32 CPU oracle (same weights) are compared stage by stage.  Diagnostic; the oracle is only the checker."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from panst3r_amd.panst3r import CONFIG_V2, build_from_config
from panst3r_amd.synthetic import fill_module_, synth_image, synth_class_embeddings
from panst3r_amd.model.common import Layout, PREC, amp_dtype, adt
from oracle.pipeline import build as build_oracle
from oracle.must3r import build_memory, mem_batches_for
import bench

H, W, V = 384, 512, 2
dev = torch.device('cuda:0')
PREC.dtype = amp_dtype(os.environ.get('PST_AMP', 'fp16'))      # process-wide operand format for the module-level calls below
BF = adt()
model = build_from_config(CONFIG_V2).eval()
fill_module_(model, seed=1)
names, emb = synth_class_embeddings(100)
state = {k: v.clone() for k, v in model.state_dict().items()}
model.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
model.to(dev)
torch.set_num_threads(bench.usable_cores())
o = build_oracle('v2')
o.load_state_dict(state, strict=True)
o.panoptic_decoder.text_encoder.class_embeddings = {n: e for n, e in zip(names, emb)}
imgs = [synth_image(i, H, W) for i in range(V)]
ts = torch.tensor([[H, W]] * V)
h, w = H // 16, W // 16
T = h * w
rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
out = {}
with torch.no_grad():
    # ---- oracle stages
    st = torch.stack(imgs)
    xe_o, pos = o.must3r_encoder(st, ts)                                  # [V,T,1024]
    xd_o = o.dino_encoder(st, ts)                                         # [V,T,1024]
    shapes = [[H, W]] * V
    mem = build_memory(o.must3r_decoder, [xe_o[i] for i in range(V)], [pos[i] for i in range(V)], shapes, mem_batches_for(V))
    ys, pms = [], []
    for i in range(V):
        _, pm, y = o.must3r_decoder.forward_list([xe_o[i][None]], [pos[i][None]], [shapes[i]], mem, render=True)
        ys.append(y[0][0]); pms.append(pm[0][0])
    y_o = torch.stack(ys)
    cat_o = torch.cat([xe_o, y_o, xd_o], -1)                             # [V,T,2816]
    pd = o.panoptic_decoder
    mix_o = pd.input_mixer(cat_o, pos)
    fpn_o, mf_o = pd.features(cat_o[None], st[None], pos[None], ts[None], max_bs=1)
    # ---- HIP stages on the same inputs
    img_d = st.to(dev)
    cat_h = torch.empty(V * T, model._cat_width(), dtype=BF, device=dev)
    model.encode_views(img_d, cat_h)
    De, Dd = 1024, 768
    out['encoder tokens'] = rel(cat_h[:, :De].float().view(V, T, De), xe_o)
    out['dinov2 tokens'] = rel(cat_h[:, De + Dd:].float().view(V, T, -1), xd_o)
    bank = model.build_memory(cat_h[:, :De].contiguous(), V, h, w)
    pm_h = model.render_views(cat_h, V, h, w, bank)
    out['decoder features'] = rel(cat_h[:, De:De + Dd].float().view(V, T, Dd), y_o)
    out['pointmaps'] = rel(pm_h, torch.stack(pms))
    # the panoptic half fed with the ORACLE's tokens (isolates its own error) and with the HIP tokens (accumulated error)
    pdh = model.panoptic_decoder
    for tag, cat_in in (('own error (oracle tokens in)', cat_o.reshape(V * T, -1).to(BF).to(dev)), ('accumulated (HIP tokens in)', cat_h)):
        mix = torch.zeros(V * T, pdh.upscaler.lr_width(), dtype=BF, device=dev)
        pdh.input_mixer.mix_tokens(cat_in, V, h, w, mix)
        out['input mixer, ' + tag] = rel(mix[:, :768].float().view(V, T, 768), mix_o)
        fpn_h, mf_h = pdh.features_tokens(cat_in, img_d, V, h, w)
        out['fpn tokens, ' + tag] = rel(fpn_h.float().view(V, h, w, -1).permute(0, 3, 1, 2), fpn_o[0])
        out['mask features, ' + tag] = rel(mf_h.float().permute(0, 3, 1, 2), mf_o[0])
# ---- inside the LoftUp stage: the image-only guidance branch (per-view min-max scaling like the demo's max_bs = 1) and the query decoder
with torch.no_grad():
    import torch.nn.functional as F
    up_o, up_h = pd.upscaler, pdh.upscaler
    g_o = torch.cat([up_o.first_conv(up_o.fourier_feat(F.interpolate(st[i:i + 1], scale_factor=0.5, mode='bilinear', align_corners=False)))
                     for i in range(V)])                                                  # [V,C,H2,W2]
    g_h = up_h.guidance_tokens(img_d, h, w)                                                # [V*P, C]
    out['loftup guidance branch (image only)'] = rel(g_h.float().view(V, H // 2, W // 2, -1).permute(0, 3, 1, 2), g_o)
    # query decoder fed with the ORACLE's keyframe features: its own error, vs fed with the HIP features
    mt_o, mt_h = pd.mask_transformer, pdh.mask_transformer
    cls_o = pd.text_encoder(names)
    ref = mt_o([[fpn_o[0][None][:, i] for i in range(V)]] if False else [[fpn_o[:, i:i + 1][0][None] for i in range(V)]],
               [mf_o[:, i:i + 1][0][None] for i in range(V)], [ts[i:i + 1][None] for i in range(V)], cls_o, multi_ar=True)
    q_o = ref['out_queries'].reshape(-1, ref['out_queries'].shape[-1])
    cls_h = pdh.text_encoder.normalized_bf16(names, dev)
    fpn_tok_o = fpn_o[0].flatten(2).transpose(1, 2).reshape(V * T, -1).to(BF).to(dev)
    mf_tok_o = mf_o[0].permute(0, 2, 3, 1).contiguous().to(BF).to(dev)
    q_h_own, _ = mt_h.decode_tokens(fpn_tok_o, mt_h.attn_feats(mf_tok_o), [(h, w)] * V, cls_h)
    out['query decoder, own error (oracle features in)'] = rel(q_h_own, q_o)
for k, v in out.items():
    print('%-52s rel-L2 %.2e' % (k, v))
print(json.dumps(out))
