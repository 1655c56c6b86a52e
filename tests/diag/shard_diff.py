"""Where do the sharded (N lock-stepped runners on one GPU) and the unsharded scene first differ?  tiny model."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import tiny
import panst3r_amd.scene as S

DEV = 'cuda:0'
variant, V, K, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
h = tiny.build(tiny.hip_ns(), variant).to(DEV)
H, W = 64, 96
imgs = {i: im.to(DEV) for i, im in enumerate(tiny.images(V, H, W))}
with torch.no_grad():
    ref = h.scene_runner(imgs, V, H, W, tiny.NAMES, num_keyframes=K, use_graphs=False)
    ref.stage1(); ref.gather1(); ref.stage2(); ref.gather2(); ref.stage3()
    kf, order, owner = S.assign_views(V, K, world)
    runners = []
    for r in range(world):
        mine = {order[i]: imgs[order[i]] for i in range(V) if owner[i] == r}
        runners.append(S.SceneRunner(S.HipBackend(h), mine, V, H, W, K, tiny.NAMES, rank=r, world=world))
    sends = []
    S._all_gather_rows = lambda t, counts, w, g: [s[:c] for s, c in zip(sends, counts)]
    for rn in runners: rn.stage1()
    sends[:] = [rn.enc_send for rn in runners]
    for rn in runners: rn.gather1()
    print('gathered encoder tokens equal:', [torch.equal(rn.enc_kf, ref.enc_kf) for rn in runners])
    for rn in runners: rn.stage2()
    sends[:] = [rn.both_send for rn in runners]
    for rn in runners: rn.gather2()
    print('gathered fpn+fm equal:', [torch.equal(rn.both_kf, ref.both_kf) for rn in runners],
          [float((rn.both_kf.float() - ref.both_kf.float()).abs().max()) for rn in runners])
    d = ref.d
    print('  fpn part:', [torch.equal(rn.both_kf[:, :d], ref.both_kf[:, :d]) for rn in runners], ' fm part:', [torch.equal(rn.both_kf[:, d:], ref.both_kf[:, d:]) for rn in runners])
    # per keyframe
    T = 24
    for k in range(K):
        print('  keyframe %d (view %d, owner %d): fpn %s fm %s' % (k, kf[k], owner[k], torch.equal(runners[0].both_kf[k * T:(k + 1) * T, :d], ref.both_kf[k * T:(k + 1) * T, :d]),
                                                                  torch.equal(runners[0].both_kf[k * T:(k + 1) * T, d:], ref.both_kf[k * T:(k + 1) * T, d:])))
    # pointmaps of own views
    for rn in runners:
        for j, i in enumerate(rn.mine):
            g, r = rn.where[j]
            gr, rr = ref.where[ref.mine.index(i)]
            print('  rank %d view %d: pointmap %s cat-dec %s cat-dino %s' % (rn.rank, order[i], torch.equal(g.pointmaps[r], gr.pointmaps[rr]),
                  torch.equal(g.cat[r * T:(r + 1) * T, 128:256], gr.cat[rr * T:(rr + 1) * T, 128:256]), torch.equal(g.cat[r * T:(r + 1) * T, 256:], gr.cat[rr * T:(rr + 1) * T, 256:])))
