"""NEGATIVE CONTROLS: the parity comparisons of tests/test_hip_model.py must FAIL when the keyframe memory bank is built wrong.

The CroCo / MUSt3R half of the path is checked against a restatement (parity unpinned), on N(0, 0.02^2) weights whose
softmaxes are close to uniform - a bound 20x the measured error would pass a bank with a keyframe missing.  Here the HIP path's bank is corrupted on
purpose in the ways a memory bug would (reference engine/must3r.py:28-69 builds it, :76-80 reads it):
    drop      the last keyframe's entries are not in the bank when the views are rendered
    mismatch  the K rows of two keyframes are exchanged while their V rows stay (keys paired with the wrong values)
    stale     one keyframe's entries are a copy of another's (an append that wrote to the wrong slot)
    nofb      the feedback term is left out of every memory entry (entry_l = h_l instead of h_l + feedback(out), the MUSt3R memory rule)
and the SAME comparison with the SAME bounds (test_hip_model.BOUNDS) is asserted to fail: at the level of the memory entries (always) and at the
level of the scene's outputs.  Every number goes to gpurun_out/parity_negative.jsonl (-> profiles/r4_parity_negative.jsonl)."""
import json
import os

import pytest
import torch

from conftest import rel_l2
import tiny
from test_hip_model import BOUNDS

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CORRUPTIONS = ('drop', 'mismatch', 'stale', 'nofb')


def _record(payload):
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'parity_negative.jsonl'), 'a') as f:
            f.write(json.dumps(payload) + '\n')
    except OSError:
        pass


class no_feedback:
    """the 'nofb' corruption: the decoder's feedback switch is turned off for the duration (entry_l = h_l); the product code carries no test hook.
    (The clean run comes first in every test, so the cached weight pack already holds the feedback MLP.)"""

    def __init__(self, dec, on):
        self.dec, self.on = dec, on

    def __enter__(self):
        self.prev = self.dec.feedback_type
        if self.on:
            self.dec.feedback_type = None

    def __exit__(self, *a):
        self.dec.feedback_type = self.prev


def corrupt_bank(bank, how, T):
    """apply one of the bank corruptions in place (bank: panst3r_amd.model.must3r.MemoryBank of >= 3 keyframes of T tokens)"""
    nk = bank.n // T
    assert nk >= 3
    if how == 'drop':
        bank.n -= T
        bank.labels, bank.nimgs = bank.labels[:-1], bank.nimgs - 1
    elif how == 'mismatch':
        a = bank.K_all[:, 0:T].clone()
        bank.K_all[:, 0:T] = bank.K_all[:, T:2 * T]
        bank.K_all[:, T:2 * T] = a
    elif how == 'stale':
        bank.K_all[:, T:2 * T] = bank.K_all[:, 0:T]
        bank.Vt_all[:, :, T:2 * T] = bank.Vt_all[:, :, 0:T]
    return bank


@pytest.fixture(scope='module', params=[('v1', 'fp16'), ('v2', 'fp16'), ('v1', 'bf16')], ids=lambda p: '%s-%s' % p)
def pair(request):
    from panst3r_amd.model.common import precision
    variant, amp = request.param
    o = tiny.build(tiny.OracleNS, variant)
    h = tiny.build(tiny.hip_ns(), variant).to(DEV)
    h.amp = amp
    with precision(amp):
        yield variant, o, h


@pytest.mark.parametrize('how', CORRUPTIONS)
def test_wrong_memory_bank_fails_the_entry_comparison(pair, how):
    """memory-chain comparison of test_memory_chain_error_vs_keyframe_index (K = 6): per-keyframe projected entries and the render of every
    keyframe against the final bank, HIP vs oracle.  Clean: inside the bounds.  Corrupted: OUTSIDE them."""
    variant, o, h = pair
    K, H, W = 6, 64, 96
    T = (H // 16) * (W // 16)
    img = torch.stack(tiny.images(K, H, W))
    ts = torch.tensor([[H, W]] * K)

    def run(corruption):
        with no_feedback(h.must3r_decoder, corruption == 'nofb'):
            with torch.no_grad():
                x, pos = o.must3r_encoder(img, ts)
                x, pos, tsb = x[None], pos[None], ts[None]
                mem_o = mem_h = None
                for a, b in [(0, 2)] + [(i, i + 1) for i in range(2, K)]:
                    mem_o, _, _ = o.must3r_decoder(x[:, a:b], pos[:, a:b], tsb[:, a:b], mem_o, render=False, return_feats=True)
                    mem_h, _, _ = h.must3r_decoder(x[:, a:b].to(DEV), pos[:, a:b].to(DEV), tsb[:, a:b], mem_h, render=False, return_feats=True)
                bank = mem_h[0]
                if corruption in ('drop', 'mismatch', 'stale'):
                    corrupt_bank(bank, corruption, T)
                entry = []
                for i in range(bank.n // T):
                    worst = 0.0
                    for l, blk in enumerate(o.must3r_decoder.blocks_dec):
                        ref = blk.cross_attn.projk(blk.norm_y(mem_o[0][l][0, i * T:(i + 1) * T]))
                        worst = max(worst, rel_l2(bank.K[l][i * T:(i + 1) * T].float().cpu(), ref))
                    entry.append(worst)
                _, pm_o, _ = o.must3r_decoder(x, pos, tsb, mem_o, render=True, return_feats=True)
                _, pm_h, _ = h.must3r_decoder(x.to(DEV), pos.to(DEV), tsb, mem_h, render=True, return_feats=True)
            return max(entry), max(rel_l2(pm_h[0, i].cpu(), pm_o[0, i]) for i in range(K)), bank.n // T
    b = BOUNDS[h.amp]
    e0, r0, n0 = run(None)
    e1, r1, n1 = run(how)
    _record(dict(test='bank entries / render', variant=variant, amp=h.amp, corruption=how, clean_entry=e0, clean_render=r0, bad_entry=e1, bad_render=r1,
                 bound_tok=b['tok'], bound_pm=b['pm']))
    assert e0 <= b['tok'] and r0 <= b['pm'] and n0 == K, (e0, r0)
    if how == 'drop':
        assert n1 == K - 1                                  # the entry count itself is the first thing the comparison sees
    else:
        assert e1 > b['tok'], (how, e1, b['tok'])           # a wrong entry is outside the bound the right ones meet
    assert r1 > b['pm'], (how, r1, b['pm'])                 # ... and so are the pointmaps rendered from the wrong bank


@pytest.mark.parametrize('how', CORRUPTIONS)
def test_wrong_memory_bank_fails_the_scene_comparison(pair, monkeypatch, how):
    """the end-to-end comparison of test_scene_end_to_end (V = 7 views, K = 4 keyframes): with the bank corrupted between the memory build and
    the render the pointmap bound must be violated; the clean scene passes the same bound."""
    variant, o, h = pair
    V, K, H, W = 7, 4, 64, 96
    T = (H // 16) * (W // 16)
    imgs = tiny.images(V, H, W)
    ts = torch.tensor([[H, W]] * V)
    pm_o, pan_o = o.forward_inference_multi_ar(imgs, ts, tiny.NAMES, num_keyframes=K)

    def run(corruption):
        build = type(h).build_memory
        if corruption in ('drop', 'mismatch', 'stale'):
            monkeypatch.setattr(type(h), 'build_memory', lambda self, *a, **k: corrupt_bank(build(self, *a, **k), corruption, T))
        try:
            with no_feedback(h.must3r_decoder, corruption == 'nofb'):
                pm_h, pan_h = h.forward_inference_multi_ar([i.to(DEV) for i in imgs], ts, tiny.NAMES, num_keyframes=K, amp=h.amp)
        finally:
            monkeypatch.setattr(type(h), 'build_memory', build)
        pm = max(rel_l2(a.cpu(), b) for a, b in zip(pm_h, pm_o))
        mk = max(rel_l2(a.cpu(), b) for a, b in zip(pan_h['pred_masks'], pan_o['pred_masks']))
        return pm, mk
    b = BOUNDS[h.amp]
    pm0, mk0 = run(None)
    pm1, mk1 = run(how)
    _record(dict(test='scene', variant=variant, amp=h.amp, corruption=how, clean_pointmaps=pm0, clean_masks=mk0, bad_pointmaps=pm1, bad_masks=mk1,
                 bound_pm=b['pm'], bound_mask_view=b['mask_view']))
    assert pm0 <= b['pm'] and mk0 <= b['mask_view'], (pm0, mk0)
    assert pm1 > b['pm'], (how, pm1, b['pm'])
