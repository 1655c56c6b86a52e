#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc passes of bench.py into per-kernel MFMA utilisation and HBM traffic (run on the GPU box).

Usage: python tools/pmc_summary.py <dir with pass sub-dirs a/ b/ c/> <out.md> <out.json>
  pass a: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES      pass b: FETCH_SIZE      pass c: WRITE_SIZE
Corrections per MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB (x1024 -> bytes); on gfx950 FETCH_SIZE
reports exactly half of the bytes of wide (16 B/lane) coalesced streaming reads -> the read side is doubled (all our
streaming reads are 16 B/lane LDS-DMA or dwordx4).  GRBM_GUI_ACTIVE and SQ counters are summed over the 8 XCDs.
"""
import csv, collections, json, os, re, sys

root, out_md, out_json = sys.argv[1:4]
SIMDS = 256 * 4


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*', '', name)
    return name.replace('pst::', '')


def load(passdir):
    cnt = collections.defaultdict(lambda: collections.defaultdict(float))     # dispatch -> counter -> value
    kname = {}
    f = [x for x in os.listdir(passdir) if x.endswith('counter_collection.csv')][0]
    for r in csv.DictReader(open(os.path.join(passdir, f))):
        cnt[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
        kname[r['Dispatch_Id']] = short(r['Kernel_Name'])
    dur = {}
    f = [x for x in os.listdir(passdir) if x.endswith('kernel_trace.csv')][0]
    for r in csv.DictReader(open(os.path.join(passdir, f))):
        dur[r['Dispatch_Id']] = float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    return cnt, kname, dur


agg = collections.defaultdict(lambda: collections.defaultdict(float))
for p in ('a', 'b', 'c'):
    cnt, kname, dur = load(os.path.join(root, p))
    for d, cs in cnt.items():
        k = kname[d]
        if not (k.startswith('gemm') or k.startswith('attn') or k.startswith('layernorm') or 'kernel' in k):
            continue
        agg[k]['calls_' + p] += 1
        agg[k]['ns_' + p] += dur.get(d, 0.0)
        for c, v in cs.items():
            agg[k][c] += v

rows = []
for k, a in agg.items():
    if a['calls_a'] == 0:
        continue
    mfma_util = a['SQ_VALU_MFMA_BUSY_CYCLES'] / max(a['GRBM_GUI_ACTIVE'] / 8 * SIMDS, 1)
    rd = 2.0 * a['FETCH_SIZE'] * 1024 / max(a['calls_b'], 1)
    wr = a['WRITE_SIZE'] * 1024 / max(a['calls_c'], 1)
    t_b = a['ns_b'] / max(a['calls_b'], 1) * 1e-9
    t_c = a['ns_c'] / max(a['calls_c'], 1) * 1e-9
    gbs = (rd / max(t_b, 1e-12) + wr / max(t_c, 1e-12)) / 1e9
    rows.append((a['ns_a'], k, int(a['calls_a']), a['ns_a'] / a['calls_a'] / 1e3, mfma_util, rd / 1e6, wr / 1e6, gbs))
rows.sort(reverse=True)
with open(out_md, 'w') as f:
    f.write('| kernel | launches | avg us (under PMC) | MFMA busy / (active cycles x 1024 SIMDs) | HBM read MB/launch (FETCH_SIZE x2) | HBM write MB/launch | HBM GB/s (of 8000 spec / 6300 achievable) |\n|---|---|---|---|---|---|---|\n')
    for _, k, n, us, mu, rd, wr, gbs in rows[:24]:
        f.write('| `%s` | %d | %.1f | %.1f %% | %.2f | %.2f | %.0f |\n' % (k, n, us, 100 * mu, rd, wr, gbs))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from panst3r_amd.build import source_hash
summary = {'_source_hash': source_hash(), '_command': 'tools/pmc_profile.sh (rocprofv3 --pmc, 3 passes of bench.py --no-cpu-baseline --steps 1 --eager --no-kernel-timing)'}
summary.update({k: {'launches': n, 'mfma_util': round(mu, 4), 'hbm_read_bytes_per_launch': rd * 1e6, 'hbm_write_bytes_per_launch': wr * 1e6,
               'hbm_gbps': round(gbs, 1)} for _, k, n, us, mu, rd, wr, gbs in rows})
json.dump(summary, open(out_json, 'w'), indent=1)
print(open(out_md).read())
