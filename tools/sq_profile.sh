#!/bin/bash
# SQ stall anatomy of the MFMA kernels (GPU box): two --pmc passes (8 SQ slots each, kernel-trace only) over a short driver
# script, then per-kernel sums.   tools/sq_profile.sh <tag> <python script + args...>
#   -> gpurun_out/<tag>_sq.md : wave-cycles split into ACTIVE / WAIT_INST (issue stall) / WAIT_ANY (waitcnt, barrier), VALU / LDS /
#      MFMA activity, MFMA-VALU co-execution, LDS bank conflicts (SQ_*_CYCLES are quad-cycles, MFMA_BUSY is cycles: MI355X_MICROARCH.md)
set -e
TAG=$1; shift
ROOT=$(pwd)
SCRIPT=$ROOT/$1; shift
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \
  --output-format csv -d $OUT/p1 -o p1 -- python $SCRIPT "$@" > $OUT/p1.log 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES \
  --output-format csv -d $OUT/p2 -o p2 -- python $SCRIPT "$@" > $OUT/p2.log 2>&1 || true
cd $ROOT
python - $OUT $ROOT/gpurun_out/${TAG}_sq.md <<'PY'
import csv, glob, sys, collections, re
out, md = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for p in ('p1', 'p2'):
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, p), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(.*', '', r['Kernel_Name'])[:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            if p == 'p1' and (r['Dispatch_Id'], r['Counter_Name']) not in seen and r['Counter_Name'] == 'SQ_WAVE_CYCLES':
                n[k] += 1
            seen.add((r['Dispatch_Id'], r['Counter_Name']))
rows = sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:12]
with open(md, 'w') as fh:
    fh.write('| kernel | launches | active % | wait_inst % | wait_any % | VALU act % | LDS act % | MFMA busy % of busy | MFMA/VALU coexec % of MFMA busy | VALU insts / MFMA | LDS bank conflict % |\n|---|---|---|---|---|---|---|---|---|---|---|\n')
    for k, c in rows:
        wc = c.get('SQ_WAVE_CYCLES', 0) or 1
        busy = c.get('SQ_BUSY_CYCLES', 0) or 1
        mf = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
        fh.write('| `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %.2f |\n' % (
            k, n[k], 100 * c.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * c.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * c.get('SQ_WAIT_ANY', 0) / wc,
            100 * c.get('SQ_ACTIVE_INST_VALU', 0) / wc, 100 * c.get('SQ_ACTIVE_INST_LDS', 0) / wc, 100 * mf / (busy * 4.0) if busy else 0,
            100 * c.get('SQ_VALU_MFMA_COEXEC_CYCLES', 0) / (mf or 1), c.get('SQ_INSTS_VALU', 0) / (c.get('SQ_INSTS_MFMA', 0) or 1),
            100 * c.get('SQ_LDS_BANK_CONFLICT', 0) / (c.get('SQ_LDS_IDX_ACTIVE', 0) or 1)))
    fh.write('\nraw sums per kernel:\n')
    for k, c in rows:
        fh.write('- `%s`: %s\n' % (k, ', '.join('%s=%.4g' % kv for kv in sorted(c.items()))))
print(open(md).read())
PY
find $OUT -name '*_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*.db' -delete
