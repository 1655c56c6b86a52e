#!/bin/bash
# PMC comparison of the two K loops on one shape: tools/pp_pmc.sh M N K kind  -> gpurun_out/pp_pmc.txt
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pp_pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PP in 0 1; do
  i=0
  for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA"; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pp${PP}_$i -o x -- python $ROOT/tools/pp_probe.py $1 $2 $3 $4 $PP 6 > /dev/null 2> $OUT/pp${PP}_$i.err
    i=$((i+1))
  done
done
cd $ROOT
python - <<'PY' > gpurun_out/pp_pmc.txt
import csv, glob, collections
for pp in (0, 1):
    agg = collections.defaultdict(list)
    for f in glob.glob('gpurun_out/pp_pmc/pp%d_*/*counter_collection.csv' % pp):
        for r in csv.DictReader(open(f)):
            if 'gemm256p' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print('pp =', pp)
    for k, v in sorted(agg.items()):
        print('  %-34s launches %3d  mean %.4g' % (k, len(v), sum(v) / len(v)))
PY
cat gpurun_out/pp_pmc.txt
