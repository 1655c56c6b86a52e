#!/bin/bash
# One command for the profile evidence of a round (run on the GPU box from the repo root, e.g. through gpurun):
#   tools/pmc_profile.sh r2        -> gpurun_out/prof_r2/{stats, a, b, c}, gpurun_out/r2_pmc_summary.{md,json}, gpurun_out/r2_bench_kernel_stats.csv
# Pass 0: rocprofv3 --kernel-trace --stats of the default bench (per-kernel average durations the bench's HIP events must agree with).
# Passes a/b/c: --pmc counters in their OWN runs (kernel-trace only; gpurun refuses pmc + sys-trace), one eager scene each:
#   a: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES      b: FETCH_SIZE      c: WRITE_SIZE   (TCC slots: they cannot share a pass)
# tools/pmc_summary.py applies the gfx950 corrections of MI355X_MICROARCH.md (KiB units, FETCH_SIZE x2) and stamps the kernel-source hash.
set -e
R=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $ROOT/bench.py --no-cpu-baseline --steps 10 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || true
for P in a b c; do
  case $P in a) C="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES";; b) C="FETCH_SIZE";; c) C="WRITE_SIZE";; esac
  timeout -k 5 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/$P -o $P -- python $ROOT/bench.py --no-cpu-baseline --steps 1 --eager --no-kernel-timing --overlap off > /dev/null 2> $OUT/$P.err || true
  # flatten: pmc_summary.py expects the csv files directly under the pass directory
  find $OUT/$P -name '*counter_collection.csv' -exec cp {} $OUT/$P/ \; ; find $OUT/$P -name '*kernel_trace.csv' -exec cp {} $OUT/$P/ \;
done
cd $ROOT
python tools/pmc_summary.py $OUT gpurun_out/${R}_pmc_summary.md gpurun_out/${R}_pmc_summary.json
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} gpurun_out/${R}_bench_kernel_stats.csv \;
cp $OUT/bench_under_rocprof.json gpurun_out/${R}_bench_under_rocprof.json
# keep the merge-back small: drop the raw traces (hundreds of MB)
find $OUT -name '*_trace.csv' -delete; find $OUT -name '*counter_collection.csv' -delete; find $OUT -name '*.db' -delete
