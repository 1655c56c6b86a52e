#!/bin/bash
# What bounds the fp32 residual-stream epilogue of the persistent 256 x 256 GEMM (31 us per tile, profiles/r6_gemm_k1024.txt)?  (GPU box)
#   1. ablation builds (tools/pp_ablate.sh build base:-DPST_ABL=0 epi:-DPST_ABL=11 epi_nold:-DPST_ABL=75 epi_nost:-DPST_ABL=139 epi_nocp:-DPST_ABL=267
#      epi_none:-DPST_ABL=459 nold:-DPST_ABL=64 nost:-DPST_ABL=128 nocp:-DPST_ABL=256): the kernel WITHOUT its K loop (epi*) = the epilogue alone,
#      without its residual loads / fp32 stores / 16-bit copy; and the whole kernel without each of the three
#   2. the same on 256 / 128 / 64 workgroups (PST_TUNE_CUS through tools/pp_time.py's 6th argument): a time per round that does not fall with fewer
#      workgroups in flight is a per-CU (latency / issue) bound, one that falls is the memory system's
ROOT=$(cd $(dirname $0)/.. && pwd); L=$ROOT/panst3r_amd/lib
for CUS in 256 128 64; do
  R=$((16 * 256 / CUS))        # keep 16 tiles per workgroup... M = rounds * 4096 rows at N = 1024 (4 column tiles)
  for N in base epi epi_nold epi_nost epi_nocp epi_none nold nost nocp; do
    printf "cus %3d %-9s " $CUS $N; PST_LIB=$L/abl/lib_$N.so python $ROOT/tools/pp_time.py $((CUS * 16 * 64)) 1024 1024 res 1 $CUS 2>/dev/null | grep us
  done
done
