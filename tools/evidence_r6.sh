# Round-6 evidence in one command on the GPU box (through gpurun): PMC passes + kernel stats of the default bench (tools/pmc_profile.sh), the bench lines of the
# configurations, the GEMM decomposition of the final kernels.  Every step is bounded by its own timeout.
set -x
R=gpurun_out
mkdir -p $R
timeout 1500 bash tools/pmc_profile.sh r6 > $R/r6_pmc_profile.log 2>&1
cp $R/r6_pmc_summary.json profiles/r6_pmc_summary.json 2>/dev/null
timeout 900 python bench.py --cpu-baseline-c2 > $R/r6_bench_default.json 2> $R/r6_bench_default.err
timeout 300 python bench.py --amp bf16 --no-cpu-baseline --no-alt-dtype > $R/r6_bench_bf16.json 2>/dev/null
timeout 300 python bench.py --variant v1 --views 8 --keyframes 8 --amp bf16 --no-cpu-baseline --no-alt-dtype > $R/r6_bench_config1_v1_8v8k_bf16.json 2>/dev/null
timeout 300 python bench.py --views 16 --keyframes 16 --amp bf16 --no-cpu-baseline --no-alt-dtype > $R/r6_bench_config2_v2_16v16k_bf16.json 2>/dev/null
timeout 400 python bench.py --views 200 --keyframes 32 --steps 5 --no-cpu-baseline --no-alt-dtype > $R/r6_bench_c5_200v32k.json 2>/dev/null
PST_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-alt-dtype --plan broadcast > $R/r6_bench_rccl_world1_broadcast.json 2>/dev/null
PST_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --no-cpu-baseline --no-alt-dtype --plan broadcast --stream-bank > $R/r6_bench_rccl_world1_broadcast_streamed.json 2>/dev/null
timeout 200 python tools/build_bench.py 16 > $R/r6_build_bench.txt 2>&1
timeout 200 python tools/build_bench.py 32 >> $R/r6_build_bench.txt 2>&1
timeout 400 python tools/gemm_k1024.py > $R/r6_gemm_k1024_final.txt 2>&1
tail -2 $R/r6_build_bench.txt; cut -c1-300 $R/r6_bench_default.json
