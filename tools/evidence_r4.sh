set -x
R=gpurun_out
bash tools/pmc_profile.sh r4 > $R/r4_pmc_profile.log 2>&1
cp $R/r4_pmc_summary.json profiles/r4_pmc_summary.json
python bench.py --cpu-baseline-c2 > $R/r4_bench_default.json 2> $R/r4_bench_default.err
python bench.py --amp bf16 --no-cpu-baseline --no-alt-dtype > $R/r4_bench_bf16.json 2>/dev/null
python bench.py --variant v1 --views 8 --keyframes 8 --no-cpu-baseline --no-alt-dtype > $R/r4_bench_config1_v1_8v8k.json 2>/dev/null
python bench.py --variant v1 --views 8 --keyframes 8 --amp bf16 --no-cpu-baseline --no-alt-dtype > $R/r4_bench_config1_v1_8v8k_bf16.json 2>/dev/null
python bench.py --views 16 --keyframes 16 --no-cpu-baseline --no-alt-dtype > $R/r4_bench_config2_v2_16v16k.json 2>/dev/null
python bench.py --views 16 --keyframes 16 --amp bf16 --no-cpu-baseline --no-alt-dtype > $R/r4_bench_config2_v2_16v16k_bf16.json 2>/dev/null
python bench.py --variant v1 --no-cpu-baseline --no-alt-dtype > $R/r4_bench_v1_50v16k.json 2>/dev/null
python bench.py --views 200 --keyframes 32 --steps 5 --no-cpu-baseline --no-alt-dtype > $R/r4_bench_c5_200v32k.json 2>/dev/null
PST_FORCE_DIST=1 python bench.py --steps 10 --no-cpu-baseline --no-alt-dtype --plan broadcast > $R/r4_bench_rccl_world1_broadcast.json 2>/dev/null
python tools/build_bench.py 16 > $R/r4_build_bench.txt 2>&1
python tools/build_bench.py 32 >> $R/r4_build_bench.txt 2>&1
python tools/shard_estimate.py --plans replicated broadcast > $R/r4_shard_estimate.txt 2>&1
python tools/shape_profile3.py > $R/r4_shape_profile.txt 2>&1
tail -2 $R/r4_build_bench.txt; cut -c1-200 $R/r4_bench_default.json
