#!/bin/bash
# Measurement builds of gemm256.hip (ablations / schedule experiments of the ping-pong K loop).  Results of ablation builds are garbage, only time matters.
#   tools/pp_ablate.sh build name1:-DPST_ABL=1 name2:-DPST_PPV=1 ...     (here, no GPU needed; -> panst3r_amd/lib/abl/lib_<name>.so)
#   tools/pp_ablate.sh run "name1 name2" M N K kind                        (GPU box)
# PST_ABL bits: 1 no DMA, 2 no LDS reads, 4 no barriers, 8 no MFMAs in the K loop.
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); L=$ROOT/panst3r_amd/lib
if [ "$1" = build ]; then
  shift; mkdir -p $L/abl
  for V in "$@"; do
    N=${V%%:*}; F=${V#*:}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function ${F//,/ } -c $ROOT/panst3r_amd/csrc/gemm256.hip -o $L/abl/gemm256_$N.o &
  done
  wait
  for V in "$@"; do
    N=${V%%:*}
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/abl/lib_$N.so $(ls $L/*.o | grep -v gemm256.o) $L/abl/gemm256_$N.o
  done
  rm -f $L/abl/*.o; ls $L/abl
else
  shift; NAMES=$1; shift
  for N in $NAMES; do
    printf "%-10s " $N; PST_LIB=$L/abl/lib_$N.so python $ROOT/tools/pp_time.py "$@" 2>/dev/null | grep us
  done
fi
