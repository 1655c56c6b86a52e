#!/bin/bash
# tools/build_ab.sh <libA.so> <libB.so>: the memory build alone (tools/build_bench.py 16) and the small attention shapes under each build in turn, same box
L=panst3r_amd/lib/libpanst3r_hip.so
cp $2 /tmp/_ab_b.so; cp $1 /tmp/_ab_a.so
for v in a b a b; do
  cp /tmp/_ab_$v.so $L
  echo "== build $v"
  python tools/build_bench.py 16 2>&1 | grep -v amdgpu.ids
  python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | grep "build" | sed 's/plain .*| prescaled/prescaled/'
done
cp /tmp/_ab_b.so $L
