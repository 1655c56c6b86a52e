#!/bin/bash
# tools/attn_ab.sh <libA.so> <libB.so>: tools/attn_bench.py (prescaled column) under each build of libpanst3r_hip.so in turn, same box
L=panst3r_amd/lib/libpanst3r_hip.so
cp $2 /tmp/_ab_b.so; cp $1 /tmp/_ab_a.so
for v in a b a b; do
  cp /tmp/_ab_$v.so $L
  echo "== build $v"
  python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/plain .*| prescaled/prescaled/'
done
cp /tmp/_ab_b.so $L
