#!/bin/bash
# tools/epi_ab.sh <libA.so> <libB.so>: tools/epi_ab.py under each build of libpanst3r_hip.so in turn (same box, same process layout)
L=panst3r_amd/lib/libpanst3r_hip.so
cp $2 /tmp/_ab_b.so; cp $1 /tmp/_ab_a.so
for v in a b a b; do
  cp /tmp/_ab_$v.so $L
  echo "== build $v"
  python tools/epi_ab.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/_ab_b.so $L
