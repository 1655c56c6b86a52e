#!/bin/bash
# A/B two builds of libpanst3r_hip.so on the same GPU box: tools/ab_bench.sh <libA.so> <libB.so> [rounds] [bench args...]
# (copies each library over panst3r_amd/lib/libpanst3r_hip.so in turn and prints the bench value; restores B at the end)
A=$1; B=$2; N=${3:-3}; shift 3
L=panst3r_amd/lib/libpanst3r_hip.so
cp $B /tmp/_ab_b.so; cp $A /tmp/_ab_a.so
for i in $(seq $N); do
  for v in a b; do
    cp /tmp/_ab_$v.so $L
    python bench.py --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('$v', b['value'], b['ms_per_step'])"
  done
done
cp /tmp/_ab_b.so $L
