#!/usr/bin/env python
"""Summarise a rocprofv3 kernel trace: per kernel name count / mean duration, and the idle gaps between consecutive kernels
(sorted by start time) -- tells launch-latency-bound chains (memory build) from kernel-time-bound ones."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0           # ignore the first N kernels (warm-up / capture)
rows = rows[skip:]
dur = collections.defaultdict(list)
gaps = []
prev_end = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    dur[r['Kernel_Name'].split('(')[0][-60:]].append(e - s)
    if prev_end is not None:
        gaps.append(max(0, s - prev_end))
    prev_end = max(prev_end or 0, e)
tot = rows and (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) or 0
busy = sum(sum(v) for v in dur.values())
print('kernels %d  span %.2f ms  sum(kernel) %.2f ms  sum(gaps) %.2f ms  mean gap %.2f us' % (len(rows), tot / 1e6, busy / 1e6, sum(gaps) / 1e6, sum(gaps) / max(1, len(gaps)) / 1e3))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print('%-62s x%-5d mean %7.2f us  total %7.2f ms' % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
