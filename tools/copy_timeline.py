#!/usr/bin/env python
"""Timeline of the device -> host copies of the streamed API call against its kernels, from a rocprofv3 --kernel-trace --memory-copy-trace run
(csv output directory as argument): per copy start / duration / GB/s relative to the first kernel of the last call, and the busy time of the kernels."""
import csv, glob, sys
d = sys.argv[1]
kt = [r for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True) for r in csv.DictReader(open(f))]
mc = [r for f in glob.glob(d + '/**/*memory_copy_trace.csv', recursive=True) for r in csv.DictReader(open(f))]
ks = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in kt)
print('copy trace columns:', list(mc[0].keys()) if mc else None)
szk = next((k for k in (mc[0].keys() if mc else []) if k.lower() in ('size', 'bytes', 'size_bytes')), None)
cs = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Direction', r.get('Name', '')), int(r.get(szk, 0) or 0) if szk else 0) for r in mc)
print('directions:', sorted({c[2] for c in cs}))
d2h = [c for c in cs if 'DEVICE_TO_HOST' in c[2].upper() or 'D2H' in c[2].upper() or 'DTOH' in c[2].upper()]
big = [c for c in (d2h or cs) if c[1] - c[0] > 1000000]     # copies longer than 1 ms: the output blocks
if not big:
    print('no large copies found; columns:', list(mc[0].keys()) if mc else None); sys.exit(0)
# the last call: the last run of large copies separated from the previous one by > 50 ms
last = [big[-1]]
for c in reversed(big[:-1]):
    if last[0][0] - c[1] > 50e6:
        break
    last.insert(0, c)
t_end = max(c[1] for c in last)
k_in = [k for k in ks if k[0] > last[0][0] - 200e6 and k[1] <= t_end + 5e6]
# first kernel of the call: after the longest kernel-free gap in that window
gaps = [(k_in[i + 1][0] - k_in[i][1], i + 1) for i in range(len(k_in) - 1)]
g = max(gaps)[1] if gaps else 0
t0 = k_in[g][0]
k_call = [k for k in k_in[g:]]
print('call: %.1f ms of kernels from first to last launch; last kernel ends at %.1f ms, last copy ends at %.1f ms' % (
    sum(e - s for s, e in k_call) / 1e6, (max(e for _, e in k_call) - t0) / 1e6, (t_end - t0) / 1e6))
for s_, e_, dr, sz in last:
    print('  copy %-14s start %7.1f ms  dur %6.2f ms  %7.1f MB  %5.1f GB/s' % (dr[:14], (s_ - t0) / 1e6, (e_ - s_) / 1e6, sz / 1e6, sz / max(e_ - s_, 1)))
