#!/usr/bin/env python
"""Per-shape kernel table of the headline scene from THREE instrumented bench runs: min / median / max of the summed ms per
(kernel, shape) so that a one-off outlier (round 2's committed table had a 26.8 ms line that other runs put at 6.7 ms) is visible as such.
    python tools/shape_profile3.py [bench args...]  > profiles/r3_shape_profile.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = []
pat = re.compile(r'^(\S+)\s+(\(.*\))\s+x(\d+)\s+([\d.]+) ms\s+([\d.]+) TF\s*$')
for i in range(3):
    env = dict(os.environ, PST_SHAPE_PROFILE='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--no-cpu-baseline', '--no-alt-dtype'] + sys.argv[1:], env=env,
                       capture_output=True, text=True)
    rows = {}
    for line in r.stderr.splitlines():
        m = pat.match(line)
        if m:
            rows[(m.group(1), m.group(2))] = (int(m.group(3)), float(m.group(4)), float(m.group(5)))
    runs.append(rows)
    print('# run %d: %d rows, bench line: %s' % (i, len(rows), r.stdout.strip()[:160]))
keys = sorted(set().union(*[set(r) for r in runs]), key=lambda k: -sorted(r[k][1] for r in runs if k in r)[len([r for r in runs if k in r]) // 2])
print('%-24s %-64s %-5s %9s %9s %9s %8s' % ('kernel', 'shape tag', 'x', 'min ms', 'median', 'max ms', 'med TF'))
for k in keys:
    v = sorted((r[k][1], r[k][2], r[k][0]) for r in runs if k in r)
    med = v[len(v) // 2]
    flag = '   <-- max > 1.5 x median' if v[-1][0] > 1.5 * med[0] else ''
    print('%-24s %-64s x%-4d %9.2f %9.2f %9.2f %8.1f%s' % (k[0], k[1], med[2], v[0][0], med[0], v[-1][0], med[1], flag))
