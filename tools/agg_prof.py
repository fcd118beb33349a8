"""Aggregate a tools/prof_layers.py dump: per-kernel totals + the P=max layer launches.  Usage: python tools/agg_prof.py file [--all]"""
import re, sys, collections
lines = open(sys.argv[1]).read().splitlines()
agg = collections.OrderedDict()
rows = []
for l in lines:
    m = re.match(r'\s*(\d+) (\S+)\s+nt=(\d+)\s+P=(\d+)\s+N=(\d+)\s+K=(\d+)\s+pairs=(\d)\s+([\d.]+) ms\s+([\d.]+) TF/s\s+([\d.]+) GB/s', l)
    if not m:
        if l.startswith('total'): print(l)
        continue
    name = m.group(2); ms = float(m.group(8))
    a = agg.setdefault(name, [0, 0.0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += float(m.group(9)) * ms; a[3] += float(m.group(10)) * ms
    rows.append((int(m.group(1)), name, int(m.group(4)), int(m.group(5)), int(m.group(6)), ms, float(m.group(9)), float(m.group(10))))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]:
    print("%-18s %3d %8.3f ms  %6.1f TF/s %7.1f GB/s" % (k, v[0], v[1], v[2] / v[1], v[3] / v[1]))
if '--all' in sys.argv:
    pmax = max(r[2] for r in rows)
    for r in rows:
        if r[2] == pmax and 'gemm' in r[1]: print("%3d %-14s N=%-4d K=%-4d %7.3f ms %6.1f TF/s %7.1f GB/s" % (r[0], r[1], r[3], r[4], r[5], r[6], r[7]))
