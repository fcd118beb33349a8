"""Idle time between kernels of a training step from a rocprofv3 --kernel-trace csv: busy time, gaps and the gap histogram over the steady-state
part of the trace (the last `frac` of the kernels).  Usage: python tools/launch_gaps.py <kernel_trace.csv> [frac=0.5]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[int(len(ev) * (1 - frac)):]
span = ev[-1][1] - ev[0][0]
busy = sum(e - s for s, e, _ in ev)
gaps = [max(0, ev[i + 1][0] - ev[i][1]) for i in range(len(ev) - 1)]
print("kernels %d  span %.3f ms  busy %.3f ms (%.1f %%)  idle %.3f ms" % (len(ev), span / 1e6, busy / 1e6, 100.0 * busy / span, (span - busy) / 1e6))
edges = [0, 1000, 2000, 4000, 8000, 16000, 32000, 64000, 1 << 62]
for lo, hi in zip(edges[:-1], edges[1:]):
    g = [x for x in gaps if lo <= x < hi]
    print("gap %6.0f..%-8s us  n=%5d  total %.3f ms" % (lo / 1e3, ("%.0f" % (hi / 1e3)) if hi < 1 << 60 else "inf", len(g), sum(g) / 1e6))
big = sorted(((gaps[i], ev[i][2][:50], ev[i + 1][2][:50]) for i in range(len(gaps))), reverse=True)[:12]
for g, a, b in big:
    print("  %.1f us after %s before %s" % (g / 1e3, a, b))
