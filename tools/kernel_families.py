"""Merge the template instantiations of a rocprofv3 --kernel-trace --stats kernel_stats CSV into kernel families.
Usage: python tools/kernel_families.py <kernel_stats.csv> [out.txt] [header line]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r["Name"]
    m = re.search(r"cnr::(\w+?)(_kernel)?(<|\()", name)
    fam = m.group(1) if m else "torch/other"
    a = agg.setdefault(fam, [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
tot = sum(a[1] for a in agg.values())
lines = [sys.argv[3]] if len(sys.argv) > 3 else []
lines.append("%-24s %8s %14s %12s %7s" % ("family", "calls", "total_ms", "avg_us", "share"))
for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append("%-24s %8d %14.3f %12.2f %6.1f%%" % (k, c, ns / 1e6, ns / c / 1e3, 100.0 * ns / tot))
if "layer_gemm_ws" in agg and "layer_gemm_ws_stream" in agg:   # bench.py times both forms of the layer kernel as one family
    c = agg["layer_gemm_ws"][0] + agg["layer_gemm_ws_stream"][0]
    ns = agg["layer_gemm_ws"][1] + agg["layer_gemm_ws_stream"][1]
    lines.append("%-24s %8d %14.3f %12.2f %6.1f%%   (layer_gemm_ws + layer_gemm_ws_stream: the family of bench.py's roofline)" %
                 ("layer_gemm_ws (both)", c, ns / 1e6, ns / c / 1e3, 100.0 * ns / tot))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
