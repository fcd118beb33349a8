"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel.  Usage: python tools/pmc_summary.py <dir-or-csv> [out.txt]"""
import csv, glob, os, sys, collections
src = sys.argv[1]
files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
for f in files:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        launches[k].add(row["Dispatch_Id"])
lines = []
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("GRBM_GUI_ACTIVE", 0))):
    wc = c.get("SQ_WAVE_CYCLES", 0)
    parts = ["%-70s launches %4d" % (k[:70], len(launches[k]))]
    for name, v in sorted(c.items()):
        if wc and name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                           "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_VMEM"):
            parts.append("%s %.1f%%" % (name[3:].lower(), 100 * v / wc))
        else:
            parts.append("%s %.3g" % (name, v))
    lines.append("  ".join(parts))
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
