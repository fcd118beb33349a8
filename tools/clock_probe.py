"""Effective shader clock per kernel family = GRBM_GUI_ACTIVE / dispatch duration (rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace).
Usage: python tools/clock_probe.py <rocprof output dir>"""
import csv, glob, os, sys, collections
d = sys.argv[1]
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
dur = {}
for f in kt:
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for f in cc:
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
            continue
        ns, name = dur[r["Dispatch_Id"]]
        key = name.split("(")[0][:60]
        a = agg[key]
        a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
for k, (cyc, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("%-62s n=%4d  avg %.3f ms  clock %.2f GHz" % (k, n, ns / n * 1e-6, cyc / ns if ns else 0))
