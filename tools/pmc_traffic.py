"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as the guide prescribes).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts the 128-byte requests of wide (16 B/lane) streaming reads at
64 B, so it is doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE matched a known 512 MiB output exactly and is used as is.
Kernel template instantiations are merged by base name (layer_gemm_ws_kernel<...> -> layer_gemm_ws).
"""
import csv, glob, json, os, re, sys, collections


def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        m = re.search(r"cnr::(\w+?)(_kernel)?<|cnr::(\w+?)(_kernel)?\(", r["Kernel_Name"])
        name = (m.group(1) or m.group(3)) if m else r["Kernel_Name"][:40]
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"]) * 1024.0
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {}
for k in fetch:
    n = fetch[k][0]
    out[k] = {"launches": n, "fetch_bytes_per_launch": round(2.0 * fetch[k][1] / n), "write_bytes_per_launch": round(write[k][1] / max(write[k][0], 1)),
              "hbm_bytes_per_launch": round(2.0 * fetch[k][1] / n + write[k][1] / max(write[k][0], 1))}
if "layer_gemm_ws" in out and "layer_gemm_ws_stream" in out:   # one family for bench.py (its timing records do not tell the two forms apart)
    g, st = out["layer_gemm_ws"], out["layer_gemm_ws_stream"]
    n = g["launches"] + st["launches"]
    out["layer_gemm_ws_general"] = g
    out["layer_gemm_ws"] = {"launches": n, **{f: round((g[f] * g["launches"] + st[f] * st["launches"]) / n)
                                               for f in ("fetch_bytes_per_launch", "write_bytes_per_launch", "hbm_bytes_per_launch")}}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only` "
                   "(4096 rays/step); FETCH_SIZE doubled per the gfx950 correction", "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
    print(k, v)
