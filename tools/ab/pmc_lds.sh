#!/bin/bash
# LDS counters per kernel (bank conflicts, LDS instruction counts, waits) over two steps of the 4096-ray bench
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06_lds; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-roofline --no-small-batch --no-torch-gpu-baseline --no-inference --no-c5 --no-loss-only"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_WAVE_CYCLES -d $OUT/lds -o l --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 $ARGS > $OUT/lds.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os, re
f = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."), "gpurun_out/r06_lds/lds/**/*counter_collection.csv"), recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]
for k, v in rows:
    idx = max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1)
    print("%-72s launches %3d  LDS insts %.3g  idx_active %.3g  bank_conflict %.3g (%.1f %% of active)  unaligned %.3g  addr_conflict %.3g  wait_lds/wave_cycles %.1f %%" % (
        k, cnt[k], v.get("SQ_INSTS_LDS", 0), v.get("SQ_LDS_IDX_ACTIVE", 0), v.get("SQ_LDS_BANK_CONFLICT", 0), 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / idx,
        v.get("SQ_LDS_UNALIGNED_STALL", 0), v.get("SQ_LDS_ADDR_CONFLICT", 0), 100 * v.get("SQ_WAIT_INST_LDS", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)))
PY
