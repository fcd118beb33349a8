#!/usr/bin/env python3
"""Board power and clock while the matrix pipe runs the three MFMAs of a split product in the shipped order / the Gray order / with both operands changing
every time (tools/probes/mfma_order_probe.hip).  Usage (GPU box, repository root): python tools/mfma_order.py"""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "tools", "probes", "mfma_order_probe.hip")
exe = "/tmp/mfma_order_probe"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", src, "-o", exe], check=True)
samples = []
stop = False
def smi():
    t = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
    w = re.search(r"Power \(W\):\s*([0-9.]+)", t); c = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", t)
    return (float(w.group(1)) if w else None), (float(c.group(1)) if c else None)
def sampler():
    while not stop:
        samples.append((time.time(),) + smi()); time.sleep(0.25)
th = threading.Thread(target=sampler, daemon=True); th.start()
p = subprocess.Popen([exe, "4"] + (["256", "shapes"] if "--shapes" in sys.argv else []), stdout=subprocess.PIPE, text=True)
marks = []
for line in p.stdout:
    line = line.strip(); marks.append((time.time(), line)); print(line, flush=True)
p.wait(); stop = True; th.join(timeout=3)
starts = [(t, l) for t, l in marks if " start" in l]; ends = [(t, l) for t, l in marks if " end" in l]
names = {"mode3": "constant operands", "mode0": "shipped order (a1 w2, a2 w1, a1 w1)", "mode1": "Gray order (a1 w2, a1 w1, a2 w1)", "mode2": "both operands change every MFMA", "mode4": "v_mfma_f32_16x16x32_f16, shipped order, two accumulators", "mode5": "v_mfma_f32_32x32x16_f16, shipped order, two accumulators"}
for (t0, l0), (t1, l1) in zip(starts, ends):
    ws = [w for t, w, c in samples if t0 + 1.0 <= t <= t1 and w]; cs = [c for t, w, c in samples if t0 + 1.0 <= t <= t1 and c]
    tf = float(re.search(r"([0-9.]+) TFLOP/s", l1).group(1))
    W = sum(ws) / max(len(ws), 1)
    print("%-40s %7.0f TFLOP/s  W mean %5.0f  sclk %5.0f MHz  -> %.3f TFLOP/J above 300 W idle" % (names[l0.split()[1]], tf, W, sum(cs) / max(len(cs), 1), tf / max(W - 300.0, 1.0)))
