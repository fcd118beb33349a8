#!/usr/bin/env python3
"""Board power and clock while the matrix pipe alone runs (tools/probes/mfma_power_probe.hip): constant operands, then random operands.
Usage (GPU box, repository root): python tools/mfma_power.py   -- builds the probe with hipcc, samples rocm-smi every 0.25 s."""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "tools", "probes", "mfma_power_probe.hip")
exe = "/tmp/mfma_power_probe"
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
p = subprocess.Popen([exe, "5"], stdout=subprocess.PIPE, text=True)
marks = []
for line in p.stdout:
    line = line.strip(); marks.append((time.time(), line)); print(line, flush=True)
p.wait(); stop = True; th.join(timeout=3)
for name in ("constant", "random"):
    t0 = [t for t, l in marks if l.startswith("phase %s start" % name)][0] + 1.0
    t1 = [t for t, l in marks if l.startswith("phase %s end" % name)][0]
    ws = [w for t, w, c in samples if t0 <= t <= t1 and w]; cs = [c for t, w, c in samples if t0 <= t <= t1 and c]
    print("%-8s operands: W mean %.0f max %.0f   sclk mean %.0f MHz   (%d samples)" % (name, sum(ws) / max(len(ws), 1), max(ws or [0]), sum(cs) / max(len(cs), 1), len(ws)))
