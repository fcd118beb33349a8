"""Audit of the generated gfx950 code of every kernel: registers, spills, scratch, and loads that are waited for with `s_waitcnt vmcnt(0)` right
after they are issued.  Compiles each csrc/*.hip with -S into a temporary directory; runs on the CPU box (hipcc cross-compiles).
Usage: python tools/asm_audit.py [min_immediate_waits=3]"""
import glob, os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "color-neus_amd", "csrc")
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tmp = tempfile.mkdtemp(prefix="cnr_asm_")
for f in sorted(glob.glob(os.path.join(src, "*.hip"))):
    out = os.path.join(tmp, os.path.basename(f)[:-4] + ".s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only", "-I" + src, f, "-o", out],
                       capture_output=True, text=True)
    if r.returncode != 0:
        print("compile failed:", f, r.stderr[-500:]); continue
    txt = open(out).read()
    lines = txt.split("\n")
    waits, kern = {}, None
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN3cnr\w+):", l)
        if m: kern = m.group(1)
        if kern and "global_load" in l:
            for j in range(i + 1, min(i + 10, len(lines))):
                if "global_load" in lines[j] or "global_store" in lines[j]: break
                if "s_waitcnt vmcnt(0)" in lines[j]:
                    waits[kern] = waits.get(kern, 0) + 1; break
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
        name, priv, vg, sp = m.groups()
        w = waits.get(name, 0)
        if int(priv) > 0 or int(sp) > 0 or w >= thr:
            d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:110]
            print("%-22s %-112s vgpr %3s  spilled %3s  scratch %5s B  load->vmcnt(0) %3d" % (os.path.basename(f), d, vg, sp, priv, w))
