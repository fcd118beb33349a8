#!/usr/bin/env python3
"""Energy ledger per kernel family (round 6, review item 1): joules per pass = board W x time, split into
   MFMA   f16 MFMA FLOP of the pass / (FLOP per joule of the matrix pipe alone on random operands, above the idle draw)
   memory algorithmic HBM bytes of the pass x (joules per byte of a plain device copy, above the idle draw: fabric + L2 + controllers + HBM)
   idle   idle W x time
   rest   what is left: LDS, registers, VALU, instruction issue, L2 weight streams.
The two unit costs are measured in the same run on the same board (copy family; tools/probes/mfma_power_probe.hip when hipcc is there).

Each family is LOOPED on its own for a few seconds through the public entry points (tools/family_power.py) while rocm-smi is sampled;
FLOP and bytes of one pass come from the library's own per-launch records (cnr_timing_enable) of one instrumented pass.

Usage (GPU box, repository root):  python tools/energy_ledger.py [--seconds 4] [--rays 4096] > gpurun_out/r06_energy_ledger.txt
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import family_power as FP   # noqa: E402

SPLIT_F16 = ("layer_gemm_ws", "layer_dw", "dw_gemm_hx", "chain_sdf_value", "chain_sdf_fwd", "chain_fwd", "chain_sdf_grad", "sweep0_dw",
             "narrow_bwd", "narrow_dx")   # 3 f16 MFMAs per fp32-equivalent product (the library's records count fp32-equivalent FLOP)


def family(name, seconds, rays):
    """one family in this process: prints a JSON record"""
    import torch
    import color_neus_amd as cn
    from color_neus_amd import synthetic
    dev = torch.device("cuda:0")
    libpath = os.environ.get("CNR_LIB") or None    # the ablation families run on the tuning build (make hip-tuning): the product has no ablation words
    lib = cn.load_library(libpath)
    cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
    torch.manual_seed(0)
    r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg, library=libpath)).to(dev)
    views = synthetic.synthetic_view(seed=1, device=dev)
    sel = torch.randperm(views[0].shape[0], generator=torch.Generator().manual_seed(7))[:rays].to(dev)
    o, d, n, f, gt, m = [x[sel] for x in views]
    setup = lambda: None
    if name.startswith("backward"):
        state = {}

        def setup():
            out = r(o, d, n, f, perturb_overwrite=0)
            state["loss"], _ = cn.compute_loss_fused(out, gt, m, library=lib)

        def body():
            for p in r.parameters():
                p.grad = None
            state["loss"].backward(retain_graph=True)
    elif name == "forward_saving":
        def body():
            with torch.no_grad():
                r(o, d, n, f, perturb_overwrite=0, forward_only=False)
    elif name == "forward_only":
        def body():
            with torch.no_grad():
                r(o, d, n, f, perturb_overwrite=0)
    elif name == "sdf_value":
        pts = torch.rand(1 << 21, 3, device=dev) * 2 - 1

        def body():
            r.sdf(pts)
    elif name == "copy":
        a = torch.empty(1 << 28, device=dev)
        b = torch.empty_like(a)

        def body():
            b.copy_(a)
    elif name == "idle":
        def body():
            time.sleep(0.05)
    else:
        raise SystemExit("unknown family " + name)
    setup()
    body()
    torch.cuda.synchronize()
    # one instrumented pass: FLOP, bytes and kernel time of the pass from the library's own records
    lib.timing_enable(True)
    lib.timing_collect()
    body()
    torch.cuda.synchronize()
    recs = lib.timing_collect()
    lib.timing_enable(False)
    fl = by = kms = 0.0
    per = {}
    for kname, kind, nt, P, N, K, pairs, ms, nbytes in recs:
        fk = (2.0 * P * N * K * max(pairs, 1) if kind != 2 else 0.0) * (3.0 if kname in SPLIT_F16 else 0.0)
        fl += fk
        by += nbytes
        kms += ms
        a = per.setdefault(kname, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += ms; a[2] += fk; a[3] += nbytes
    if name == "copy":
        by = 2.0 * (1 << 30)
    s = FP.Sampler()
    s.start()
    t0 = time.time()
    n_done = 0
    while time.time() - t0 < seconds:
        for _ in range(8):
            body()
        n_done += 8
        torch.cuda.synchronize()
    dt = time.time() - t0
    s.stop = True
    s.join(timeout=3)
    ws = [w for t, w, c in s.samples if t > 1.0 and w is not None]
    cs = [c for t, w, c in s.samples if t > 1.0 and c is not None]
    mean = lambda v: sum(v) / len(v) if v else float("nan")
    print("LEDGER " + json.dumps({"family": name, "ms_per_pass": dt / n_done * 1e3, "W": mean(ws), "Wmax": max(ws) if ws else None, "sclk": mean(cs),
                                  "f16_mfma_tflop": fl / 1e12, "gbytes": by / 1e9, "kernel_ms": kms, "launches": len(recs),
                                  "kernels": {k: [v[0], round(v[1], 4), round(v[2] / 1e12, 4), round(v[3] / 1e9, 4)] for k, v in per.items()}}), flush=True)


def mfma_unit_cost():
    """W and TFLOP/s of the matrix pipe alone on random operands (tools/mfma_power.py); None if the probe cannot be built here"""
    try:
        t = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_power.py")], capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return None
    tf = re.search(r"phase random end:.*?([0-9.]+) TFLOP/s", t)
    w = re.search(r"random\s+operands: W mean ([0-9.]+)", t)
    return (float(tf.group(1)), float(w.group(1)), t) if tf and w else None


def mfma_shape_costs():
    """(TFLOP/s, W) of the matrix pipe alone per MFMA shape (tools/mfma_order.py --shapes): {"16x16x32": .., "32x32x16": ..}; {} if the probe cannot run here"""
    try:
        t = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mfma_order.py"), "--shapes"], capture_output=True, text=True, timeout=180).stdout
    except Exception:
        return {}, ""
    out = {}
    for shape in ("16x16x32", "32x32x16"):
        m = re.search(r"v_mfma_f32_%s_f16[^\n]*?([0-9.]+) TFLOP/s\s+W mean\s+([0-9.]+)" % shape, t)
        if m:
            out[shape] = (float(m.group(1)), float(m.group(2)))
    return out, t


# kernels whose split-f16 products are issued as v_mfma_f32_16x16x32_f16 (round 6): the fused layer + weight-gradient kernel and the stream form of the
# layer kernel (6 of the 7 launches of the layer_gemm_ws family in a step; the family is priced at that shape).  Everything else is on 32x32x16.
MFMA16_KERNELS = ("layer_dw", "layer_gemm_ws")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--family", default=None)
    ap.add_argument("--no-mfma-probe", action="store_true")
    a = ap.parse_args()
    if a.family:
        family(a.family, a.seconds, a.rays)
        return
    fams = [("idle", {}), ("copy", {}), ("sdf_value", {}), ("forward_only", {}), ("forward_saving", {}), ("backward", {}),
            ("backward_no_dw_mfma", {"CNR_FDW_DBG": "1"}), ("backward_no_product_mfma", {"CNR_FDW_DBG": "4"}), ("backward_no_mfma", {"CNR_FDW_DBG": "5"})]
    rec = {}
    tuning = os.path.join(ROOT, "tools", "_build", "libcolorneus_hip_tuning.so")
    for fam, env in fams:
        if env and not os.path.isfile(tuning):
            print("# %s skipped: no tuning build (make -C color-neus_amd/csrc hip-tuning)" % fam)
            continue
        e = dict(os.environ, **env, **({"CNR_LIB": tuning} if env else {}))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--family", fam, "--seconds", str(a.seconds), "--rays", str(a.rays)],
                             env=e, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith("LEDGER "):
                rec[fam] = json.loads(line[7:])
    idle_w = rec["idle"]["W"]
    cp = rec["copy"]
    j_per_gb = (cp["W"] - idle_w) * cp["ms_per_pass"] * 1e-3 / cp["gbytes"]
    mf = None if a.no_mfma_probe else mfma_unit_cost()
    if mf:
        tflop_per_j = mf[0] / (mf[1] - idle_w)
        src = "measured in this run: %.0f TFLOP/s at %.0f W" % (mf[0], mf[1])
    else:
        tflop_per_j = 1603.0 / (1236.0 - 297.0)
        src = "profiles/r05_mfma_power.txt: 1603 TFLOP/s at 1236 W over 297 W idle"
    shapes, shapes_txt = ({}, "") if a.no_mfma_probe else mfma_shape_costs()
    price = {"32x32x16": tflop_per_j, "16x16x32": tflop_per_j}
    for sh, (tf, w) in shapes.items():
        price[sh] = tf / (w - idle_w)
    def kprice(name):
        return price["16x16x32"] if name in MFMA16_KERNELS else price["32x32x16"]
    def fam_mfma_j(r, drop=None):
        """MFMA joules of a family = sum over its kernels at each kernel's shape price; drop = (kernel, TFLOP) switched off in an ablation run"""
        j = 0.0
        for k, v in r["kernels"].items():
            tf = v[2] - (drop[1] if drop and drop[0] == k else 0.0)
            j += max(tf, 0.0) / kprice(k)
        return j
    print("# energy ledger, %d rays (%d points) per pass; rocm-smi every 0.25 s while one family loops for %.0f s (first second dropped)" % (a.rays, a.rays * 128, a.seconds))
    print("# unit costs above the idle draw (%.0f W): memory %.1f pJ/B (device copy: %.2f TB/s of traffic at %.0f W); matrix pipe %.2f TFLOP/J (%s)" %
          (idle_w, j_per_gb * 1e3, cp["gbytes"] / cp["ms_per_pass"], cp["W"], tflop_per_j, src))
    if shapes:
        print("# matrix pipe by MFMA shape (tools/mfma_order.py --shapes, same run): " + "; ".join("%s %.0f TFLOP/s at %.0f W = %.2f TFLOP/J" % (sh, tf, w, price[sh]) for sh, (tf, w) in sorted(shapes.items())) +
              " -- layer_dw and the layer_gemm_ws family are priced at 16x16x32, every other kernel at 32x32x16")
    print("# rest = J - MFMA - memory - idle: LDS, registers, VALU, issue, L2 weight streams (the fused launches' second read of their input tile is an L2 hit and not in 'GB')")
    print("%-26s %8s %7s %6s %7s | %7s %7s %7s %7s | %6s %6s  %s" % ("family", "ms/pass", "W", "MHz", "J/pass", "MFMA J", "mem J", "idle J", "rest J", "TFLOP", "GB", "rest/J"))
    for fam, _ in fams:
        r = rec.get(fam)
        if not r:
            continue
        J = r["W"] * r["ms_per_pass"] * 1e-3
        jm = fam_mfma_j(r)
        if fam == "backward_no_dw_mfma" or fam == "backward_no_product_mfma":
            # half of the fused launches' MFMAs are switched off: take them out of the FLOP count
            k = rec["backward"]["kernels"].get("layer_dw", [0, 0, 0, 0])
            jm = fam_mfma_j(r, ("layer_dw", 0.5 * k[2]))
        if fam == "backward_no_mfma":
            k = rec["backward"]["kernels"].get("layer_dw", [0, 0, 0, 0])
            jm = fam_mfma_j(r, ("layer_dw", k[2]))
        jb = r["gbytes"] * j_per_gb
        ji = idle_w * r["ms_per_pass"] * 1e-3
        rest = J - jm - jb - ji
        print("%-26s %8.3f %7.0f %6.0f %7.2f | %7.2f %7.2f %7.2f %7.2f | %6.2f %6.2f  %.2f" %
              (fam, r["ms_per_pass"], r["W"], r["sclk"], J, jm, jb, ji, rest, r["f16_mfma_tflop"], r["gbytes"], rest / J if J else 0))
    print()
    print("# kernels of one pass (launches, ms, f16 MFMA TFLOP, algorithmic GB) -- from one instrumented pass; a family's J is apportioned by these")
    for fam in ("sdf_value", "forward_saving", "backward"):
        r = rec.get(fam)
        if not r:
            continue
        J = r["W"] * r["ms_per_pass"] * 1e-3
        print("%s: %.3f ms kernel time of %.3f ms per pass" % (fam, r["kernel_ms"], r["ms_per_pass"]))
        for k, v in sorted(r["kernels"].items(), key=lambda kv: -kv[1][1]):
            share = v[1] / r["kernel_ms"] if r["kernel_ms"] else 0
            jk = J * share
            jmk = v[2] / kprice(k)
            jbk = v[3] * j_per_gb
            jik = idle_w * r["ms_per_pass"] * 1e-3 * share
            print("   %-22s x%-3d %7.3f ms  %6.3f TFLOP %6.2f GB | J/launch %.4f = MFMA %.4f + mem %.4f + idle %.4f + rest %.4f" %
                  (k, v[0], v[1], v[2], v[3], jk / v[0], jmk / v[0], jbk / v[0], jik / v[0], (jk - jmk - jbk - jik) / v[0]))
    if mf:
        print()
        print("# matrix pipe alone (tools/mfma_power.py):")
        for line in mf[2].splitlines():
            print("#   " + line)
    if shapes_txt:
        print("# matrix pipe by shape (tools/mfma_order.py --shapes):")
        for line in shapes_txt.splitlines():
            print("#   " + line)


if __name__ == "__main__":
    main()
