"""debug: per-key errors of the HIP path on the DTU-size variant fixtures, next to the float32 oracle's own distance from float64"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, numpy as np
import _golden as G, _native as N
from oracle import colorneus_oracle as O
for name in ["neus_dtu_nonormal", "dtu_rel_alt", "dtu_nown_skip6", "neus_dtu_sharp"]:
    for tag in ("det", "jit"):
        res = N.run_native(name, tag, None, "cuda:0", fixed_z=True)
        fx, r, out = res[0], res[1], res[2]
        cfg, P = G.weights_of(name, fx)
        o, d = torch.from_numpy(fx["rays_o"]), torch.from_numpy(fx["rays_d"])
        near, far, z = torch.from_numpy(fx[f"{tag}:near"]), torch.from_numpy(fx[f"{tag}:far"]), torch.from_numpy(fx[f"{tag}:z_vals"])
        o64 = O.render({k: v.double() for k, v in P.items()}, cfg, o.double(), d.double(), near.double(), far.double(), z_vals=z.double())
        o32 = O.render(P, cfg, o, d, near, far, z_vals=z)
        line = []
        for k in G.OUTPUT_KEYS:
            if f"{tag}:out_{k}" in fx:
                ref = fx[f"{tag}:out_{k}"]
                e_g = G.relerr(out[k].detach().cpu().reshape(ref.shape), ref)
                e_64 = G.relerr(out[k].detach().cpu().reshape(ref.shape), o64[k].detach().reshape(ref.shape))
                r_64 = G.relerr(torch.from_numpy(ref), o64[k].detach().reshape(ref.shape))
                line.append("%s hip-gold %.1e hip-f64 %.1e gold-f64 %.1e" % (k, e_g, e_64, r_64))
        print(name, tag)
        for l in line:
            print("    ", l)
