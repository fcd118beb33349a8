"""Diagnostic (GPU box): error of the `gradients` output (normals) against the float64 oracle, with and without CNR_NO_NARROW_DX (child processes)."""
import os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
CHILD = r'''
import os, sys
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch, test_edge_batches as T, _native as N
from oracle import colorneus_oracle as O
ocfg = O.dtu_config(); P = O.init_params(ocfg, seed=5, trained_like=True)
for R in (33, 96, 256):
    o, d, near, far, t_rand, gt, mask = T._batch(R, 100 + R)
    r = N.make_renderer(ocfg, P, None, torch.device("cuda:0"))
    z = O.sample_z(P, ocfg, o, d, near, far, t_rand)
    out = r(o.cuda(), d.cuda(), near.cuda(), far.cuda(), z_vals=z.cuda())
    P64 = {k: v.double() for k, v in P.items()}
    oo = O.render(P64, ocfg, o.double(), d.double(), near.double(), far.double(), z_vals=z.double())
    o32 = O.render(P, ocfg, o, d, near, far, z_vals=z)
    g, g64, g32 = out["gradients"].detach().cpu().double(), oo["gradients"], o32["gradients"].double()
    sc = float(g64.abs().max())
    print(R, "hip max %.3e rms %.3e | f32 oracle max %.3e rms %.3e" % (float((g - g64).abs().max()) / sc, float((g - g64).pow(2).mean().sqrt()) / sc,
                                                                     float((g32 - g64).abs().max()) / sc, float((g32 - g64).pow(2).mean().sqrt()) / sc))
'''
for env in ({}, {"CNR_NO_NARROW_DX": "1"}):
    print("env", env)
    r = subprocess.run([sys.executable, "-c", CHILD, root], env=dict(os.environ, **env), capture_output=True, text=True)
    print(r.stdout[-1500:], r.stderr[-800:] if r.returncode else "")
