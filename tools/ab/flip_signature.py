"""Diagnostic (GPU box): where do the two forms of the gradient chain's end differ in the colour head's weight gradient?  A flipped ReLU unit of
the last hidden layer shows up as ONE column.  Usage: python tools/ab/flip_signature.py [R=33] [seed_offset=100]"""
import os, subprocess, sys, tempfile
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
R = sys.argv[1] if len(sys.argv) > 1 else "33"
off = sys.argv[2] if len(sys.argv) > 2 else "100"
CHILD = r'''
import os, sys
root, R, off, path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch, test_edge_batches as T, _native as N
import color_neus_amd as cn
from oracle import colorneus_oracle as O
ocfg = O.dtu_config(); P = O.init_params(ocfg, seed=5, trained_like=True)
o, d, near, far, t_rand, gt, mask = T._batch(R, off + R)
r = N.make_renderer(ocfg, P, None, torch.device("cuda:0"))
z = O.sample_z(P, ocfg, o, d, near, far, t_rand)
out = r(o.cuda(), d.cuda(), near.cuda(), far.cuda(), z_vals=z.cuda())
loss, _ = cn.compute_loss(out, gt.cuda(), mask.cuda()); loss.backward()
np.savez(path, **{n: p.grad.detach().cpu().numpy() for n, p in r.named_parameters()})
'''
import numpy as np
tmp = tempfile.mkdtemp()
g = {}
for tag, env in (("dx", {}), ("fp32", {"CNR_NO_NARROW_DX": "1"})):
    path = os.path.join(tmp, tag + ".npz")
    r = subprocess.run([sys.executable, "-c", CHILD, root, R, off, path], env=dict(os.environ, **env), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    g[tag] = dict(np.load(path))
for n in ("color_network.lin3.weight_v", "color_network.lin2.weight_v", "color_network.lin3.bias"):
    a, b = g["dx"][n].astype(np.float64), g["fp32"][n].astype(np.float64)
    d = np.abs(a - b) / np.abs(b).max()
    print(n, a.shape, "max rel diff %.2e" % d.max())
    if d.ndim == 2:
        col = d.max(axis=0); row = d.max(axis=1)
        print("   columns above 1e-4:", np.nonzero(col > 1e-4)[0][:12], " rows above 1e-4:", np.nonzero(row > 1e-4)[0][:12], " (of %d x %d)" % a.shape)
