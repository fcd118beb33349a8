"""Diagnostic (GPU box): is the deviation of the HIP parameter gradients at a tiny batch explained by ReLU kinks?  The float64 oracle is run twice:
as is, and with the HIP build's own normals error (HIP normals - float64 normals, ~2e-7) added to its normals.  If the float64 gradients move by what
the HIP gradients deviate by, the float64 objective itself is that sensitive at this batch (a kink event), not the backward kernels.
Usage: python tools/ab/kink_probe.py [R=33] [seed_offset=100]   (runs both forms of the gradient chain's end in child processes)"""
import os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 33
off = int(sys.argv[2]) if len(sys.argv) > 2 else 100
CHILD = r'''
import os, sys
root, R, off = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch, test_edge_batches as T, _native as N
import color_neus_amd as cn
from oracle import colorneus_oracle as O
ocfg = O.dtu_config(); P = O.init_params(ocfg, seed=5, trained_like=True)
o, d, near, far, t_rand, gt, mask = T._batch(R, off + R)
r = N.make_renderer(ocfg, P, None, torch.device("cuda:0"))
z = O.sample_z(P, ocfg, o, d, near, far, t_rand)
out = r(o.cuda(), d.cuda(), near.cuda(), far.cuda(), z_vals=z.cuda())
loss, _ = cn.compute_loss(out, gt.cuda(), mask.cuda()); loss.backward()
ghip = {n: p.grad.detach().cpu().double() for n, p in r.named_parameters()}
def run(delta):
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    orig = O.sdf_forward
    def patched(P_, cfg_, x, want_grad=False):
        res = orig(P_, cfg_, x, want_grad)
        if want_grad and delta is not None: return res[0], res[1], res[2] + delta
        return res
    O.sdf_forward = patched
    try:
        oo = O.render(P64, ocfg, o.double(), d.double(), near.double(), far.double(), z_vals=z.double())
    finally:
        O.sdf_forward = orig
    l, _ = O.compute_loss(oo, gt.double(), mask.double()); l.backward()
    return oo, {k: v.grad for k, v in P64.items()}
oo, gA = run(None)
delta = (out["gradients"].detach().cpu().double() - oo["gradients"].detach()).reshape(-1, 3)
_, gB = run(delta)
print("normals error of this build: max %.2e of scale" % (float(delta.abs().max()) / float(oo["gradients"].abs().max())))
rows = []
for n in gA:
    if gA[n] is None: continue
    sc = float(gA[n].abs().max())
    if sc == 0: continue
    rows.append((float((ghip[n].reshape(gA[n].shape) - gA[n]).abs().max()) / sc, float((gB[n] - gA[n]).abs().max()) / sc,
                 float((ghip[n].reshape(gA[n].shape) - gB[n]).abs().max()) / sc, n))
rows.sort(reverse=True)
print("%-44s %12s %16s %16s" % ("tensor (5 largest HIP deviations)", "HIP - f64", "f64(+dn) - f64", "HIP - f64(+dn)"))
for a, b, c, n in rows[:5]: print("%-44s %12.2e %16.2e %16.2e" % (n, a, b, c))
'''
for env in ({}, {"CNR_NO_NARROW_DX": "1"}):
    print("env", env)
    r = subprocess.run([sys.executable, "-c", CHILD, root, str(R), str(off)], env=dict(os.environ, **env), capture_output=True, text=True)
    print(r.stdout[-2500:], r.stderr[-1500:] if r.returncode else "")
