"""Child process of tests/test_edge_batches.py: one forward + backward of R rays (for every R of a comma-separated list) through the HIP library
under the CNR_* switches of the environment it was started with (the library reads them once per process); writes every output and every
parameter gradient to one .npz per R (argv[3] is a path template with %d for R)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    Rs, cfg_name, out_tmpl = [int(x) for x in sys.argv[1].split(",")], sys.argv[2], sys.argv[3]
    for R in Rs:
        one(R, cfg_name, out_tmpl % R)


def one(R, cfg_name, out_path):
    import color_neus_amd as cn
    import _golden as G
    import _native as N
    import test_edge_batches as T
    from oracle import colorneus_oracle as O      # inputs only (weights, rays, the sample positions): both sides of the comparison share them
    dev = torch.device("cuda:0")
    ocfg = T._config(cfg_name)
    P = O.init_params(ocfg, seed=5, trained_like=True)
    o, d, near, far, t_rand, gt, mask = T._batch(R, 100 + R)
    z = O.sample_z(P, ocfg, o, d, near, far, t_rand)
    r = N.make_renderer(ocfg, P, None, dev)
    od, dd = o.to(dev).requires_grad_(True), d.to(dev).requires_grad_(True)
    out = r(od, dd, near.to(dev), far.to(dev), z_vals=z.to(dev))
    loss, _ = cn.compute_loss(out, gt.to(dev), mask.to(dev))
    loss.backward()
    res = {"out:" + k: out[k].detach().cpu().numpy() for k in G.OUTPUT_KEYS if k in out}
    res.update({n: p.grad.detach().cpu().numpy() for n, p in r.named_parameters() if p.grad is not None})
    res["d_rays_o"], res["d_rays_d"] = od.grad.cpu().numpy(), dd.grad.cpu().numpy()
    np.savez(out_path, **res)


if __name__ == "__main__":
    main()
