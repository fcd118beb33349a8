"""Edge batch sizes through the product route: empty, single-ray and ragged batches (ray counts that do not fill a 32-point tile row
group, a wavefront of rays or a weight-gradient slab), forward + backward, against the oracle on the same inputs.
The reference accepts any batch size, including none (torch ops on empty tensors; NeuS.py:294-408)."""
import os

import pytest
import torch

import _golden as G
import _native as N

TOL = 1e-4


def _batch(R, seed):
    from oracle import colorneus_oracle as O
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(R, 3, generator=g)
    o = o / o.norm(dim=-1, keepdim=True).clamp_min(1e-6) * 2.7
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) * 0.3 - o, dim=-1)
    near, far = O.near_far_from_sphere(o, d) if R else (torch.zeros(0, 1), torch.zeros(0, 1))
    t_rand = torch.rand(R, 1, generator=g)
    gt = torch.rand(R, 3, generator=g)
    mask = (torch.rand(R, generator=g) > 0.3).float()
    return o, d, near, far, t_rand, gt, mask


def _config(cfg_name):
    from oracle import colorneus_oracle as O
    if cfg_name == "dtu":
        return O.dtu_config()
    if cfg_name == "dtu_pe4":   # DTU widths with a 27-column embedding (SDF MULTIRES 4): the first SDF layer's K pads to 32, not 48
        c = O.dtu_config()
        c.sdf.multires = 4
        return c
    return O.tiny_config()


def _check(R, library, device, cfg_name):
    from oracle import colorneus_oracle as O
    import color_neus_amd as cn
    ocfg = _config(cfg_name)
    P = O.init_params(ocfg, seed=5, trained_like=True)
    o, d, near, far, t_rand, gt, mask = _batch(R, 100 + R)
    r = N.make_renderer(ocfg, P, library, device)
    if R:
        z = O.sample_z(P, ocfg, o, d, near, far, t_rand)
    else:
        z = torch.zeros(0, ocfg.n_samples + ocfg.n_importance)
    out = r(o.to(device), d.to(device), near.to(device), far.to(device), z_vals=z.to(device))
    loss, _ = cn.compute_loss(out, gt.to(device), mask.to(device))
    for p in r.parameters():
        p.grad = None
    loss.backward()
    M = ocfg.n_samples + ocfg.n_importance
    assert out["color_fine"].shape == (R, 3) and out["weights"].shape[0] == R and out["gradients"].shape == (R, M, 3)
    if R == 0:
        # nothing to render: outputs are empty, the eikonal mean is 0 / 1e-5 = 0 and every gradient is exactly zero
        assert float(out["gradient_error"].detach()) == 0.0
        for n_, p in r.named_parameters():
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n_
        return
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    oo = O.render(P64, ocfg, o.double(), d.double(), near.double(), far.double(), z_vals=z.double())
    l64, _ = O.compute_loss(oo, gt.double(), mask.double())
    l64.backward()
    # the reference's own float32 round-off on these inputs (same oracle in float32): the yardstick of the bulk rule below
    P32 = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    o32 = O.render(P32, ocfg, o, d, near, far, z_vals=z)
    l32, _ = O.compute_loss(o32, gt, mask)
    l32.backward()
    loose = {"weights": 5e-4, "weight_max": 5e-4, "cdf_fine": 5e-4}
    for k in G.OUTPUT_KEYS:
        e = G.relerr(out[k].detach().cpu().reshape(oo[k].shape), oo[k].detach())
        assert e < loose.get(k, TOL), (R, k, e)
    assert abs(float(loss.detach()) - float(l64.detach())) < TOL * max(1.0, abs(float(l64.detach())))
    # parameter gradients at each tensor's own scale against float64 (the golden gate's rule: hard cap 1e-3, bulk within
    # max(1e-4, 3 x the reference's own float32 round-off))
    for n_, p in r.named_parameters():
        ref = P64[n_].grad
        if ref is None:
            continue
        got = p.grad.detach().cpu().double().reshape(ref.shape)
        scale = float(ref.abs().max())
        if scale == 0.0:
            assert float(got.abs().max()) < 1e-12, n_
            continue
        err = float((got - ref).abs().max()) / scale
        spread = float((P32[n_].grad.double() - ref).abs().max()) / scale
        # cap: 1e-3 at the golden batch sizes; a batch of a few dozen rays has so few points per ReLU unit that ONE pre-activation within
        # float32 round-off of the kink moves an entry by far more than that.  Measured on the HIP build at 33 rays
        # (tools/ab/flip_signature.py): the two forms of the gradient chain's end, whose normals agree to 1e-7, differ in ONE row (unit 169)
        # of color_network.lin3.weight_v by 2e-2 of the tensor's scale and in every entry of the biases below it by up to 6e-3 -- the
        # signature of a single flipped unit at the heaviest point; the float64 objective itself does not move under that perturbation
        # (tools/ab/kink_probe.py).  Hence below 64 rays: 5e-2 per entry, and the bulk rule on the MEDIAN entry at 1e-3.  Wrong tile or
        # slot handling at a ragged size shows as O(1) errors or NaN; accuracy is held by the strict gates at 2048+ points (test_hip_parity.py)
        small = R < 64
        assert err < max(5e-2 if small else 5e-3, 3.0 * spread), (R, n_, err, spread)
        bulk = 10.0 * TOL if small else TOL
        frac = float(((got - ref).abs() / scale > max(bulk, 3.0 * spread)).double().mean())
        assert frac < (0.5 if small else 0.25), (R, n_, frac, spread)


@pytest.mark.parametrize("R", [0, 1, 3, 33])
def test_edge_batches_emu(R):
    _check(R, N.EMU_LIB, torch.device("cpu"), "tiny")


@pytest.mark.gpu
@pytest.mark.parametrize("R", [0, 1, 3, 33, 130])
def test_edge_batches_hip(R):
    _check(R, None, torch.device("cuda:0"), "dtu")


@pytest.mark.gpu
@pytest.mark.parametrize("R", [33, 130])
def test_narrow_embedding_hip(R):
    """SDF MULTIRES 4 (27 embedding columns, K padded to 32) at DTU widths: the narrow-input kernels (cnr_sweep0.hip, cnr_narrow_bwd.hip) then run
    with one of their three k16 blocks empty -- its LDS columns must read as zeros whatever earlier kernels left there."""
    _check(R, None, torch.device("cuda:0"), "dtu_pe4")


_CHILD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_edge_child.py")
# Fallback forms of the default kernels, by what they replace.  BACKWARD-only switches leave the forward pass -- hence every ReLU decision --
# untouched: the gradients of the two forms differ by round-off only (summation order of the weight-gradient partial sums, which launch
# forms a product), never by a flipped unit.  FORWARD switches change the rounding of the forward values themselves: their gradients may
# differ by a flipped ReLU unit (up to 2e-2 of one row), so for them only the outputs are compared.
_BACKWARD_FALLBACKS = [("CNR_NO_SWEEP0",), ("CNR_NO_NARROW_BWD",), ("CNR_NO_FDW",), ("CNR_NO_HEAD_BWD", "CNR_NO_STRIP_BWD"), ("CNR_NO_TOP_FUSE",), ("CNR_FDW_SPLIT",)]
_FORWARD_FALLBACKS = [("CNR_NO_CHAIN_FWD", "CNR_NO_CHAIN_SDF"), ("CNR_NO_NARROW_DX",), ("CNR_NO_FUSED",)]


_RAGGED = [1, 3, 33, 130]


def _child_grads(Rs, cfg_name, switches, tmp_path):
    """{R: {name: array}} of one child process run under ``switches`` (one process renders every R: the start-up is what costs)."""
    import subprocess
    import sys
    import numpy as np
    tmpl = os.path.join(str(tmp_path), "g_%d_" + ("_".join(switches) or "default") + ".npz")
    env = {k: v for k, v in os.environ.items() if not k.startswith("CNR_")}
    env.update({k: "1" for k in switches})
    r = subprocess.run([sys.executable, _CHILD, ",".join(str(x) for x in Rs), cfg_name, tmpl], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return {R: dict(np.load(tmpl % R)) for R in Rs}


@pytest.mark.gpu
def test_ragged_batches_default_kernels_match_fallback_kernels_hip(tmp_path):
    """The tight gate at ragged sizes (1, 3, 33, 130 rays; the float64 gate above is loose below 64 rays because of ReLU kinks): the default
    path against the fallback forms of its kernels on the SAME rays, in child processes (the switches are read once per process).
    Backward-only fallbacks: outputs bit-identical, every gradient tensor within 2e-4 of its own largest entry -- the two forms are both
    float32-class evaluations of second-order terms (softplus'' = 100 sigma (1 - sigma) amplifies a last-bit difference; measured up to
    3.9e-5 at 3 rays on the SDF tensors, 4e-6 on the others), while a wrong last tile, row or partial-sum slot moves a tensor by percent.
    Forward fallbacks: every output within 1e-4 (measured 3.6e-5 on the per-sample weights at inv_s = 665)."""
    import numpy as np
    base_all = _child_grads(_RAGGED, "dtu", (), tmp_path)
    worst = {}
    for sw in _BACKWARD_FALLBACKS:
        alt_all = _child_grads(_RAGGED, "dtu", sw, tmp_path)
        for R in _RAGGED:
            base, alt = base_all[R], alt_all[R]
            assert set(alt) == set(base)
            for k in base:
                if k.startswith("out:"):
                    assert np.array_equal(alt[k], base[k]), (R, sw, k)
                    continue
                scale = float(np.abs(base[k]).max())
                if scale == 0.0:
                    assert float(np.abs(alt[k]).max()) == 0.0, (R, sw, k)
                    continue
                err = float(np.abs(alt[k].astype(np.float64) - base[k]).max()) / scale
                worst[(R,) + sw] = max(worst.get((R,) + sw, 0.0), err)
                assert np.isfinite(err) and err < 2e-4, (R, sw, k, err)
    for sw in _FORWARD_FALLBACKS:
        alt_all = _child_grads(_RAGGED, "dtu", sw, tmp_path)
        for R in _RAGGED:
            base, alt = base_all[R], alt_all[R]
            for k in base:
                if k.startswith("out:"):
                    scale = max(float(np.abs(base[k]).max()), 1e-30)
                    err = float(np.abs(alt[k].astype(np.float64) - base[k]).max()) / scale
                    assert np.isfinite(err) and err < 1e-4, (R, sw, k, err)
                else:
                    assert np.isfinite(alt[k]).all(), (R, sw, k)
    print("ragged batches, worst backward-fallback differences:", {" ".join(str(x) for x in k): "%.1e" % v for k, v in worst.items()})


def _check_eval_sizes(library, device, cfg_name, sizes):
    """sdf() / extract_color() at point counts that do not fill a tile, a chain group or a chunk: every prefix of the largest query gives the
    same values as the full query (rows are independent; the chain kernel picks its tile shape by the point count, so sdf agrees to
    round-off, 4e-7 measured, not to the bit), the full query matches the oracle, and an empty query returns empty arrays."""
    from oracle import colorneus_oracle as O
    ocfg = _config(cfg_name)
    P = O.init_params(ocfg, seed=0, trained_like=True)
    r = N.make_renderer(ocfg, P, library, device)
    g = torch.Generator().manual_seed(3)
    nmax = max(sizes)
    pts = (torch.rand(nmax, 3, generator=g) * 2.0 - 1.0)
    full_s = r.sdf(pts.to(device)).cpu()
    full_c = torch.from_numpy(r.extract_color(pts.numpy(), device))
    ref_s = O.sdf_value(P, ocfg.sdf, pts[:512])
    assert G.relerr(full_s[:512], ref_s.reshape(-1, 1)) < TOL
    for n in sizes:
        s = r.sdf(pts[:n].to(device)).cpu()
        c = torch.from_numpy(r.extract_color(pts[:n].numpy(), device))
        assert s.shape == (n, 1) and c.shape == (n, 3)
        if n:
            ds, dc = float((s - full_s[:n]).abs().max()), float((c - full_c[:n]).abs().max())
            assert ds < 1e-5 and dc < 1e-5, (n, ds, dc)


def test_eval_edge_sizes_emu():
    _check_eval_sizes(N.EMU_LIB, torch.device("cpu"), "tiny", [0, 1, 31, 33, 129, 600])


@pytest.mark.gpu
def test_eval_edge_sizes_hip():
    # 262144 = one evaluation chunk (cnr_plan.cpp kEvalChunk): one point more starts a second pass
    _check_eval_sizes(None, torch.device("cuda:0"), "dtu", [0, 1, 31, 33, 127, 129, 4097, 262144, 262145])
