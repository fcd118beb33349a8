"""Forward-only render (cnr_render_forward_only): the inference use of the path -- NeuS_Trainer.validate_image (NeuS_Trainer.py:236-245 reads
color_fine and depth of every EVAL_RAY_SIZE chunk) and evaluation.py.

  * values: every output BIT-IDENTICAL to the saving forward (cnr_render_forward) -- the value-producing launches are the same kernels on the
    same operands, only what would be stored for a backward pass is left out -- at the golden sample positions and through the sampler;
  * against the reference: color_fine / depth (and the other outputs) of the reference goldens at 1e-4, incl. the non-default call
    arguments (cos_anneal_ratio, background_rgb);
  * selection: under torch.no_grad() (or with nothing that requires grad) the module takes the forward-only entry point by itself; a call
    that can be differentiated never does;
  * the scratch buffer holds no state: results do not depend on its previous contents, and it is smaller than the training context;
  * early-termination compaction on this path (prune_eps > 0; the chain-fused colour / relight launch reads its rows through the ballot /
    popcount index list on the HIP build): geometry outputs untouched, kept samples' colour outputs identical, skipped ones zero, pixel error
    bounded by eps per skipped sample.
CPU: emulation build; GPU (-m gpu): HIP build."""
import ctypes as C
import os

import pytest
import torch

import _golden as G
import _native as N

KEYS = ["color_fine", "s_val", "cdf_fine", "weight_sum", "weight_max", "gradients", "weights", "gradient_error", "inside_sphere", "depth",
        "global_color", "delta_relight", "z_vals", "eik_sums"]


def _inputs(name, device, tag="jit"):
    fx = G.load(name)
    ocfg, P = G.weights_of(name, fx)
    t = lambda k: torch.from_numpy(fx[k]).to(device)
    return fx, ocfg, P, t("rays_o"), t("rays_d"), t(f"{tag}:near"), t(f"{tag}:far"), t(f"{tag}:z_vals")


def _check_values(library, device, name):
    fx, ocfg, P, o, d, near, far, z = _inputs(name, device)
    r = N.make_renderer(ocfg, P, library, device)
    kw = G.call_kwargs(fx, device)
    with torch.no_grad():
        sav = r(o, d, near, far, z_vals=z, forward_only=False, **kw)
        fwd = r(o, d, near, far, z_vals=z, forward_only=True, **kw)
        auto = r(o, d, near, far, z_vals=z, **kw)
    for k in KEYS:
        if k in sav:
            assert torch.equal(sav[k], fwd[k]), (name, k, float((sav[k] - fwd[k]).abs().max()))
            assert torch.equal(fwd[k], auto[k]), (name, k)
    assert set(fwd) == set(sav)
    # against the reference's goldens (the inference consumers read color_fine and depth)
    for k in G.OUTPUT_KEYS:
        if f"jit:out_{k}" in fx:
            ref = fx[f"jit:out_{k}"]
            assert G.relerr(fwd[k].cpu().reshape(ref.shape), ref) < 1e-4, (name, k)
    # through the sampler as well (same jitter draw for both forms)
    t_rand = torch.from_numpy(fx["jit:t_rand"])
    outs = []
    orig = torch.rand
    try:
        torch.rand = lambda *a, **k: t_rand.clone()
        with torch.no_grad():
            for fo in (False, True):
                outs.append(r(o, d, near, far, forward_only=fo, **kw))
    finally:
        torch.rand = orig
    for k in KEYS:
        if k in outs[0]:
            assert torch.equal(outs[0][k], outs[1][k]), (name, "sampler", k)


def _check_selection_and_scratch(library, device, name):
    import color_neus_amd as cn
    fx, ocfg, P, o, d, near, far, z = _inputs(name, device)
    r = N.make_renderer(ocfg, P, library, device)
    lib = r._lib
    calls = []
    real_fo, real_fw = lib.lib.cnr_render_forward_only, lib.lib.cnr_render_forward

    class Spy:
        def __init__(self, fn, tag):
            self.fn, self.tag = fn, tag

        def __call__(self, *a):
            calls.append(self.tag)
            return self.fn(*a)
    lib.lib.cnr_render_forward_only, lib.lib.cnr_render_forward = Spy(real_fo, "only"), Spy(real_fw, "saving")
    try:
        with torch.no_grad():
            r(o, d, near, far, z_vals=z)
        assert calls == ["only"], calls
        calls.clear()
        for p in r.parameters():
            p.requires_grad_(False)
        r(o, d, near, far, z_vals=z)                               # grad mode on, but nothing requires grad
        assert calls == ["only"], calls
        calls.clear()
        for p in r.parameters():
            p.requires_grad_(True)
        out = r(o, d, near, far, z_vals=z)                         # a differentiable call keeps its context
        assert calls == ["saving"], calls
        loss, _ = cn.compute_loss(out, torch.from_numpy(fx["rgb_gt"]).to(device), torch.from_numpy(fx["mask"]).to(device))
        loss.backward()
        assert all(p.grad is not None for p in r.parameters())
        calls.clear()
        for p in r.parameters():
            p.requires_grad_(False)
        o2 = o.clone().requires_grad_(True)
        r(o2, d, near, far, z_vals=z)                              # d rays requested: differentiable as well
        assert calls == ["saving"], calls
    finally:
        lib.lib.cnr_render_forward_only, lib.lib.cnr_render_forward = real_fo, real_fw
    R = o.shape[0]
    nb_inf = lib.lib.cnr_infer_scratch_bytes(C.byref(r._ccfg), R)
    nb_ctx = lib.lib.cnr_ctx_bytes(C.byref(r._ccfg), R)
    assert 0 < nb_inf < nb_ctx, (nb_inf, nb_ctx)
    # no state in the scratch: poisoned contents change nothing
    with torch.no_grad():
        ref = r(o, d, near, far, z_vals=z, forward_only=True)
        orig_empty = torch.empty

        def poisoned_empty(*a, **k):
            t = orig_empty(*a, **k)
            if t.dtype == torch.uint8 and t.numel() > 4096:
                t.fill_(0xFF)
            return t
        try:
            torch.empty = poisoned_empty
            got = r(o, d, near, far, z_vals=z, forward_only=True)
            got_p = r(o, d, near, far, z_vals=z, forward_only=True, prune_eps=1e-3)
        finally:
            torch.empty = orig_empty
        ref_p = r(o, d, near, far, z_vals=z, forward_only=True, prune_eps=1e-3)
    for k in KEYS:
        if k in ref:
            assert torch.equal(ref[k], got[k]), k
            assert torch.equal(ref_p[k], got_p[k]), ("pruned", k)
    # a buffer that is too small is refused, not overrun
    real_sz = lib.lib.cnr_infer_scratch_bytes

    class Half:
        def __call__(self, *a):
            return real_sz(*a) // 2
    lib.lib.cnr_infer_scratch_bytes = Half()
    try:
        with torch.no_grad(), pytest.raises(RuntimeError, match="too small"):
            r(o, d, near, far, z_vals=z, forward_only=True)
    finally:
        lib.lib.cnr_infer_scratch_bytes = real_sz


def _check_prune(library, device, name):
    fx, ocfg, P, o, d, near, far, z = _inputs(name, device)
    r = N.make_renderer(ocfg, P, library, device)
    eps = 1e-3
    with torch.no_grad():
        full = r(o, d, near, far, z_vals=z, forward_only=True)
        pr = r(o, d, near, far, z_vals=z, forward_only=True, prune_eps=eps)
        pr_sav = r(o, d, near, far, z_vals=z, forward_only=False, prune_eps=eps)     # the compact-copy form of the saving layout
    keep = full["weights"] >= eps
    assert 0 < int(keep.sum()) < keep.numel()
    for k in ("weights", "depth", "weight_sum", "weight_max", "cdf_fine", "gradients", "gradient_error", "s_val", "inside_sphere"):
        assert torch.equal(full[k], pr[k]), k
    n_pruned = (~keep).sum(-1, keepdim=True).float()
    assert bool(((full["color_fine"] - pr["color_fine"]).abs() <= n_pruned * eps + 1e-6).all())
    # ... and therefore against the REFERENCE's pixel (golden color_fine at these sample positions): eps per skipped sample on top of the 1e-4 gate
    ref = torch.from_numpy(fx["jit:out_color_fine"]).to(pr["color_fine"].device)
    assert bool(((pr["color_fine"] - ref).abs() <= n_pruned * eps + 1e-4 * float(ref.abs().max())).all())
    if "delta_relight" in full:
        assert torch.equal(pr["delta_relight"][keep], full["delta_relight"][keep])       # same rows through the same arithmetic
        assert float(pr["delta_relight"][~keep].abs().max()) == 0.0
    # both compaction forms (index list / compact copies) composite the same kept samples
    assert float((pr["color_fine"] - pr_sav["color_fine"]).abs().max()) < 1e-6
    # eps so large that nothing is kept, and so small that everything is
    with torch.no_grad():
        none = r(o, d, near, far, z_vals=z, forward_only=True, prune_eps=2.0)
        all_ = r(o, d, near, far, z_vals=z, forward_only=True, prune_eps=1e-30)
    bg = 0.0
    assert float(none["color_fine"].abs().max()) == bg
    assert float((all_["color_fine"] - full["color_fine"]).abs().max()) < 1e-6


emu = pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")


@emu
@pytest.mark.parametrize("name", ["tiny_sharp", "tiny_neus_sharp", "tiny_sharp_anneal", "dtu_sharp"])
def test_forward_only_values_emu(name):
    _check_values(N.EMU_LIB, "cpu", name)


@emu
def test_forward_only_selection_and_scratch_emu():
    _check_selection_and_scratch(N.EMU_LIB, "cpu", "tiny_sharp")


@emu
@pytest.mark.parametrize("name", ["tiny_sharp", "tiny_neus_sharp"])
def test_forward_only_prune_emu(name):
    _check_prune(N.EMU_LIB, "cpu", name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_sharp", "tiny_neus_sharp", "dtu_sharp", "dtu_init", "neus_dtu_sharp", "dtu_sharp_anneal", "dtu_noimp_sharp"])
def test_forward_only_values_hip(name):
    _check_values(None, "cuda:0", name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_sharp", "dtu_sharp"])
def test_forward_only_selection_and_scratch_hip(name):
    _check_selection_and_scratch(None, "cuda:0", name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_sharp", "tiny_neus_sharp", "dtu_sharp", "neus_dtu_sharp"])
def test_forward_only_prune_hip(name):
    _check_prune(None, "cuda:0", name)


@pytest.mark.gpu
def test_forward_only_full_size_matches_saving_forward_hip():
    """8192 rays x 128 samples (one EVAL chunk of the bench's inference leg) on the DTU renderer block, through the sampler: the forward-only call and
    the saving forward agree to the bit in every output; the pruned call keeps the geometry outputs and stays within eps per skipped sample."""
    import color_neus_amd as cn
    from color_neus_amd import synthetic
    dev = "cuda:0"
    cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
    torch.manual_seed(0)
    r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
    o, d, near, far, _, _ = synthetic.synthetic_view(seed=3, device=dev)
    sel = torch.randperm(o.shape[0], generator=torch.Generator().manual_seed(5))[:8192].to(dev)
    o, d, near, far = o[sel], d[sel], near[sel], far[sel]
    with torch.no_grad():
        a = r(o, d, near, far, perturb_overwrite=0, forward_only=False)
        b = r(o, d, near, far, perturb_overwrite=0, forward_only=True)
        c = r(o, d, near, far, perturb_overwrite=0, forward_only=True, prune_eps=1e-4)
    for k in KEYS:
        assert torch.equal(a[k], b[k]), k
    for k in ("weights", "depth", "weight_sum", "gradients", "z_vals"):
        assert torch.equal(b[k], c[k]), k
    keep = b["weights"] >= 1e-4
    assert bool(((b["color_fine"] - c["color_fine"]).abs() <= (~keep).sum(-1, keepdim=True).float() * 1e-4 + 1e-6).all())
    assert torch.equal(c["delta_relight"][keep], b["delta_relight"][keep])


@pytest.mark.gpu
def test_a_ray_renders_the_same_in_every_chunk_size_hip():
    """validate_image renders a view in EVAL_RAY_SIZE chunks (NeuS_Trainer.py:233-245): the pixel of a ray must not depend on how many rays share
    its chunk.  The library picks tile shapes by point count -- the sampler's value chain runs <1, 2>, <2, 2> or <4, 1> tiles below / above 8192 and
    32768 points per launch -- and every shape must form every point's values in the same order (round 6: <2, 2> used to add the halves of the sdf dot
    product in another order than <4, 1>: one ulp of sdf, a 1.5e-7 colour difference on 215 of 640 000 pixels between 8192- and 65536-ray chunks).
    4096 rays in one call against the same rays in chunks of 1024, 256 and 96 (ragged): every output of every ray bit-identical."""
    import color_neus_amd as cn
    from color_neus_amd import synthetic
    dev = torch.device("cuda:0")
    cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
    torch.manual_seed(0)
    r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
    views = synthetic.synthetic_view(seed=1, device=dev)
    sel = torch.randperm(views[0].shape[0], generator=torch.Generator().manual_seed(3))[:4096].to(dev)
    o, d, n, f = [x[sel] for x in views[:4]]
    per_ray = ["color_fine", "depth", "weight_sum", "weight_max", "weights", "cdf_fine", "gradients", "z_vals", "global_color", "delta_relight", "inside_sphere"]
    with torch.no_grad():
        whole = r(o, d, n, f, perturb_overwrite=0)
        for chunk in (1024, 256, 96):
            parts = [r(o[a:a + chunk], d[a:a + chunk], n[a:a + chunk], f[a:a + chunk], perturb_overwrite=0) for a in range(0, 4096, chunk)]
            for k in per_ray:
                got = torch.cat([p[k].reshape(p[k].shape[0], -1) for p in parts], 0)
                ref = whole[k].reshape(4096, -1)
                assert torch.equal(got, ref), (chunk, k, int((got != ref).any(dim=1).sum()), float((got - ref).abs().max()))
