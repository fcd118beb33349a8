"""The fused training loss (cnr_loss_sums / cnr_loss_grads) against compute_loss -- the build's restatement of
NeuS_Trainer.compute_loss that the golden loss vectors pin (test_oracle_golden.py) -- values and gradients w.r.t. every renderer output."""
import os

import pytest
import torch

import _native as N

VARIANTS = [dict(), dict(rgb_loss_type="l1"), dict(include_mask=False), dict(lambda_mask=0.0), dict(lambda_relight=0.0)]


def _case(device, R=53, M=12, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = {"color_fine": torch.rand(R, 3, generator=g), "weight_sum": torch.rand(R, 1, generator=g) * 1.2 - 0.1,   # some outside the clip range
           "gradient_error": torch.tensor(0.3), "delta_relight": torch.randn(R, M, 3, generator=g) * 0.1}
    out = {k: v.to(device).requires_grad_(True) for k, v in out.items()}
    gt = torch.rand(R, 3, generator=g).to(device)
    mask = (torch.rand(R, generator=g) > 0.4).float().to(device)
    return out, gt, mask


def _compare(lib, device):
    import color_neus_amd as cn
    for kw in VARIANTS:
        for use_mask in (True, False):
            out, gt, mask = _case(device)
            m = mask if use_mask else None
            l1, d1 = cn.compute_loss(out, gt, m, **kw)
            g1 = torch.autograd.grad(l1, list(out.values()), allow_unused=True)
            l2, d2 = cn.compute_loss_fused(out, gt, m, library=lib, **kw)
            g2 = torch.autograd.grad(l2 * 1.0, list(out.values()), allow_unused=True)
            assert abs(float(l1) - float(l2)) < 1e-6 * max(1.0, abs(float(l1))), (kw, use_mask)
            for k in d1:
                assert abs(float(d1[k]) - float(d2[k])) < 1e-6 * max(1.0, abs(float(d1[k]))), (kw, use_mask, k)
            for name, a, b in zip(out.keys(), g1, g2):
                za = torch.zeros_like(out[name]) if a is None else a
                zb = torch.zeros_like(out[name]) if b is None else b
                assert float((za - zb).abs().max()) <= 1e-6 * max(float(za.abs().max()), 1e-6), (kw, use_mask, name)


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_fused_loss_matches_compute_loss_cpu_emulation():
    import color_neus_amd as cn
    _compare(cn.load_library(N.EMU_LIB), "cpu")


@pytest.mark.gpu
def test_fused_loss_matches_compute_loss_hip():
    import color_neus_amd as cn
    _compare(cn.load_library(), "cuda:0")


@pytest.mark.gpu
def test_fused_loss_end_to_end_gradients_match():
    """A whole training step with the fused loss gives the parameter gradients of the step with the torch loss."""
    import color_neus_amd as cn
    import _golden as G
    res = {}
    for fused in (False, True):
        fx = G.load("dtu_sharp")
        ocfg, P = G.weights_of("dtu_sharp", fx)
        r = N.make_renderer(ocfg, P, None, "cuda:0")
        t = lambda k: torch.from_numpy(fx[k]).to("cuda:0")
        out = r(t("rays_o"), t("rays_d"), t("det:near"), t("det:far"), z_vals=t("det:z_vals"))
        fn = cn.compute_loss_fused if fused else cn.compute_loss
        loss, _ = fn(out, t("rgb_gt"), t("mask"))
        loss.backward()
        res[fused] = (float(loss), {k: p.grad.clone() for k, p in r.named_parameters()})
    assert abs(res[0][0] - res[1][0]) < 1e-6 * abs(res[0][0])
    for k, g in res[0][1].items():
        assert float((g - res[1][1][k]).abs().max()) <= 1e-4 * float(g.abs().max()), k   # own scale per tensor


def _loss_goldens(device, library):
    """compute_loss_fused directly against the loss:* vectors captured from the reference's objective (NeuS_Trainer.py:129-171),
    mask on and off, values and the gradients of every input."""
    import _golden as G
    import color_neus_amd as cn
    fx = G.load("functions")
    t = lambda k: torch.from_numpy(fx["loss:" + k]).to(device)
    for with_mask, key in ((True, "l_on"), (False, "l_off")):
        leaves = {k: t(k).clone().requires_grad_(True) for k in ("cf", "ws", "dr", "ge")}
        rd = {"color_fine": leaves["cf"], "weight_sum": leaves["ws"], "delta_relight": leaves["dr"], "gradient_error": leaves["ge"]}
        mask = t("m") if with_mask else None
        loss, parts = cn.compute_loss_fused(rd, t("gt"), mask, library=library)
        assert abs(float(loss.detach()) - float(fx["loss:" + key])) < 1e-6 * abs(float(fx["loss:" + key])), key
        loss.backward()
        ref = {k: t(k).clone().requires_grad_(True) for k in ("cf", "ws", "dr", "ge")}
        rl, _ = cn.compute_loss({"color_fine": ref["cf"], "weight_sum": ref["ws"], "delta_relight": ref["dr"], "gradient_error": ref["ge"]}, t("gt"), mask)
        rl.backward()
        for k in leaves:
            if ref[k].grad is None:
                continue
            assert leaves[k].grad is not None, k
            assert float((leaves[k].grad - ref[k].grad).abs().max()) <= 1e-6 * max(float(ref[k].grad.abs().max()), 1e-12), (key, k)


@pytest.mark.gpu
def test_fused_loss_against_reference_goldens_hip():
    _loss_goldens("cuda:0", None)


def test_fused_loss_against_reference_goldens_emu():
    import os
    if not os.path.isfile(N.EMU_LIB):
        pytest.skip("emulation library not built")
    import color_neus_amd as cn
    _loss_goldens("cpu", cn.load_library(N.EMU_LIB))


def _loss_only(library, device, name="tiny_sharp"):
    """training_outputs="loss_only" (no [R][M][3] dict tensors, relight term from per-ray sums) gives the loss and the
    gradients of the default dict mode (NeuS.py:387-408 return dict + compute_loss)."""
    import _golden as G
    import color_neus_amd as cn
    fx = G.load(name)
    ocfg, P = G.weights_of(name, fx)
    lib = cn.load_library(library)
    res = {}
    for mode in ("dict", "loss_only"):
        r = N.make_renderer(ocfg, P, library, device)
        t = lambda k: torch.from_numpy(fx[k]).to(device)
        o, d = t("rays_o").requires_grad_(True), t("rays_d").requires_grad_(True)
        out = r(o, d, t("jit:near"), t("jit:far"), z_vals=t("jit:z_vals"), training_outputs=mode)
        if mode == "loss_only":
            assert "gradients" not in out and "delta_relight" not in out and out["delta_relight_ray_sum"].shape == (o.shape[0],)
        else:
            want = out["delta_relight"].detach().sum(dim=(1, 2))
        loss, parts = cn.compute_loss_fused(out, t("rgb_gt"), t("mask"), library=lib)
        loss.backward()
        res[mode] = (out, float(loss.detach()), {k: p.grad.detach().cpu().clone() for k, p in r.named_parameters()}, o.grad.cpu().clone(), d.grad.cpu().clone(),
                     {k: float(v.detach()) for k, v in parts.items()})
    a, b = res["dict"], res["loss_only"]
    got = b[0]["delta_relight_ray_sum"].detach()
    assert float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))
    assert abs(a[1] - b[1]) <= 2e-6 * abs(a[1])
    for k in a[5]:
        assert abs(a[5][k] - b[5][k]) <= 2e-6 * max(abs(a[5][k]), 1e-6), k
    for k in a[2]:   # the per-ray relight sum is folded in another order than the [R][M][3] sum: the seed of the relight branch moves by ~1e-6 relative
        den = float(a[2][k].abs().max())
        assert float((a[2][k] - b[2][k]).abs().max()) <= 1e-5 * den + 1e-12, k
    for x, y in ((a[3], b[3]), (a[4], b[4])):
        assert float((x - y).abs().max()) <= 1e-5 * float(x.abs().max())
    for k in ("color_fine", "weights", "weight_sum", "gradient_error", "depth", "global_color"):
        assert torch.equal(a[0][k].detach().cpu(), b[0][k].detach().cpu()), k


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_loss_only_training_outputs_emu():
    _loss_only(N.EMU_LIB, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny_sharp", "dtu_sharp"])
def test_loss_only_training_outputs_hip(name):
    _loss_only(None, "cuda:0", name)


def _one_launch_forms(library, device):
    """cnr_loss_forward / cnr_loss_backward (one launch each) against the entry points they replace (cnr_loss_sums[_ray] + cnr_loss_combine,
    cnr_loss_coef + cnr_loss_grads) on the same inputs: the same partial sums folded in the same order and the same scalar arithmetic, so every
    output must agree to the bit -- for both forms of the relight term, with and without a mask, several times in a row (the completion
    counter of the forward launch has to be back at zero after each call)."""
    import ctypes as C
    import color_neus_amd as cn
    from color_neus_amd._lib import CnrLossConfig
    lib = cn.load_library(library)
    L = lib.lib
    g = torch.Generator().manual_seed(7)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else C.c_void_p(0)
    for R, M, use_mask, per_ray in ((4096, 128, True, False), (1000, 16, True, True), (53, 12, False, False), (1, 8, True, True)):
        color = torch.rand(R, 3, generator=g).to(device); gt = torch.rand(R, 3, generator=g).to(device)
        wsum = torch.rand(R, generator=g).to(device); mask = (torch.rand(R, generator=g) > 0.3).float().to(device) if use_mask else None
        drel = (torch.randn(R, generator=g) if per_ray else torch.randn(R, M, 3, generator=g) * 0.1).to(device)
        gerr = torch.rand(1, generator=g).to(device); gl = torch.tensor([0.7], device=device)
        cfg = CnrLossConfig(1.0, 0.1, 0.1, 1.0, 0, 1)
        nb = L.cnr_loss_scratch_bytes(R)
        # the scratch contract of the one-launch forms: zeroed ONCE (its tail is the completion counter), then reused call after call
        scr = torch.zeros(nb, dtype=torch.uint8, device=device)
        for rep in range(3):
            s1, o1 = torch.empty(4, device=device), torch.empty(6, device=device)
            fn = L.cnr_loss_sums_ray if per_ray else L.cnr_loss_sums
            lib.check(fn(C.byref(cfg), p(color), p(wsum), p(drel), p(gt), p(mask), R, M, p(s1), p(scr), nb, st), "sums")
            lib.check(L.cnr_loss_combine(C.byref(cfg), p(s1), p(gerr), float(R), M, int(use_mask), 1, p(o1), st), "combine")
            s2, o2 = torch.empty(4, device=device), torch.empty(6, device=device)
            lib.check(L.cnr_loss_forward(C.byref(cfg), p(color), p(wsum), p(drel), int(per_ray), p(gt), p(mask), p(gerr), R, M, float(R), int(use_mask), 1,
                                         p(s2), p(o2), p(scr), nb, st), "forward")
            assert torch.equal(s1, s2) and torch.equal(o1, o2), (R, M, rep, s1, s2, o1, o2)
            c1, dc1, dw1 = torch.empty(4, device=device), torch.empty(R, 3, device=device), torch.empty(R, device=device)
            lib.check(L.cnr_loss_coef(C.byref(cfg), p(gl), p(o1[5:6].contiguous()), float(R), M, int(use_mask), 1, p(c1), st), "coef")
            lib.check(L.cnr_loss_grads(C.byref(cfg), p(color), p(wsum), p(gt), p(mask), R, M, p(c1), p(dc1), p(dw1), p(None), st), "grads")
            c2, dc2, dw2 = torch.empty(4, device=device), torch.empty(R, 3, device=device), torch.empty(R, device=device)
            dr2 = torch.empty(R, device=device)
            lib.check(L.cnr_loss_backward(C.byref(cfg), p(color), p(wsum), p(gt), p(mask), R, M, p(gl), p(o2[5:6].contiguous()), p(None), float(R), int(use_mask), 1,
                                          p(c2), p(dc2), p(dw2), p(dr2), st), "backward")
            assert torch.equal(c1, c2) and torch.equal(dc1, dc2) and torch.equal(dw1, dw2), (R, M, rep)
            assert torch.equal(dr2, c2[2] * mask if use_mask else c2[2].expand(R))
            assert int(scr[-16:].view(torch.int32)[0]) == 0      # the counter is back at zero
            # ---- the ray-sharded forms on the same rays: stats = {the three sums, eik_sums, eik_sums[1]}; with nothing added by other ranks the
            # combine step must reproduce the single-process scalars for gradient_error = num / (den + 1e-5), and eik_factor = 1
            eik = torch.tensor([0.37, 123.0], device=device)
            st8, o8 = torch.empty(8, device=device), torch.empty(8, device=device)
            lib.check(L.cnr_loss_shard_stats(C.byref(cfg), p(color), p(wsum), p(drel), int(per_ray), p(gt), p(mask), p(eik), R, M, p(st8), p(scr), nb, st), "stats")
            assert torch.equal(st8[:3], s1[:3]) and torch.equal(st8[3:6], torch.stack([eik[0], eik[1], eik[1]]))
            lib.check(L.cnr_loss_shard_combine(C.byref(cfg), p(st8), float(R), M, int(use_mask), 1, p(o8), st), "shard_combine")
            g1 = (eik[0] / (eik[1] + 1e-5)).reshape(1)
            o3 = torch.empty(6, device=device)
            lib.check(L.cnr_loss_combine(C.byref(cfg), p(s1), p(g1), float(R), M, int(use_mask), 1, p(o3), st), "combine")
            assert torch.equal(o8[:6], o3) and float(o8[6]) == 1.0
            # a second rank's statistics added (what the all-reduce does): global ratio and this rank's factor
            st8b = st8.clone()
            st8b[:5] += torch.tensor([1.5, 2.5, -0.25, 0.11, 77.0], device=device)
            lib.check(L.cnr_loss_shard_combine(C.byref(cfg), p(st8b), float(2 * R), M, int(use_mask), 1, p(o8), st), "shard_combine")
            assert abs(float(o8[2]) - float(st8b[3] / (st8b[4] + 1e-5))) < 1e-7 and abs(float(o8[6]) - float((eik[1] + 1e-5) / (st8b[4] + 1e-5))) < 1e-7
            c3 = torch.empty(4, device=device)
            lib.check(L.cnr_loss_backward(C.byref(cfg), p(color), p(wsum), p(gt), p(mask), R, M, p(gl), p(o8[5:6].contiguous()), p(o8[6:7].contiguous()), float(2 * R),
                                          int(use_mask), 1, p(c3), p(dc2), p(dw2), p(None), st), "backward")
            assert abs(float(c3[3]) - 0.7 * 0.1 * float(o8[6])) < 1e-7


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_one_launch_loss_forms_emu():
    _one_launch_forms(N.EMU_LIB, torch.device("cpu"))


@pytest.mark.gpu
def test_one_launch_loss_forms_hip():
    _one_launch_forms(None, torch.device("cuda:0"))
