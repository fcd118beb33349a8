"""Fused per-parameter clip + Adam (cnr_clip_adam_step) against the reference loop's two calls: clip_grad_norm_ on each parameter
tensor (lib/utils/net_utils.py:174-184) followed by torch.optim.Adam(betas=(0.9, 0.99), eps=1e-8) (net_utils.py:88)."""
import os

import pytest
import torch

import _golden as G
import _native as N
import color_neus_amd as cn


def _run(library, device, steps=5, max_norm=0.05):
    g = torch.Generator().manual_seed(3)
    shapes = [(257, 256), (256,), (256, 1), (1,), (3, 259), (217, 39), (65536 // 64, 70)]
    ref = [torch.randn(s, generator=g).to(device).requires_grad_(True) for s in shapes]
    ours = [p.detach().clone().requires_grad_(True) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=5e-4, betas=(0.9, 0.99), eps=1e-8)
    o_our = cn.ClipAdam(ours, lr=5e-4, betas=(0.9, 0.99), eps=1e-8, max_norm=max_norm, library=library)
    for it in range(steps):
        scale = [1e-3, 1.0, 30.0, 1e-6, 0.2][it % 5]          # below / above the clip threshold, tiny gradients
        for p, q in zip(ref, ours):
            gr = (torch.randn(p.shape, generator=g) * scale).to(device)
            if it == 2:
                gr = gr * (torch.rand(p.shape, generator=g) < 0.1).to(device)   # sparse gradient
            p.grad = gr.clone()
            q.grad = gr.clone()
        o_ref.param_groups[0]["lr"] = o_our.param_groups[0]["lr"] = 5e-4 * (0.9 ** it)   # a scheduler changes lr every step
        for p in ref:
            torch.nn.utils.clip_grad_norm_(p, max_norm, 2)
        o_ref.step()
        o_our.step()
    for p, q in zip(ref, ours):
        assert float((p.detach() - q.detach()).abs().max()) <= 2e-6 * max(1.0, float(p.detach().abs().max())), p.shape
    st = o_our.state[ours[0]]
    flat_m = torch.cat([o_ref.state[p]["exp_avg"].reshape(-1) for p in ref])
    flat_v = torch.cat([o_ref.state[p]["exp_avg_sq"].reshape(-1) for p in ref])
    assert float((st["exp_avg"] - flat_m).abs().max()) <= 1e-6 * float(flat_m.abs().max())
    assert float((st["exp_avg_sq"] - flat_v).abs().max()) <= 5e-6 * float(flat_v.abs().max())   # (the clip coefficient enters squared)
    assert st["steps"] == [steps] * len(ours)


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
@pytest.mark.parametrize("max_norm", [0.05, None])
def test_clip_adam_matches_torch_emu(max_norm):
    if max_norm is None:
        g = torch.Generator().manual_seed(1)
        p = torch.randn(300, generator=g).requires_grad_(True)
        q = p.detach().clone().requires_grad_(True)
        a, b = torch.optim.Adam([p], lr=1e-3, betas=(0.9, 0.99)), cn.ClipAdam([q], lr=1e-3, library=N.EMU_LIB)
        for _ in range(3):
            gr = torch.randn(300, generator=g)
            p.grad, q.grad = gr.clone(), gr.clone()
            a.step(); b.step()
        assert float((p - q).abs().max()) < 1e-6
    else:
        _run(N.EMU_LIB, "cpu", max_norm=max_norm)


@pytest.mark.gpu
def test_clip_adam_matches_torch_hip():
    _run(None, "cuda:0")


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_backward_lays_gradients_out_in_one_flat_buffer():
    """All parameter gradients of a render backward tile one contiguous buffer in canonical order (what the in-place bucket
    all-reduce and the fused optimiser step rely on), and a training step through ClipAdam moves the parameters."""
    from color_neus_amd import optim
    fx, r, out, loss, grads, o, d = N.run_native("tiny_sharp", "jit", N.EMU_LIB, "cpu", fixed_z=True)
    params = r._ordered_params()
    flat = optim.flat_view_of_grads(params)
    assert flat is not None and flat.numel() == sum(p.numel() for p in params)
    off = 0
    for p in params:
        assert torch.equal(flat[off:off + p.numel()].view_as(p), p.grad)
        off += p.numel()
    before = [p.detach().clone() for p in params]
    opt = cn.ClipAdam(params, lr=1e-3, max_norm=1.0, library=N.EMU_LIB)
    opt.step()
    assert any(float((a - b.detach()).abs().max()) > 0 for a, b in zip(before, params))


def _run_intermittent(library, device):
    """A parameter that receives a gradient only on some steps (auxiliary / background parameters) keeps its moments and its own step
    count, exactly like torch.optim.Adam's per-parameter state; the others are unaffected by its absence."""
    g = torch.Generator().manual_seed(5)
    shapes = [(64, 33), (64,), (5, 7), (129,)]
    ref = [torch.randn(s, generator=g).to(device).requires_grad_(True) for s in shapes]
    ours = [p.detach().clone().requires_grad_(True) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=1e-2, betas=(0.9, 0.99), eps=1e-8)
    o_our = cn.ClipAdam(ours, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, max_norm=None, library=library)
    for it in range(6):
        for i, (p, q) in enumerate(zip(ref, ours)):
            if (i == 2 and it % 2 == 1) or (i == 0 and it == 3):   # no gradient for these tensors on these steps (incl. the first one of the group)
                p.grad = None
                q.grad = None
                continue
            gr = torch.randn(p.shape, generator=g).to(device)
            p.grad = gr.clone()
            q.grad = gr.clone()
        o_ref.step()
        o_our.step()
    for p, q in zip(ref, ours):
        assert float((p.detach() - q.detach()).abs().max()) <= 2e-6 * max(1.0, float(p.detach().abs().max())), p.shape
    assert o_our.state[ours[0]]["steps"] == [5, 6, 3, 6]
    sd = o_our.state_dict()            # round trip: the scratch buffer is not part of the state
    assert set(sd["state"][0].keys()) == {"exp_avg", "exp_avg_sq", "steps"}
    o2 = cn.ClipAdam(ours, lr=1e-2, betas=(0.9, 0.99), eps=1e-8, max_norm=None, library=library)
    o2.load_state_dict(sd)
    assert o2.state[ours[0]]["steps"] == [5, 6, 3, 6]


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_clip_adam_intermittent_gradients_emu():
    _run_intermittent(N.EMU_LIB, "cpu")


@pytest.mark.gpu
def test_clip_adam_intermittent_gradients_hip():
    _run_intermittent(None, "cuda:0")


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_clip_adam_loads_state_of_the_previous_layout():
    """A state_dict saved by the build whose group state held ONE ``step`` count (and no ``steps`` list) resumes: the moments cover every
    parameter of the group there, so the per-parameter step counts are that count; a layout whose moments cover only a subset is refused
    with a clear message instead of a KeyError."""
    g = torch.Generator().manual_seed(4)
    ps = [torch.randn(40, 7, generator=g).requires_grad_(True), torch.randn(9, generator=g).requires_grad_(True)]
    a = cn.ClipAdam(ps, lr=1e-3, max_norm=1.0, library=N.EMU_LIB)
    for p in ps:
        p.grad = torch.randn(p.shape, generator=g)
    a.step(); a.step()
    sd = a.state_dict()
    st = sd["state"][0]
    old = {"exp_avg": st["exp_avg"].clone(), "exp_avg_sq": st["exp_avg_sq"].clone(), "step": 2}   # the earlier layout
    qs = [p.detach().clone().requires_grad_(True) for p in ps]
    b = cn.ClipAdam(qs, lr=1e-3, max_norm=1.0, library=N.EMU_LIB)
    b.load_state_dict({"state": {0: old}, "param_groups": sd["param_groups"]})
    for p, q in zip(ps, qs):
        gr = torch.randn(p.shape, generator=g)
        p.grad, q.grad = gr.clone(), gr.clone()
    a.step(); b.step()
    for p, q in zip(ps, qs):
        assert torch.equal(p.detach(), q.detach())
    assert b.state[qs[0]]["steps"] == [3, 3]
    c = cn.ClipAdam([q.detach().clone().requires_grad_(True) for q in qs], lr=1e-3, library=N.EMU_LIB)
    c.load_state_dict({"state": {0: {"exp_avg": old["exp_avg"][:280].clone(), "exp_avg_sq": old["exp_avg_sq"][:280].clone(), "step": 2}},
                       "param_groups": sd["param_groups"]})
    for q in c.param_groups[0]["params"]:
        q.grad = torch.zeros_like(q)
    with pytest.raises(RuntimeError, match="earlier layout"):
        c.step()


@pytest.mark.skipif(not os.path.isfile(N.EMU_LIB), reason="emulation library not built")
def test_capturable_mode_reads_the_step_scalars_from_device_memory_emu():
    """ClipAdam(capturable=True) -- cnr_adam_config.hyper_dev (ABI 8): lr and the two bias corrections read from three device floats that
    prepare_step() refreshes, so that step() can sit in a captured HIP graph -- must produce the same bits as the by-value form, with a
    scheduler changing lr every step; step() without prepare_step() is an error."""
    g = torch.Generator().manual_seed(5)
    shapes = [(33, 7), (5,), (4100,)]
    w0 = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) * sc for s in shapes] for sc in (1e-3, 1.0, 30.0, 0.2, 1e-6, 2.0)]

    def run(cap):
        ps = [w.clone().requires_grad_(True) for w in w0]
        opt = cn.ClipAdam(ps, lr=5e-4, betas=(0.9, 0.99), eps=1e-8, max_norm=0.05, library=N.EMU_LIB, capturable=cap)
        for it, gs in enumerate(grads):
            opt.param_groups[0]["lr"] = 5e-4 * (0.9 ** it)
            for p, gr in zip(ps, gs):
                p.grad = gr.clone()
            if cap:
                opt.prepare_step()
            opt.step()
        return [p.detach() for p in ps], opt.state[ps[0]]

    (a, sa), (b, sb) = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]) and sa["steps"] == sb["steps"]
    p = torch.zeros(3, requires_grad=True)
    p.grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="prepare_step"):
        cn.ClipAdam([p], library=N.EMU_LIB, capturable=True).step()
