"""GPU: a whole training step (ray generation -> renderer forward -> fused loss -> backward -> clip + Adam) replayed as ONE HIP graph
(color-neus_amd/graph.py) against the same steps enqueued launch by launch: same pixels, same CPU-generator jitter draws, same optimiser
schedule -- the parameters after 5 steps (2 warm-up calls inside the capture helper + 3 replays) must be BIT-IDENTICAL, and so must every
loss value on the way.  Also: ClipAdam(capturable=True) enqueued step by step equals the default ClipAdam bit for bit."""
import pytest
import torch

import color_neus_amd as cn
from color_neus_amd import rays as raygen, synthetic
from color_neus_amd.graph import GraphedStep, PinnedStager

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
R, H, W = 256, 64, 64


def _setup(capturable):
    lib = cn.load_library()
    cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)   # DTU widths: the chain-fused kernels
    torch.manual_seed(0)
    r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(DEV)
    opt = cn.ClipAdam(r._ordered_params(), lr=5e-4, betas=(0.9, 0.99), eps=1e-8, max_norm=1.0, library=lib, capturable=capturable)
    cam = synthetic.synthetic_camera(H, W, seed=1, device=DEV)
    perm = torch.randperm(H * W, generator=torch.Generator().manual_seed(7)).to(DEV)
    return lib, r, opt, cam, perm


def _fn(lib, r, opt, cam):
    c2w, focal, image, mask = cam
    params = list(r.parameters())

    def fn(idx, t):
        o, d, rgb, msel, near, far = raygen._generate(lib, idx, R, c2w, focal, H, W, True, False, image=image, mask=mask, origin=None, radius=1.0,
                                                      want_nearfar=True)
        out = r(o, d, near, far, t_rand=t)
        loss, _ = cn.compute_loss_fused(out, rgb, msel, library=lib)
        for p in params:
            p.grad = None
        loss.backward()
        opt.step()
        return loss
    return fn


def _eager(capturable, nsteps):
    lib, r, opt, cam, perm = _setup(capturable)
    fn = _fn(lib, r, opt, cam)
    torch.manual_seed(2)
    losses = []
    for i in range(nsteps):
        idx = perm[i * R:(i + 1) * R].clone()
        t = torch.rand([R, 1]).to(DEV)
        if capturable:
            opt.prepare_step()
        losses.append(float(fn(idx, t)))
    torch.cuda.synchronize()
    return losses, {k: v.detach().clone() for k, v in r.state_dict().items()}, opt


def test_capturable_clip_adam_is_the_default_clip_adam():
    l0, p0, _ = _eager(False, 4)
    l1, p1, _ = _eager(True, 4)
    assert l0 == l1
    for k in p0:
        assert torch.equal(p0[k], p1[k]), k


def test_graph_replay_is_bit_identical_to_the_enqueued_steps():
    nsteps = 5
    l_ref, p_ref, opt_ref = _eager(False, nsteps)
    lib, r, opt, cam, perm = _setup(True)
    static = {"idx": torch.empty(R, dtype=torch.int64, device=DEV), "t": torch.empty(R, 1, dtype=torch.float32, device=DEV)}
    state = {"i": 0, "stager": PinnedStager()}

    def refresh():
        i = state["i"]
        static["idx"].copy_(perm[i * R:(i + 1) * R])
        static["t"].copy_(state["stager"].to_device(torch.rand([R, 1]), DEV))
        state["i"] = i + 1

    torch.manual_seed(2)
    g = GraphedStep(_fn(lib, r, opt, cam), static, optimizer=opt, warmup=2, before_each=refresh)   # steps 0 and 1 run eagerly inside
    losses = [float(g.replay()) for _ in range(nsteps - 2)]                                             # steps 2, 3, 4: replays
    torch.cuda.synchronize()
    assert losses == l_ref[2:], (losses, l_ref)
    for k, v in r.state_dict().items():
        assert torch.equal(v, p_ref[k]), k
    # the optimiser state took the same path: moments bit-identical, step counts equal
    s_ref = opt_ref.state[opt_ref.param_groups[0]["params"][0]]
    s_g = opt.state[opt.param_groups[0]["params"][0]]
    assert torch.equal(s_ref["exp_avg"], s_g["exp_avg"]) and torch.equal(s_ref["exp_avg_sq"], s_g["exp_avg_sq"]) and s_ref["steps"] == s_g["steps"]
