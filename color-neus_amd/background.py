"""N_OUTSIDE > 0: the NeRF++ background of NeuS (lib/models/renderers/NeuS.py:95-134, 313-369; NeRF, fields.py:192-274) on the device library.

No shipped configuration enables it (N_OUTSIDE is absent from every config/*.yml, NeuS.py:82).  Everything arithmetic runs behind the C ABI
(include/colorneus_render.h, "N_OUTSIDE > 0"): this module holds the background network's PARAMETERS under the reference's names
(``nerf.pts_linears.0.weight`` ...: reference checkpoints load with strict=True) and the three autograd edges around the library calls

    OutsideZ        cnr_outside_z / cnr_outside_z_backward                        z_vals_outside, sorted z_vals_feed          NeuS.py:315-338, 353-355
    Background      cnr_background_forward / cnr_background_backward              render_core_outside: alpha, sigmoid(rgb)    NeuS.py:95-134
    CompositeBg     cnr_composite_background_forward / ..._backward               render_core's mixing + compositing          NeuS.py:236-292, Color_NeuS.py:66-138

The foreground fields (sdf, normals, colours per sample) come from the render call's per-sample outputs and take their gradients back through
it (renderer._RenderFunction).  No tensor arithmetic of the path is done in torch here; without the library the module fails loudly."""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream_of(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else C.c_void_p(0)


class NeRF(nn.Module):
    """Parameter storage of the reference's NeRF background network with its defaults (NeuS.__init__ falls back to NeRF() for every cfg,
    NeuS.py:87-91): 8 x 256 ReLU layers on PE-10 of the 4-vector (x / r, 1 / r), skip at 4, view branch on PE-4 (fields.py:215-231)."""

    def __init__(self, D=8, W=256, d_in=4, d_in_view=3, multires=10, multires_view=4, skips=(4,)):
        super().__init__()
        self.D, self.W, self.multires, self.multires_view, self.skips = D, W, multires, multires_view, tuple(skips)
        ch = d_in * (1 + 2 * multires)
        ch_view = d_in_view * (1 + 2 * multires_view)
        self.pts_linears = nn.ModuleList([nn.Linear(ch, W)] + [nn.Linear(W + ch, W) if i in self.skips else nn.Linear(W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(ch_view + W, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        self._order = None

    def config(self):
        mask = 0
        for i in self.skips:
            mask |= 1 << i
        return _lib.CnrNerfConfig(D=self.D, W=self.W, multires=self.multires, multires_view=self.multires_view, skip_mask=mask)

    def ordered_params(self, lib):
        """Parameters in the library's canonical order (cnr_nerf_param_info), checked against this module's own names and shapes."""
        if self._order is None:
            cfg = self.config()
            named = dict(self.named_parameters())
            n = lib.lib.cnr_nerf_param_count(C.byref(cfg))
            if n < 0:
                lib.check(n, "cnr_nerf_param_count")
            order = []
            for i in range(n):
                buf = C.create_string_buffer(128)
                rows, cols = C.c_int(), C.c_int()
                lib.check(lib.lib.cnr_nerf_param_info(C.byref(cfg), i, buf, 128, C.byref(rows), C.byref(cols)), "cnr_nerf_param_info")
                name = buf.value.decode()
                if name not in named or named[name].numel() != rows.value * cols.value:
                    raise RuntimeError(f"background parameter inventory mismatch at {name}")
                order.append(name)
            if set(order) != set(named):
                raise RuntimeError("background parameter inventory mismatch between module and library")
            self._order = order
        named = dict(self.named_parameters())
        return [named[k] for k in self._order]

    def forward(self, *a, **k):
        raise RuntimeError("the background network is evaluated by the render library (background.Background), not by this module")


class OutsideZ(torch.autograd.Function):
    """z_vals_feed = sort(cat(z_vals, far / flip(zz) + 1 / n_samples)) and the source index of every entry."""

    @staticmethod
    def forward(ctx, lib, far, t_rand, z_vals, n_samples, n_outside):
        R, M = z_vals.shape
        dev = z_vals.device
        far_c = far.detach().reshape(-1).contiguous().float()
        z_c = z_vals.detach().contiguous().float()
        t_c = t_rand.detach().to(dev).contiguous().float() if t_rand is not None else None
        z_feed = torch.empty(R, M + n_outside, dtype=torch.float32, device=dev)
        src = torch.empty(R, M + n_outside, dtype=torch.int32, device=dev)
        lib.check(lib.lib.cnr_outside_z(_ptr(far_c), _ptr(t_c), _ptr(z_c), R, M, n_outside, n_samples, _ptr(z_feed), _ptr(src), _stream_of(z_c)),
                  "cnr_outside_z")
        ctx.lib, ctx.meta = lib, (R, M, n_samples, n_outside, far.shape, t_c is not None)
        ctx.save_for_backward(src, t_c if t_c is not None else torch.empty(0, device=dev))
        ctx.mark_non_differentiable(src)
        return z_feed, src

    @staticmethod
    def backward(ctx, d_z_feed, _d_src):
        src, t_c = ctx.saved_tensors
        R, M, n_samples, n_outside, far_shape, has_t = ctx.meta
        lib = ctx.lib
        dz = d_z_feed.contiguous().float()
        d_far = torch.empty(R, dtype=torch.float32, device=dz.device)
        d_z = torch.empty(R, M, dtype=torch.float32, device=dz.device) if ctx.needs_input_grad[3] else None
        lib.check(lib.lib.cnr_outside_z_backward(_ptr(t_c if has_t else None), _ptr(src), _ptr(dz), R, M, n_outside, n_samples, _ptr(d_far), _ptr(d_z),
                                                 _stream_of(dz)), "cnr_outside_z_backward")
        return None, d_far.reshape(far_shape), None, d_z, None, None


class Background(torch.autograd.Function):
    """render_core_outside (NeuS.py:95-134): per-sample alpha and colour of the background network at z_feed."""

    @staticmethod
    def forward(ctx, lib, ncfg, sample_dist, rays_o, rays_d, z_feed, *params):
        R, MF = z_feed.shape
        dev = z_feed.device
        o, d, zf = rays_o.detach().contiguous().float(), rays_d.detach().contiguous().float(), z_feed.detach().contiguous().float()
        plist = [p.detach().contiguous().float() for p in params]
        parr = (C.c_void_p * len(plist))(*[p.data_ptr() for p in plist])
        alpha = torch.empty(R, MF, dtype=torch.float32, device=dev)
        color = torch.empty(R, MF, 3, dtype=torch.float32, device=dev)
        nb = lib.lib.cnr_background_ctx_bytes(C.byref(ncfg), R, MF)
        buf = torch.empty(nb, dtype=torch.uint8, device=dev)
        lib.check(lib.lib.cnr_background_forward(C.byref(ncfg), parr, _ptr(o), _ptr(d), _ptr(zf), R, MF, float(sample_dist), _ptr(alpha), _ptr(color),
                                                 _ptr(buf), nb, _stream_of(zf)), "cnr_background_forward")
        ctx.lib, ctx.ncfg, ctx.meta = lib, ncfg, (R, MF, float(sample_dist), len(plist))
        ctx.save_for_backward(o, d, zf, color, buf, *plist)
        ctx.set_materialize_grads(False)
        return alpha, color

    @staticmethod
    def backward(ctx, d_alpha, d_color):
        lib, ncfg = ctx.lib, ctx.ncfg
        R, MF, sample_dist, npar = ctx.meta
        sv = ctx.saved_tensors
        o, d, zf, color, buf = sv[:5]
        plist = list(sv[5:])
        dev = zf.device
        da = d_alpha.contiguous().float() if d_alpha is not None else None
        dc = d_color.contiguous().float() if d_color is not None else None
        parr = (C.c_void_p * npar)(*[p.data_ptr() for p in plist])
        flat = torch.empty(sum(p.numel() for p in plist), dtype=torch.float32, device=dev)
        gviews, off = [], 0
        for p in plist:
            gviews.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        garr = (C.c_void_p * npar)(*[g.data_ptr() for g in gviews])
        d_o, d_d = torch.empty(R, 3, dtype=torch.float32, device=dev), torch.empty(R, 3, dtype=torch.float32, device=dev)
        d_zf = torch.empty(R, MF, dtype=torch.float32, device=dev)
        nb = lib.lib.cnr_background_bwd_scratch_bytes(C.byref(ncfg), R, MF)
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        lib.check(lib.lib.cnr_background_backward(C.byref(ncfg), parr, _ptr(o), _ptr(d), _ptr(zf), R, MF, sample_dist, _ptr(buf), buf.numel(), _ptr(color),
                                                  _ptr(da), _ptr(dc), garr, _ptr(d_o), _ptr(d_d), _ptr(d_zf), _ptr(scratch), nb, _stream_of(zf)),
                  "cnr_background_backward")
        return (None, None, None, d_o, d_d, d_zf, *gviews)


_COMP_OUT = ["color_fine", "s_val", "cdf_fine", "weight_sum", "weight_max", "weights", "gradient_error", "depth", "global_color"]


class CompositeBg(torch.autograd.Function):
    """The tail of render_core with a background: S-density alpha, inside / outside mixing, compositing over M + N_OUTSIDE samples."""

    @staticmethod
    def forward(ctx, lib, sample_dist, cos_anneal_ratio, background_rgb, rays_o, rays_d, z_vals, z_feed, sdf, grads, color, gcolor, bg_alpha, bg_color,
                variance):
        R, M = z_vals.shape
        MF = z_feed.shape[1]
        dev = z_vals.device
        f32 = dict(dtype=torch.float32, device=dev)
        c = lambda t: t.detach().contiguous().float() if t is not None else None
        t = dict(rays_o=c(rays_o), rays_d=c(rays_d), z_vals=c(z_vals), z_feed=c(z_feed), sdf=c(sdf), grads=c(grads), color=c(color), gcolor=c(gcolor),
                 bg_alpha=c(bg_alpha), bg_color=c(bg_color), variance=c(variance).reshape(-1), bg=c(background_rgb))
        out = dict(color_fine=torch.empty(R, 3, **f32), s_val=torch.empty(R, 1, **f32), cdf_fine=torch.empty(R, M, **f32), weight_sum=torch.empty(R, 1, **f32),
                   weight_max=torch.empty(R, 1, **f32), weights=torch.empty(R, MF, **f32), gradient_error=torch.empty((), **f32),
                   inside_sphere=torch.empty(R, M, **f32), depth=torch.empty(R, **f32), global_color=torch.empty(R, 3, **f32) if gcolor is not None else None,
                   eik_sums=torch.empty(2, **f32))
        cin = CompositeBg._cin(t, R, M, MF, sample_dist, cos_anneal_ratio)
        cout = _lib.CnrOutputs(**{k: _ptr(out.get(k)) for k in _lib.OUTPUT_FIELDS})
        nb = lib.lib.cnr_composite_background_scratch_bytes(R)
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        lib.check(lib.lib.cnr_composite_background_forward(C.byref(cin), C.byref(cout), _ptr(scratch), nb, _stream_of(t["z_vals"])), "cnr_composite_background_forward")
        ctx.lib, ctx.meta = lib, (R, M, MF, float(sample_dist), float(cos_anneal_ratio), gcolor is not None, background_rgb is not None, variance.shape)
        ctx.save_for_backward(*[t[k] if t[k] is not None else torch.empty(0, **f32) for k in
                                ("rays_o", "rays_d", "z_vals", "z_feed", "sdf", "grads", "color", "gcolor", "bg_alpha", "bg_color", "variance", "bg")],
                              out["weights"], out["eik_sums"])
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(out["inside_sphere"], out["eik_sums"])
        ctx.has_gc = gcolor is not None
        res = [out[k] for k in _COMP_OUT if out[k] is not None]
        return (*res, out["inside_sphere"], out["eik_sums"])

    @staticmethod
    def _cin(t, R, M, MF, sample_dist, cos_anneal_ratio):
        return _lib.CnrBgCompositeIn(rays_o=_ptr(t["rays_o"]), rays_d=_ptr(t["rays_d"]), z_vals=_ptr(t["z_vals"]), z_feed=_ptr(t["z_feed"]), n_rays=R, n_z=M,
                                     n_feed=MF, sample_dist=float(sample_dist), sdf_samples=_ptr(t["sdf"]), gradients=_ptr(t["grads"]),
                                     color_samples=_ptr(t["color"]), global_color_samples=_ptr(t["gcolor"]), bg_alpha=_ptr(t["bg_alpha"]),
                                     bg_color=_ptr(t["bg_color"]), variance=_ptr(t["variance"]), cos_anneal_ratio=float(cos_anneal_ratio),
                                     background_rgb=_ptr(t["bg"]))

    @staticmethod
    def backward(ctx, *gouts):
        lib = ctx.lib
        R, M, MF, sample_dist, car, has_gc, has_bg, var_shape = ctx.meta
        sv = ctx.saved_tensors
        keys = ("rays_o", "rays_d", "z_vals", "z_feed", "sdf", "grads", "color", "gcolor", "bg_alpha", "bg_color", "variance", "bg")
        t = {k: (v if v.numel() > 0 else None) for k, v in zip(keys, sv[:12])}
        weights, eik_sums = sv[12], sv[13]
        dev = weights.device
        f32 = dict(dtype=torch.float32, device=dev)
        names = [k for k in _COMP_OUT if has_gc or k != "global_color"]
        g = {k: (v.contiguous().float() if v is not None else None) for k, v in zip(names, gouts[:len(names)])}
        go = _lib.CnrOutGrads(**{k: _ptr(g.get(k)) for k in _lib.OUT_GRAD_FIELDS})
        cin = CompositeBg._cin(t, R, M, MF, sample_dist, car)
        cout = _lib.CnrOutputs(**{k: _ptr({"weights": weights, "eik_sums": eik_sums}.get(k)) for k in _lib.OUTPUT_FIELDS})
        # (composite_bg_args checks the forward output pointers: the backward only reads weights and eik_sums, the rest may be any valid buffers)
        dummyR = torch.empty(R, max(M, 3), **f32)
        for k in ("color_fine", "s_val", "cdf_fine", "weight_sum", "weight_max", "inside_sphere", "depth", "gradient_error", "global_color"):
            setattr(cout, k, _ptr(dummyR))
        d = dict(d_sdf_samples=torch.empty(R, M, **f32), d_gradients=torch.empty(R, M, 3, **f32), d_color_samples=torch.empty(R, M, 3, **f32),
                 d_global_color_samples=torch.empty(R, M, 3, **f32) if has_gc else None, d_bg_alpha=torch.empty(R, MF, **f32),
                 d_bg_color=torch.empty(R, MF, 3, **f32), d_variance=torch.empty(1, **f32), d_rays_d=torch.empty(R, 3, **f32),
                 d_z_vals=torch.zeros(R, M, **f32), d_z_feed=torch.empty(R, MF, **f32))
        gi = _lib.CnrBgCompositeGrads(**{k: _ptr(v) for k, v in d.items()})
        nb = lib.lib.cnr_composite_background_scratch_bytes(R)
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        lib.check(lib.lib.cnr_composite_background_backward(C.byref(cin), C.byref(cout), C.byref(go), C.byref(gi), _ptr(scratch), nb, _stream_of(weights)),
                  "cnr_composite_background_backward")
        need = ctx.needs_input_grad
        return (None, None, None, None, None, d["d_rays_d"] if need[5] else None, d["d_z_vals"] if need[6] else None, d["d_z_feed"] if need[7] else None,
                d["d_sdf_samples"], d["d_gradients"], d["d_color_samples"], d["d_global_color_samples"], d["d_bg_alpha"], d["d_bg_color"],
                d["d_variance"].reshape(var_shape))
