"""ctypes binding of the C ABI declared in include/colorneus_render.h.

The product library is ``libcolorneus_hip.so`` next to this file (built by ``__graft_entry__.build()`` /
``make -C color-neus_amd/csrc hip``).  There is NO fallback: if it is missing, loading raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


class CnrConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("type", "n_samples", "n_importance", "up_sample_steps", "sdf_d_hidden", "sdf_n_layers",
                                         "sdf_d_out", "sdf_multires", "sdf_skip_mask", "sdf_weight_norm")] + \
               [("sdf_scale", C.c_float)] + \
               [(n, C.c_int32) for n in ("col_mode", "col_d_feature", "col_d_hidden", "col_n_layers", "col_multires_view",
                                         "col_weight_norm", "col_squeeze_out", "rel_d_hidden", "rel_n_layers", "rel_y_in_layer",
                                         "rel_multires_view", "rel_include_grad", "rel_inv_sigmoid")]


_FP = C.c_void_p


class CnrInputs(C.Structure):
    _fields_ = [("rays_o", _FP), ("rays_d", _FP), ("near_", _FP), ("far_", _FP), ("t_rand", _FP), ("z_vals_override", _FP),
                ("background_rgb", _FP), ("n_rays", C.c_int64), ("cos_anneal_ratio", C.c_float), ("prune_eps", C.c_float)]


OUTPUT_FIELDS = ["color_fine", "s_val", "cdf_fine", "weight_sum", "weight_max", "gradients", "weights", "gradient_error",
                 "inside_sphere", "depth", "global_color", "delta_relight", "z_vals", "eik_sums", "sdf_samples", "color_samples",
                 "global_color_samples", "delta_relight_ray_sum"]
OUT_GRAD_FIELDS = ["color_fine", "s_val", "cdf_fine", "weight_sum", "weight_max", "gradients", "weights", "gradient_error",
                   "depth", "global_color", "delta_relight", "sdf_samples", "color_samples", "global_color_samples",
                   "delta_relight_per_ray"]


class CnrOutputs(C.Structure):
    _fields_ = [(n, _FP) for n in OUTPUT_FIELDS]


class CnrOutGrads(C.Structure):
    _fields_ = [(n, _FP) for n in OUT_GRAD_FIELDS]


class CnrNerfConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("D", "W", "multires", "multires_view", "skip_mask")]


class CnrBgCompositeIn(C.Structure):
    _fields_ = [("rays_o", _FP), ("rays_d", _FP), ("z_vals", _FP), ("z_feed", _FP), ("n_rays", C.c_int64), ("n_z", C.c_int32), ("n_feed", C.c_int32),
                ("sample_dist", C.c_float), ("sdf_samples", _FP), ("gradients", _FP), ("color_samples", _FP), ("global_color_samples", _FP),
                ("bg_alpha", _FP), ("bg_color", _FP), ("variance", _FP), ("cos_anneal_ratio", C.c_float), ("background_rgb", _FP)]


class CnrBgCompositeGrads(C.Structure):
    _fields_ = [(n, _FP) for n in ("d_sdf_samples", "d_gradients", "d_color_samples", "d_global_color_samples", "d_bg_alpha", "d_bg_color",
                                   "d_variance", "d_rays_d", "d_z_vals", "d_z_feed")]


class CnrKernelTiming(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("kind", C.c_int32), ("nt", C.c_int32), ("P", C.c_int64), ("N", C.c_int32),
                ("K", C.c_int32), ("pairs", C.c_int32), ("ms", C.c_float), ("bytes", C.c_double)]


class CnrLossConfig(C.Structure):
    _fields_ = [("lambda_fine", C.c_float), ("lambda_eikonal", C.c_float), ("lambda_mask", C.c_float), ("lambda_relight", C.c_float),
                ("rgb_l1", C.c_int32), ("include_mask", C.c_int32)]


class CnrAdamConfig(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("max_norm", C.c_float),
                ("step", C.c_int32), ("hyper_dev", C.c_void_p)]


class CnrInGrads(C.Structure):
    _fields_ = [("d_params", C.POINTER(_FP)), ("d_rays_o", _FP), ("d_rays_d", _FP), ("d_near", _FP), ("d_far", _FP)]


_MODE = {"idr": 0, "no_view_dir": 1, "no_normal": 2}


def c_config(cfg) -> CnrConfig:
    mask = 0
    for l in cfg.sdf_skip_in:
        mask |= 1 << int(l)
    return CnrConfig(type=1 if cfg.type == "Color_NeuS" else 0, n_samples=cfg.n_samples, n_importance=cfg.n_importance,
                     up_sample_steps=cfg.up_sample_steps, sdf_d_hidden=cfg.sdf_d_hidden, sdf_n_layers=cfg.sdf_n_layers,
                     sdf_d_out=cfg.sdf_d_out, sdf_multires=cfg.sdf_multires, sdf_skip_mask=mask,
                     sdf_weight_norm=int(cfg.sdf_weight_norm), sdf_scale=float(cfg.sdf_scale), col_mode=_MODE[cfg.col_mode],
                     col_d_feature=cfg.col_d_feature, col_d_hidden=cfg.col_d_hidden, col_n_layers=cfg.col_n_layers,
                     col_multires_view=cfg.col_multires_view, col_weight_norm=int(cfg.col_weight_norm),
                     col_squeeze_out=int(cfg.col_squeeze_out), rel_d_hidden=cfg.rel_d_hidden, rel_n_layers=cfg.rel_n_layers,
                     rel_y_in_layer=cfg.rel_y_in_layer, rel_multires_view=cfg.rel_multires_view,
                     rel_include_grad=int(cfg.rel_include_grad), rel_inv_sigmoid=int(cfg.rel_inv_sigmoid))


EXPORTS = ["cnr_abi_version", "cnr_backend_name", "cnr_last_error", "cnr_param_count", "cnr_param_info", "cnr_ctx_bytes",
           "cnr_bwd_scratch_bytes", "cnr_render_forward", "cnr_render_backward", "cnr_infer_scratch_bytes", "cnr_render_forward_only", "cnr_sdf_eval_scratch_bytes", "cnr_sdf_eval",
           "cnr_sdf_grid_scratch_bytes", "cnr_sdf_grid", "cnr_sdf_grid_slab_scratch_bytes", "cnr_sdf_grid_slab", "cnr_vertex_color_scratch_bytes", "cnr_vertex_color",
           "cnr_timing_enable", "cnr_timing_collect", "cnr_loss_scratch_bytes", "cnr_loss_sums", "cnr_loss_sums_ray", "cnr_loss_grads", "cnr_loss_combine", "cnr_loss_coef", "cnr_loss_forward", "cnr_loss_backward", "cnr_loss_shard_stats", "cnr_loss_shard_combine",
           "cnr_sample_pdf", "cnr_sample_pdf_u", "cnr_up_sample", "cnr_clip_adam_step", "cnr_clip_adam_scratch_bytes", "cnr_gen_rays", "cnr_gen_rays_backward", "cnr_sample_z", "cnr_mc_scratch_bytes", "cnr_mc_count", "cnr_mc_emit",
           "cnr_linear_scratch_bytes", "cnr_linear_forward", "cnr_linear_backward",
           "cnr_nerf_param_count", "cnr_nerf_param_info", "cnr_outside_z", "cnr_outside_z_backward", "cnr_background_ctx_bytes",
           "cnr_background_bwd_scratch_bytes", "cnr_background_forward", "cnr_background_backward", "cnr_composite_background_scratch_bytes",
           "cnr_composite_background_forward", "cnr_composite_background_backward"]


class RenderLibrary:
    def __init__(self, path):
        if not os.path.isfile(path):
            raise RuntimeError(
                f"Color-NeuS HIP library not found at {path}. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C color-neus_amd/csrc hip`. There is no CPU/PyTorch fallback for the render path.")
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        L.cnr_abi_version.restype = C.c_int
        L.cnr_backend_name.restype = C.c_char_p
        L.cnr_last_error.restype = C.c_char_p
        L.cnr_param_count.argtypes = [C.POINTER(CnrConfig)]
        L.cnr_param_info.argtypes = [C.POINTER(CnrConfig), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        for f in ("cnr_ctx_bytes", "cnr_bwd_scratch_bytes", "cnr_infer_scratch_bytes", "cnr_sdf_eval_scratch_bytes", "cnr_vertex_color_scratch_bytes"):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [C.POINTER(CnrConfig), C.c_int64]
        L.cnr_sdf_grid_scratch_bytes.restype = C.c_size_t
        L.cnr_sdf_grid_scratch_bytes.argtypes = [C.POINTER(CnrConfig), C.c_int32]
        L.cnr_render_forward.argtypes = [C.POINTER(CnrConfig), C.POINTER(_FP), C.POINTER(CnrInputs), C.POINTER(CnrOutputs),
                                         _FP, C.c_size_t, _FP]
        L.cnr_render_forward_only.argtypes = [C.POINTER(CnrConfig), C.POINTER(_FP), C.POINTER(CnrInputs), C.POINTER(CnrOutputs),
                                              _FP, C.c_size_t, _FP]
        L.cnr_render_backward.argtypes = [C.POINTER(CnrConfig), C.POINTER(_FP), C.POINTER(CnrInputs), C.POINTER(CnrOutputs),
                                          _FP, C.c_size_t, C.POINTER(CnrOutGrads), C.POINTER(CnrInGrads), _FP, C.c_size_t, _FP]
        L.cnr_sdf_eval.argtypes = [C.POINTER(CnrConfig), C.POINTER(_FP), _FP, C.c_int64, C.c_float, _FP, _FP, C.c_size_t, _FP]
        L.cnr_sdf_grid.argtypes = [C.POINTER(CnrConfig), C.POINTER(_FP), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32,
                                   _FP, _FP, C.c_size_t, _FP]
        L.cnr_sdf_grid_slab_scratch_bytes.restype = C.c_size_t
        L.cnr_sdf_grid_slab_scratch_bytes.argtypes = [C.POINTER(CnrConfig), C.c_int32, C.c_int32, C.c_int32]
        L.cnr_sdf_grid_slab.argtypes = [C.POINTER(CnrConfig), C.POINTER(_FP), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.c_int32, C.c_int32,
                                        _FP, _FP, C.c_size_t, _FP]
        L.cnr_linear_scratch_bytes.restype = C.c_size_t
        L.cnr_linear_scratch_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int32]
        L.cnr_linear_forward.argtypes = [_FP, C.c_int64, C.c_int32, _FP, _FP, C.c_int32, C.c_int32, _FP, _FP, C.c_size_t, _FP]
        L.cnr_linear_backward.argtypes = [_FP, _FP, _FP, C.c_int64, C.c_int32, _FP, C.c_int32, C.c_int32, _FP, _FP, _FP, _FP, C.c_size_t, _FP]
        L.cnr_vertex_color.argtypes = [C.POINTER(CnrConfig), C.POINTER(_FP), _FP, C.c_int64, _FP, _FP, C.c_size_t, _FP]
        L.cnr_loss_scratch_bytes.restype = C.c_size_t
        L.cnr_loss_scratch_bytes.argtypes = [C.c_int64]
        L.cnr_loss_sums.argtypes = [C.POINTER(CnrLossConfig), _FP, _FP, _FP, _FP, _FP, C.c_int64, C.c_int32, _FP, _FP, C.c_size_t, _FP]
        L.cnr_loss_sums_ray.argtypes = [C.POINTER(CnrLossConfig), _FP, _FP, _FP, _FP, _FP, C.c_int64, C.c_int32, _FP, _FP, C.c_size_t, _FP]
        L.cnr_loss_grads.argtypes = [C.POINTER(CnrLossConfig), _FP, _FP, _FP, _FP, C.c_int64, C.c_int32, _FP, _FP, _FP, _FP, _FP]
        L.cnr_loss_combine.argtypes = [C.POINTER(CnrLossConfig), _FP, _FP, C.c_float, C.c_int32, C.c_int32, C.c_int32, _FP, _FP]
        L.cnr_loss_coef.argtypes = [C.POINTER(CnrLossConfig), _FP, _FP, C.c_float, C.c_int32, C.c_int32, C.c_int32, _FP, _FP]
        L.cnr_loss_forward.argtypes = [C.POINTER(CnrLossConfig), _FP, _FP, _FP, C.c_int32, _FP, _FP, _FP, C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_int32,
                                       _FP, _FP, _FP, C.c_size_t, _FP]
        L.cnr_loss_backward.argtypes = [C.POINTER(CnrLossConfig), _FP, _FP, _FP, _FP, C.c_int64, C.c_int32, _FP, _FP, _FP, C.c_float, C.c_int32, C.c_int32,
                                        _FP, _FP, _FP, _FP, _FP]
        L.cnr_loss_shard_stats.argtypes = [C.POINTER(CnrLossConfig), _FP, _FP, _FP, C.c_int32, _FP, _FP, _FP, C.c_int64, C.c_int32, _FP, _FP, C.c_size_t, _FP]
        L.cnr_loss_shard_combine.argtypes = [C.POINTER(CnrLossConfig), _FP, C.c_float, C.c_int32, C.c_int32, C.c_int32, _FP, _FP]
        L.cnr_sample_pdf.argtypes = [_FP, _FP, C.c_int64, C.c_int32, C.c_int32, _FP, _FP]
        L.cnr_sample_pdf_u.argtypes = [_FP, _FP, _FP, C.c_int64, C.c_int32, C.c_int32, _FP, _FP]
        L.cnr_up_sample.argtypes = [_FP, _FP, _FP, _FP, C.c_int64, C.c_int32, C.c_int32, C.c_float, _FP, _FP]
        L.cnr_clip_adam_step.argtypes = [C.POINTER(CnrAdamConfig), C.c_int32, C.POINTER(C.c_int64), C.POINTER(_FP), C.POINTER(_FP), _FP, _FP, _FP,
                                         C.c_size_t, _FP]
        L.cnr_clip_adam_scratch_bytes.restype = C.c_size_t
        L.cnr_clip_adam_scratch_bytes.argtypes = [C.c_int32, C.POINTER(C.c_int64)]
        L.cnr_gen_rays.argtypes = [_FP, C.c_int64, _FP, C.c_int32, _FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP, _FP, _FP, C.c_float,
                                   _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP]
        L.cnr_gen_rays_backward.argtypes = [_FP, C.c_int64, _FP, C.c_int32, _FP, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _FP, C.c_float,
                                            _FP, _FP, _FP, _FP, _FP, _FP, _FP, C.c_size_t, _FP]
        L.cnr_sample_z.argtypes = [C.POINTER(CnrConfig), C.POINTER(_FP), C.POINTER(CnrInputs), _FP, _FP, C.c_size_t, _FP]
        L.cnr_mc_scratch_bytes.restype = C.c_size_t
        L.cnr_mc_scratch_bytes.argtypes = [C.c_int32]
        L.cnr_mc_count.argtypes = [_FP, C.c_int32, C.c_float, _FP, C.c_size_t, _FP, _FP]
        L.cnr_mc_emit.argtypes = [_FP, C.c_int32, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), _FP, C.c_size_t, _FP, _FP, _FP]
        L.cnr_timing_enable.argtypes = [C.c_int]
        L.cnr_timing_enable.restype = None
        L.cnr_timing_collect.argtypes = [C.POINTER(CnrKernelTiming), C.c_int]
        L.cnr_nerf_param_count.argtypes = [C.POINTER(CnrNerfConfig)]
        L.cnr_nerf_param_info.argtypes = [C.POINTER(CnrNerfConfig), C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.cnr_outside_z.argtypes = [_FP, _FP, _FP, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _FP, _FP, _FP]
        L.cnr_outside_z_backward.argtypes = [_FP, _FP, _FP, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _FP, _FP, _FP]
        for f in ("cnr_background_ctx_bytes", "cnr_background_bwd_scratch_bytes"):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [C.POINTER(CnrNerfConfig), C.c_int64, C.c_int32]
        L.cnr_background_forward.argtypes = [C.POINTER(CnrNerfConfig), C.POINTER(_FP), _FP, _FP, _FP, C.c_int64, C.c_int32, C.c_float, _FP, _FP, _FP,
                                             C.c_size_t, _FP]
        L.cnr_background_backward.argtypes = [C.POINTER(CnrNerfConfig), C.POINTER(_FP), _FP, _FP, _FP, C.c_int64, C.c_int32, C.c_float, _FP, C.c_size_t,
                                              _FP, _FP, _FP, C.POINTER(_FP), _FP, _FP, _FP, _FP, C.c_size_t, _FP]
        L.cnr_composite_background_scratch_bytes.restype = C.c_size_t
        L.cnr_composite_background_scratch_bytes.argtypes = [C.c_int64]
        L.cnr_composite_background_forward.argtypes = [C.POINTER(CnrBgCompositeIn), C.POINTER(CnrOutputs), _FP, C.c_size_t, _FP]
        L.cnr_composite_background_backward.argtypes = [C.POINTER(CnrBgCompositeIn), C.POINTER(CnrOutputs), C.POINTER(CnrOutGrads),
                                                        C.POINTER(CnrBgCompositeGrads), _FP, C.c_size_t, _FP]
        if L.cnr_abi_version() != 8:
            raise RuntimeError("colorneus library ABI mismatch")

    @property
    def backend(self) -> str:
        return self.lib.cnr_backend_name().decode()

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.cnr_last_error().decode()}")

    def timing_enable(self, on: bool):
        self.lib.cnr_timing_enable(1 if on else 0)

    def timing_collect(self, max_records=65536):
        """Per-launch records [(name, kind, nt, P, N, K, pairs, ms, bytes)] since the last collect (synchronises the events)."""
        buf = (CnrKernelTiming * max_records)()
        n = self.lib.cnr_timing_collect(buf, max_records)
        return [(buf[i].name.decode(), buf[i].kind, buf[i].nt, buf[i].P, buf[i].N, buf[i].K, buf[i].pairs, buf[i].ms, buf[i].bytes)
                for i in range(min(n, max_records))]

    def param_inventory(self, ccfg):
        n = self.lib.cnr_param_count(C.byref(ccfg))
        if n < 0:
            raise RuntimeError(f"unsupported renderer configuration: {self.lib.cnr_last_error().decode()}")
        out = []
        buf = C.create_string_buffer(128)
        r, c = C.c_int(), C.c_int()
        for i in range(n):
            self.check(self.lib.cnr_param_info(C.byref(ccfg), i, buf, 128, C.byref(r), C.byref(c)), "cnr_param_info")
            out.append((buf.value.decode(), r.value, c.value))
        return out


def library_path() -> str:
    return os.path.join(_HERE, "libcolorneus_hip.so")


_cached = {}


def load_library(path=None) -> RenderLibrary:
    """Load the HIP library (default) or an explicitly given build (tests pass the CPU-emulation build explicitly)."""
    path = os.path.abspath(path or library_path())
    if path not in _cached:
        _cached[path] = RenderLibrary(path)
    return _cached[path]
