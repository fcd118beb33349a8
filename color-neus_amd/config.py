"""Renderer configuration: the MODEL.RENDERER sub-tree of the reference's YAML files
(config/Color_NeuS_dtu.yml:23-60).  Accepts a yacs CfgNode, a plain dict or any object with ``get``/attributes;
reads the same keys with the same defaults as the reference classes (NeuS.py:80-85, fields.py:19-29,126-134,296-303)."""
from dataclasses import dataclass, field
from typing import List, Optional


def _get(node, key, default):
    if node is None:
        return default
    if hasattr(node, "get"):
        return node.get(key, default)
    return getattr(node, key, default)


@dataclass
class RenderConfig:
    type: str = "Color_NeuS"
    n_samples: int = 64
    n_importance: int = 64
    n_outside: int = 0
    up_sample_steps: int = 4
    perturb: float = 1.0
    N: int = 64
    # SDF
    sdf_d_in: int = 3
    sdf_d_out: int = 257
    sdf_d_hidden: int = 256
    sdf_n_layers: int = 8
    sdf_skip_in: List[int] = field(default_factory=lambda: [4])
    sdf_multires: int = 6
    sdf_bias: float = 0.5
    sdf_scale: float = 3.0
    sdf_geometric_init: bool = True
    sdf_weight_norm: bool = True
    sdf_inside_outside: bool = False
    # colour
    col_d_feature: int = 256
    col_mode: str = "idr"
    col_d_in: int = 9
    col_d_out: int = 3
    col_d_hidden: int = 256
    col_n_layers: int = 4
    col_weight_norm: bool = True
    col_multires_view: int = 4
    col_squeeze_out: bool = True
    # relight (Color_NeuS only)
    rel_d_in: int = 6
    rel_d_out: int = 3
    rel_d_hidden: int = 256
    rel_n_layers: int = 4
    rel_y_in_layer: int = 3
    rel_multires_view: int = 4
    rel_include_grad: bool = True
    rel_inv_sigmoid: bool = True
    # deviation
    init_val: float = 0.3

    @property
    def n_total(self) -> int:
        return self.n_samples + self.n_importance

    def validate(self):
        if self.type not in ("NeuS", "Color_NeuS"):
            raise ValueError(f"unknown renderer TYPE {self.type!r}")
        if self.type == "Color_NeuS" and self.col_mode != "no_view_dir":
            raise AssertionError("Color_NeuS requires COLOR.MODE == 'no_view_dir'")  # Color_NeuS.py:14
        if self.n_outside < 0 or self.n_samples + self.n_importance + self.n_outside > 256:
            raise ValueError("N_OUTSIDE must be >= 0 and N_SAMPLES + N_IMPORTANCE + N_OUTSIDE <= 256")
        if self.sdf_d_in != 3 or self.col_d_out != 3 or self.rel_d_out != 3:
            raise NotImplementedError("only 3-D points / RGB outputs are supported")
        if self.col_mode not in ("idr", "no_view_dir", "no_normal"):
            raise ValueError(f"no such mode: {self.col_mode}")


def config_from_node(node) -> RenderConfig:
    s, c, r, d = (_get(node, k, None) for k in ("SDF", "COLOR", "RELIGHT", "DEVIATION"))
    cfg = RenderConfig(
        type=_get(node, "TYPE", "Color_NeuS"),
        n_samples=_get(node, "N_SAMPLES", 64), n_importance=_get(node, "N_IMPORTANCE", 64),
        n_outside=_get(node, "N_OUTSIDE", 0), up_sample_steps=_get(node, "UP_SAMPLE_STEPS", 4),
        perturb=_get(node, "PERTURB", 1.0), N=_get(node, "N", 64),
        sdf_d_in=_get(s, "D_IN", 3), sdf_d_out=_get(s, "D_OUT", 257), sdf_d_hidden=_get(s, "D_HIDDEN", 256),
        sdf_n_layers=_get(s, "N_LAYERS", 8), sdf_skip_in=list(_get(s, "SKIP_IN", [4])), sdf_multires=_get(s, "MULTIRES", 6),
        sdf_bias=_get(s, "BIAS", 0.5), sdf_scale=_get(s, "SCALE", 3.0), sdf_geometric_init=_get(s, "GEOMETRIC_INIT", True),
        sdf_weight_norm=_get(s, "WEIGHT_NORM", True), sdf_inside_outside=_get(s, "INSIDE_OUTSIDE", False),
        col_d_feature=_get(c, "D_FEATURE", 256), col_mode=_get(c, "MODE", "idr"), col_d_in=_get(c, "D_IN", 9),
        col_d_out=_get(c, "D_OUT", 3), col_d_hidden=_get(c, "D_HIDDEN", 256), col_n_layers=_get(c, "N_LAYERS", 4),
        col_weight_norm=_get(c, "WEIGHT_NORM", True), col_multires_view=_get(c, "MULTIRES_VIEW", 4),
        col_squeeze_out=_get(c, "SQUEEZE_OUT", True),
        rel_d_in=_get(r, "D_IN", 6), rel_d_out=_get(r, "D_OUT", 3), rel_d_hidden=_get(r, "D_HIDDEN", 256),
        rel_n_layers=_get(r, "N_LAYERS", 4), rel_y_in_layer=_get(r, "Y_IN_LAYER", 3),
        rel_multires_view=_get(r, "MULTIRES_VIEW", 4), rel_include_grad=_get(r, "INCLUDE_GRAD", True),
        rel_inv_sigmoid=_get(r, "INV_SIGMOID", True),
        init_val=_get(d, "INIT_VAL", 0.3))
    return cfg
