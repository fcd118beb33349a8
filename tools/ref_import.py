"""Import harness for the upstream reference (THIS CONTAINER ONLY).

The reference checkout at /root/reference is pure Python but its module-level
imports pull in packages that are not installed here (yacs, termcolor, mcubes,
cv2, torchvision, pytorch3d, imageio, trimesh, kornia, git ...).  None of them
takes part in the arithmetic of the rendering hot path, so we pre-seed
``sys.modules`` with inert stand-ins and then import the reference unmodified.

Used only by tools/gen_golden.py (golden-vector capture) and by optional
oracle-vs-reference checks that are skipped when /root/reference is absent.
Nothing from the reference is copied into this repository.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("COLORNEUS_REFERENCE", "/root/reference")


class CfgNode(dict):
    """Minimal stand-in for yacs.config.CfgNode: dict with attribute access."""

    def __init__(self, init=None, new_allowed=False, **kw):
        super().__init__()
        init = {} if init is None else init
        for k, v in init.items():
            if isinstance(v, dict) and not isinstance(v, CfgNode):
                v = CfgNode(v)
            self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def defrost(self):
        pass

    def freeze(self):
        pass

    def set_new_allowed(self, flag):
        pass

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            self[k] = v

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self.merge_from_other_cfg(CfgNode(yaml.safe_load(f)))

    def dump(self, *a, **k):
        import yaml
        return yaml.safe_dump(dict(self))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    if "yacs" not in sys.modules:
        _mod("yacs")
        _mod("yacs.config", CfgNode=CfgNode)
    if "termcolor" not in sys.modules:
        _mod("termcolor", colored=lambda s, *a, **k: s)
    if "mcubes" not in sys.modules:
        _mod("mcubes", marching_cubes=lambda *a, **k: (_ for _ in ()).throw(RuntimeError("mcubes stub")))
    if "cv2" not in sys.modules:
        _mod("cv2", setNumThreads=lambda n: None, COLORMAP_HOT=11)
    if "torchvision" not in sys.modules:
        tv = _mod("torchvision")
        tr = _mod("torchvision.transforms", ToTensor=object)
        fn = _mod("torchvision.transforms.functional")
        tv.transforms = tr
        tr.functional = fn
    if "pytorch3d" not in sys.modules:
        _mod("pytorch3d")
        names = ["axis_angle_to_matrix", "axis_angle_to_quaternion", "euler_angles_to_matrix",
                 "matrix_to_euler_angles", "matrix_to_quaternion", "matrix_to_rotation_6d",
                 "quaternion_to_axis_angle", "quaternion_to_matrix", "rotation_6d_to_matrix"]
        _mod("pytorch3d.transforms", **{n: None for n in names})
    for name in ["imageio", "trimesh", "kornia", "plyfile"]:
        if name not in sys.modules:
            _mod(name)
    if "kornia.metrics" not in sys.modules:
        _mod("kornia.metrics", ssim=None)
    if "git" not in sys.modules:
        _mod("git", Repo=object)
    if "tensorboardX" not in sys.modules:
        _mod("tensorboardX", SummaryWriter=object)


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lib", "models", "renderers"))


def import_reference():
    """Returns (Color_NeuS, NeuS, CfgNode, modules dict) from the reference."""
    if not reference_available():
        raise RuntimeError("reference checkout not present at %s" % REFERENCE_ROOT)
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import logging
    from lib.models.renderers.Color_NeuS import Color_NeuS
    from lib.models.renderers.NeuS import NeuS
    from lib.models.renderers import fields
    from lib.models.tools import ray_utils
    from lib.utils import transform
    try:
        from lib.utils.logger import logger
        logger.setLevel(logging.ERROR)
    except Exception:
        pass
    return Color_NeuS, NeuS, CfgNode, dict(fields=fields, ray_utils=ray_utils, transform=transform)
