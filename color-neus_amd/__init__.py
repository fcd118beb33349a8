"""MI355X-native volume renderer for Color-NeuS (hand-written HIP kernels behind a C ABI).

Public surface (mirrors the reference's RENDERER plug-in interface, lib/utils/builder.py:309):
    ColorNeuSRenderer / NeuSRenderer   nn.Modules with NeuS.forward's signature and return dict
    RENDERER, build_renderer            registry look-alike; register_into(reference_registry) for drop-in use
    load_library                        the ctypes binding of libcolorneus_hip.so
"""
from .config import RenderConfig, config_from_node  # noqa: F401
from ._lib import load_library, library_path, RenderLibrary  # noqa: F401
from .renderer import ColorNeuSRenderer, NeuSRenderer, RENDERER, build_renderer, register_into, sample_pdf  # noqa: F401
from .loss import compute_loss, compute_loss_fused  # noqa: F401
from . import parallel, rays, synthetic, optim, meshio  # noqa: F401
from .meshio import write_ply  # noqa: F401
from .optim import ClipAdam  # noqa: F401

__all__ = ["RenderConfig", "config_from_node", "load_library", "library_path", "RenderLibrary", "ColorNeuSRenderer",
           "NeuSRenderer", "RENDERER", "build_renderer", "register_into", "compute_loss", "compute_loss_fused", "parallel", "rays", "synthetic", "sample_pdf", "optim", "ClipAdam", "meshio", "write_ply"]
