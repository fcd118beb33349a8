"""Optimiser step of the training loop on the device library: per-parameter gradient clipping + Adam in one launch.

Replaces, in a reference-style loop (train.py:70-77),

    loss.backward()
    clip_gradient(optimizer, max_norm, norm_type)      # lib/utils/net_utils.py:174-184: clip_grad_norm_ on EACH parameter tensor
    optimizer.step()                                   # torch.optim.Adam(lr, betas=(0.9, 0.99), eps=1e-8), net_utils.py:88

by ``ClipAdam(params, lr, max_norm=...).step()``: ~150 small launches become one (cnr_clip_adam_step).  It is a
``torch.optim.Optimizer`` (param_groups / lr schedulers / state_dict work as usual); the moments live in two flat buffers.

The renderer's backward writes all parameter gradients into ONE flat buffer (the ``p.grad`` tensors are views into it), so a
ray-sharded run all-reduces that buffer directly (parallel.allreduce_gradients) -- no concatenation, no copy back."""
import ctypes as C

import torch

from . import _lib


def flat_view_of_grads(params):
    """If the gradients of ``params`` (in this order) tile one contiguous buffer, return that buffer as a 1-D view, else None."""
    grads = [p.grad for p in params]
    if not grads or any(g is None or not g.is_contiguous() or g.dtype != torch.float32 for g in grads):
        return None
    ptr = grads[0].data_ptr()
    for g in grads:
        if g.data_ptr() != ptr:
            return None
        ptr += g.numel() * 4
    g0 = grads[0]
    total = sum(g.numel() for g in grads)
    st = g0.untyped_storage()
    off = g0.storage_offset()
    if (off + total) * 4 > st.nbytes():   # consecutive addresses, but not inside one allocation
        return None
    return torch.empty(0, dtype=torch.float32, device=g0.device).set_(st, off, (total,))


class ClipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=5e-4, betas=(0.9, 0.99), eps=1e-8, max_norm=None, library=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, max_norm=max_norm)
        super().__init__(params, defaults)
        self._library = library
        self._lib_obj = None

    @property
    def _lib(self):
        if self._lib_obj is None:
            lib = self._library
            self._lib_obj = lib if isinstance(lib, _lib.RenderLibrary) else _lib.load_library(lib)
        return self._lib_obj

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("ClipAdam: parameters must be contiguous float32")
            st = self.state.setdefault(ps[0], {})   # group state hangs off the group's first parameter (state_dict-compatible)
            sizes = [p.numel() for p in ps]
            total = sum(sizes)
            if "exp_avg" not in st or st["exp_avg"].numel() != total:
                st["step"] = 0
                st["exp_avg"] = torch.zeros(total, dtype=torch.float32, device=ps[0].device)
                st["exp_avg_sq"] = torch.zeros(total, dtype=torch.float32, device=ps[0].device)
            st["step"] += 1
            b1, b2 = group["betas"]
            mn = group.get("max_norm")
            cfg = _lib.CnrAdamConfig(lr=float(group["lr"]), beta1=float(b1), beta2=float(b2), eps=float(group["eps"]),
                                     max_norm=float(mn) if mn else 0.0, step=int(st["step"]))
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            n = len(ps)
            sz = (C.c_int64 * n)(*sizes)
            pw = (C.c_void_p * n)(*[p.data_ptr() for p in ps])
            pg = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
            dev = ps[0].device
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
            nb = self._lib.lib.cnr_clip_adam_scratch_bytes(n, sz)
            if "scratch" not in st or st["scratch"].numel() < nb:
                st["scratch"] = torch.empty(nb, dtype=torch.uint8, device=dev)
            rc = self._lib.lib.cnr_clip_adam_step(C.byref(cfg), n, sz, pw, pg, C.c_void_p(st["exp_avg"].data_ptr()),
                                                  C.c_void_p(st["exp_avg_sq"].data_ptr()), C.c_void_p(st["scratch"].data_ptr()), nb, stream)
            self._lib.check(rc, "cnr_clip_adam_step")
        return loss
