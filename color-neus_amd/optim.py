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
    """clip_gradient (per parameter tensor, lib/utils/net_utils.py:174-184) + Adam (net_utils.py:88) as one library call per step.

    ``capturable=True``: the step-dependent scalars (lr, 1 - beta1^t, sqrt(1 - beta2^t)) are read by the kernel from three floats in device
    memory instead of being passed by value, so that ``step()`` can be captured ONCE in a HIP graph (graph.GraphedStep) and replayed for every
    step: call ``prepare_step()`` before each step / replay -- it advances the step count and refreshes the three floats (pinned staging, no
    host stall); ``step()`` itself then only enqueues the launches.  Results are identical to the default mode (same float32 values)."""

    def __init__(self, params, lr=5e-4, betas=(0.9, 0.99), eps=1e-8, max_norm=None, library=None, capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, max_norm=max_norm)
        super().__init__(params, defaults)
        self._library = library
        self._lib_obj = None
        self._scratch = {}
        self.capturable = bool(capturable)
        self._hyper = {}      # group index -> (device float[3], PinnedStager)
        self._prepared = False

    @property
    def _lib(self):
        if self._lib_obj is None:
            lib = self._library
            self._lib_obj = lib if isinstance(lib, _lib.RenderLibrary) else _lib.load_library(lib)
        return self._lib_obj

    def _group_state(self, gi, group):
        allp = list(group["params"])
        st = self.state.setdefault(allp[0], {})
        total = sum(p.numel() for p in allp)
        dev = allp[0].device
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros(total, dtype=torch.float32, device=dev)
            st["exp_avg_sq"] = torch.zeros(total, dtype=torch.float32, device=dev)
            st["steps"] = [0] * len(allp)
        if "steps" not in st and "step" in st and st["exp_avg"].numel() == total:   # state of an earlier build: one step count for the whole group
            st["steps"] = [int(st.pop("step"))] * len(allp)
        return allp, st, dev

    @torch.no_grad()
    def prepare_step(self):
        """capturable mode: advance every group's step count and refresh its device-side {lr, 1 - beta1^t, sqrt(1 - beta2^t)} (enqueued on the current
        stream, before the step / graph replay that reads them).  All parameters of a group step together in this mode."""
        if not self.capturable:
            raise RuntimeError("prepare_step() belongs to ClipAdam(capturable=True)")
        from .graph import PinnedStager
        for gi, group in enumerate(self.param_groups):
            allp, st, dev = self._group_state(gi, group)
            if len(set(st["steps"])) != 1:
                raise RuntimeError("ClipAdam(capturable=True): the parameters of a group must share one step count")
            t = st["steps"][0] + 1
            st["steps"] = [t] * len(allp)
            # exactly what cnr_clip_adam_step forms from its by-value arguments: the betas as float32 values, the powers in double, the results as float32
            b1, b2 = (float(torch.tensor(b, dtype=torch.float32)) for b in group["betas"])
            vals = torch.tensor([float(group["lr"]), 1.0 - b1 ** t, (1.0 - b2 ** t) ** 0.5], dtype=torch.float64).float()
            if gi not in self._hyper:
                self._hyper[gi] = (torch.empty(3, dtype=torch.float32, device=dev), PinnedStager())
            buf, stager = self._hyper[gi]
            buf.copy_(stager.to_device(vals, dev))
        self._prepared = True

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            allp = list(group["params"])
            if not any(p.grad is not None for p in allp):
                continue
            for p in allp:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("ClipAdam: parameters must be contiguous float32")
            # The moments of a group live in two flat buffers laid out over ALL its parameters in group order (offsets fixed for the life
            # of the optimiser), so a parameter that receives a gradient only on some steps keeps its moments and its own step count --
            # like torch.optim.Adam's per-parameter state.  The state hangs off the group's first parameter (state_dict-compatible).
            _, st, dev = self._group_state(gi, group)
            sizes_all = [p.numel() for p in allp]
            total = sum(sizes_all)
            if self.capturable and gi not in self._hyper:
                raise RuntimeError("ClipAdam(capturable=True): call prepare_step() before step()")
            if "steps" not in st and "step" in st:
                # state saved by an earlier build: one step count for the group, moments laid out over the parameters that had gradients.
                # Loadable when that was every parameter of the group (the renderer's case); otherwise the offsets are not recoverable.
                if st["exp_avg"].numel() != total:
                    raise RuntimeError("ClipAdam: cannot migrate a state_dict of the earlier layout whose moments cover only a subset of the "
                                       "group's parameters (%d of %d elements); restart the optimiser state" % (st["exp_avg"].numel(), total))
                st["steps"] = [int(st.pop("step"))] * len(allp)
            if st["exp_avg"].numel() != total or len(st["steps"]) != len(allp):
                raise RuntimeError("ClipAdam: the parameter list of a group changed after the first step")
            b1, b2 = group["betas"]
            mn = group.get("max_norm")
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
            offs = [0]
            for z in sizes_all:
                offs.append(offs[-1] + z)
            # runs of consecutive parameters that have a gradient and share a step count: one library call each (normally one per group)
            i = 0
            while i < len(allp):
                if allp[i].grad is None:
                    i += 1
                    continue
                j = i
                while j + 1 < len(allp) and allp[j + 1].grad is not None and st["steps"][j + 1] == st["steps"][i]:
                    j += 1
                ps = allp[i:j + 1]
                if not self.capturable:
                    for k in range(i, j + 1):
                        st["steps"][k] += 1
                cfg = _lib.CnrAdamConfig(lr=float(group["lr"]), beta1=float(b1), beta2=float(b2), eps=float(group["eps"]),
                                         max_norm=float(mn) if mn else 0.0, step=max(1, int(st["steps"][i])),
                                         hyper_dev=self._hyper[gi][0].data_ptr() if self.capturable else None)
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
                n = len(ps)
                sz = (C.c_int64 * n)(*sizes_all[i:j + 1])
                pw = (C.c_void_p * n)(*[p.data_ptr() for p in ps])
                pg = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
                nb = self._lib.lib.cnr_clip_adam_scratch_bytes(n, sz)
                key = (gi, str(dev))
                if key not in self._scratch or self._scratch[key].numel() < nb:   # scratch is not optimiser state: kept out of state_dict
                    self._scratch[key] = torch.empty(nb, dtype=torch.uint8, device=dev)
                rc = self._lib.lib.cnr_clip_adam_step(C.byref(cfg), n, sz, pw, pg, C.c_void_p(st["exp_avg"].data_ptr() + 4 * offs[i]),
                                                      C.c_void_p(st["exp_avg_sq"].data_ptr() + 4 * offs[i]),
                                                      C.c_void_p(self._scratch[key].data_ptr()), nb, stream)
                self._lib.check(rc, "cnr_clip_adam_step")
                i = j + 1
        return loss
