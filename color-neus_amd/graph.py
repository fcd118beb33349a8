"""A whole training step as ONE HIP graph (round 6).

The render library only enqueues kernels on the caller's stream and allocates nothing (include/colorneus_render.h), the fused loss and
ClipAdam(capturable=True) likewise: ray generation -> renderer forward -> loss -> backward -> clip + Adam of a fixed batch size is a fixed
sequence of ~65 launches on fixed addresses, so it can be captured once (torch.cuda.graph = hipStreamBeginCapture on a side stream + a
private memory pool) and replayed per step with ONE host call.  What changes between steps travels through static device buffers that the
caller refreshes before each replay: the pixel indices, the per-ray jitter draw (the CPU generator is consumed exactly as in the eager step:
torch.rand([R, 1]), NeuS.py:325), the optimiser's step-dependent scalars.  At 512-1024 rays per step the eager step spends ~5 % of its time
in launch gaps and host work (profiles/r05_launch_gaps_512rays.txt); the replay has none.
"""
import torch


class PinnedStager:
    """Host -> device copies of small tensors without stalling the host: a plain .to(device) from pageable memory blocks until the stream has
    drained.  A ring of pinned buffers, an event per buffer guards its reuse (the copy issued `depth` calls ago has long finished)."""

    def __init__(self, depth=3):
        self.depth = depth
        self.ring = {}

    def to_device(self, t_cpu, dev):
        dev = torch.device(dev)
        if dev.type != "cuda":
            return t_cpu.to(dev)
        key = (tuple(t_cpu.shape), t_cpu.dtype, dev)
        st = self.ring.get(key)
        if st is None:
            st = {"i": 0, "buf": [torch.empty(t_cpu.shape, dtype=t_cpu.dtype, pin_memory=True) for _ in range(self.depth)], "ev": [None] * self.depth}
            self.ring[key] = st
        i = st["i"]
        st["i"] = (i + 1) % self.depth
        if st["ev"][i] is not None:
            st["ev"][i].synchronize()
        st["buf"][i].copy_(t_cpu)
        out = st["buf"][i].to(dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        st["ev"][i] = ev
        return out


class GraphedStep:
    """fn(**static) -> loss (a scalar tensor) captured in a HIP graph.

    ``static``: dict of DEVICE tensors at fixed addresses -- everything that differs from step to step enters ``fn`` through them; refresh them
    with ``.copy_()`` (stream-ordered, no host stall) and then call ``replay()``.  ``fn`` runs the whole step -- forward, loss, ``backward()``,
    ``optimizer.step()`` of a ClipAdam(capturable=True) -- and must not touch the host (no .item(), no pageable host -> device copies, no CPU
    generator draw: draw outside and copy into a static buffer).  ``warmup`` eager calls on a side stream come first (lazy one-time set-up:
    kernel attributes, optimiser state, scratch buffers), then the capture.  ``loss`` (static output) holds the last replay's value.
    """

    def __init__(self, fn, static, optimizer=None, warmup=2, before_each=None):
        dev = next(iter(static.values())).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedStep needs a GPU (HIP graph capture)")
        self.fn, self.static, self.optimizer, self.before_each = fn, static, optimizer, before_each
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._pre()
                fn(**static)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self._pre()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = fn(**static)
        # the capture itself ran nothing: the step whose scalars _pre() staged is the first replay's
        self._pending_pre = True

    def _pre(self):
        if self.before_each is not None:
            self.before_each()
        if self.optimizer is not None:
            self.optimizer.prepare_step()

    def replay(self):
        if self._pending_pre:
            self._pending_pre = False
        else:
            self._pre()
        self.graph.replay()
        return self.loss
