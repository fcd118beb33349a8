"""Two training steps in flight on two HIP streams of one device, issued from two threads: the library keeps no state between calls that a
result depends on (include/colorneus_render.h: every buffer is the caller's, all work is enqueued on the caller's stream), so each thread must
get, bit for bit, what it gets when it runs alone -- render forward / backward, the one-launch loss (its completion counter lives in the caller's
per-stream scratch since round 5; a library-global counter used to make overlapping loss launches lose their fold) and the forward-only call."""
import threading

import pytest
import torch

import _native as N

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(seed, R):
    from oracle import colorneus_oracle as O      # weights / rays only: inputs of both the serial and the concurrent run
    import test_edge_batches as T
    ocfg = O.dtu_config()
    P = O.init_params(ocfg, seed=seed, trained_like=True)
    o, d, near, far, t_rand, gt, mask = T._batch(R, 300 + seed)
    r = N.make_renderer(ocfg, P, None, DEV)
    z = O.sample_z(P, ocfg, o, d, near, far, t_rand)
    return r, [t.to(DEV) for t in (o, d, near, far, z, gt, mask)]


def _work(r, inp, stream, steps, out):
    import color_neus_amd as cn
    o, d, near, far, z, gt, mask = inp
    res = []
    with torch.cuda.stream(stream):
        for _ in range(steps):
            for p in r.parameters():
                p.grad = None
            rd = r(o, d, near, far, z_vals=z)
            loss, parts = cn.compute_loss_fused(rd, gt, mask)
            loss.backward()
            with torch.no_grad():
                fo = r(o, d, near, far, z_vals=z)                      # the forward-only entry point
            res.append((loss.detach().clone(), [p.grad.detach().clone() for p in r.parameters()], rd["color_fine"].detach().clone(),
                        fo["color_fine"].clone(), parts["eikonal_loss"].detach().clone()))
        stream.synchronize()
    out.extend(res)


def test_two_streams_two_threads_match_the_serial_runs():
    steps = 4
    jobs = [_setup(11, 130), _setup(12, 97)]
    serial = []
    for r, inp in jobs:
        acc = []
        _work(r, inp, torch.cuda.Stream(device=DEV), steps, acc)
        serial.append(acc)
    torch.cuda.synchronize()
    conc = [[], []]
    streams = [torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)]
    errs = []

    def run(i):
        try:
            _work(jobs[i][0], jobs[i][1], streams[i], steps, conc[i])
        except Exception as e:      # surface a failure of the worker thread in the test
            errs.append(e)
    ths = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    torch.cuda.synchronize()
    for i in range(2):
        assert len(conc[i]) == steps
        for s in range(steps):
            l0, g0, c0, f0, e0 = serial[i][s]
            l1, g1, c1, f1, e1 = conc[i][s]
            assert torch.equal(l0, l1) and torch.equal(c0, c1) and torch.equal(f0, f1) and torch.equal(e0, e1), (i, s)
            assert torch.equal(c1, f1), "forward-only == saving forward"
            for a, b in zip(g0, g1):
                assert torch.equal(a, b), (i, s)
