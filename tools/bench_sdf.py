"""SDF value-chain micro-benchmark (sdf_network.sdf on n points, DTU-size network): python tools/bench_sdf.py [n ...]"""
import sys, time, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _native as N
from oracle import colorneus_oracle as O
ocfg = O.dtu_config()
P = O.init_params(ocfg, seed=0, dtype=torch.float32, trained_like=True)
r = N.make_renderer(ocfg, P, None, "cuda:0")
sizes = [int(a) for a in sys.argv[1:]] or [8192, 65536, 262144, 1 << 21]
for n in sizes:
    pts = (torch.rand(n, 3) * 2 - 1).to("cuda:0")
    for _ in range(3): r.sdf(pts)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): r.sdf(pts)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("n=%8d  %.3f ms  %.1f Mpts/s  %.1f TFLOP/s-equiv" % (n, dt * 1e3, n / dt / 1e6, n / dt * 2 * 524544 / 1e12))
    lib = r._lib
    lib.timing_enable(True)
    r.sdf(pts)
    torch.cuda.synchronize()
    agg = {}
    for name, kind, nt, P, N, K, pairs, ms, nbytes in lib.timing_collect():
        a = agg.setdefault(name, [0.0, 0]); a[0] += ms; a[1] += 1
    lib.timing_enable(False)
    print("   per call:", ", ".join("%s %.3f ms x%d" % (k, v[0], v[1]) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])))
