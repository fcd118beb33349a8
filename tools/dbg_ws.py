import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, _golden as G, _native as N
name = sys.argv[1] if len(sys.argv) > 1 else "dtu_sharp"
res = []
for it in range(2):
    fx, r, out, loss, grads, o, d = N.run_native(name, "jit", None, "cuda:0", fixed_z=True)
    res.append(({k: v.detach().cpu() for k, v in out.items()}, {k: v.detach().cpu() for k, v in grads.items()}, o.grad.cpu()))
print("mode", "WS off" if os.environ.get("CNR_DISABLE_WS") else "WS on")
for k in res[0][0]:
    a, b = res[0][0][k], res[1][0][k]
    ref = fx.get("jit:out_" + k)
    e = G.relerr(a.reshape(ref.shape), ref) if ref is not None else float("nan")
    print("out %-16s run-to-run equal %s   err vs golden %.2e" % (k, bool(torch.equal(a, b)), e))
bad = G.check_param_grads(fx, "jit", res[0][1], 1e-4)
print("bad grads:", bad[:6])
neq = [k for k in res[0][1] if not torch.equal(res[0][1][k], res[1][1][k])]
print("nondeterministic grads:", neq[:10])
