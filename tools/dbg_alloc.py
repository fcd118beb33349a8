import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, color_neus_amd as cn
from color_neus_amd import synthetic
dev = torch.device("cuda:0")
cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)
torch.manual_seed(0)
r = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg)).to(dev)
o_all, d_all, near_all, far_all, rgb_all, mask_all = synthetic.synthetic_view(seed=1, device=dev)
for R in (1024, 2048, 4096):
    o, d, n, f, gt, m = [x[:R] for x in (o_all, d_all, near_all, far_all, rgb_all, mask_all)]
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = r(o, d, n, f)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss, _ = cn.compute_loss(out, gt, m)
        for p in r.parameters(): p.grad = None
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        st = torch.cuda.memory_stats()
        print(R, it, "fwd %.1f ms bwd %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3), "reserved %.1f GB" % (st["reserved_bytes.all.current"]/2**30), "num_device_alloc", st.get("num_device_alloc"), "retries", st.get("num_alloc_retries"), flush=True)
