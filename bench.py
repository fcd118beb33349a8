#!/usr/bin/env python3
"""Benchmark of the Color-NeuS render hot path on MI355X.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N ...          starts the N ranks itself (a child torch.distributed.run, before any GPU call); same as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    ... --scaling strong --rays-total 4096      BASELINE config C4: a fixed 4096-ray batch split over the N ranks (512 per GPU at N = 8)

A step = forward + loss + backward (+ RCCL gradient all-reduce for N > 1) + per-parameter clip + Adam of the Color_NeuS DTU
renderer block (config/Color_NeuS_dtu.yml: 8x256 SDF net, 4x256 colour net, 4x256 relight net, 64 coarse + 64 importance
samples in 4 up-sampling steps) over one batch of synthetic rays of an 800x800 view; inputs resident in HBM.  Rays shard across
ranks (weak scaling by default: --rays is per GPU).  Rank 0 prints ONE JSON line.

Extra objects on the JSON line:
  roofline            dominant kernel: algorithmic HBM bytes per launch (operand matrices read once + outputs written once,
                      DESIGN.md section 4) / per-launch HIP-event time against the 8 TB/s HBM3E peak (hbm_frac), and its matrix-core
                      work (3 f16 MFMAs per fp32 product) against the 2.5 PFLOP/s dense f16 peak (mfma_frac).  Measured in a second,
                      event-instrumented pass over the same steps.
  kernel_breakdown    per kernel family: ms per step, GB/s (algorithmic), TFLOP/s -- includes the sampler / compositor kernels.
  step_ms             per-step HIP-event times of the K timed steps (events on the launch stream, rank 0): median, p10, p90, and the
                      rate the median implies (SURVEY 8d protocol); `value` itself stays K steps over the barrier-to-barrier wall time.
  inference           forward-only rays/s (validate_image, NeuS_Trainer.py:216-277) without and with the early-termination compaction
                      (prune_eps 1e-4: wavefront ballot / popcount, north_star), bounded sample.
  c5                  BASELINE config 5 on this GPU: 512^3 SDF lattice (extract_fields), device marching cubes, 500 k vertex colours.
  small_batch         rays/s of the same step at 512 and 1024 rays per step (the reference trains at N_RAYS 1024; 512 is C4's per-GPU share).
  torch_gpu_baseline  the plain-PyTorch restatement of the reference algorithm (oracle/) on the same GPU, bounded sample: the
                      stand-in for "reference single-GPU PyTorch" (the reference's own Python cannot travel to the GPU box).
  cpu_baseline        the same restatement on the host cores (contract object: best thread count), plus cpu_baseline_1thread (the
                      reference pins OMP/MKL to one thread, train.py:4-8); warmed, bounded samples, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 64 FLOP/clk/SIMD
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense f16 / bf16 MFMA (no sparsity)
# What the matrix pipe ALONE sustains on this board when its operands are not constants (tools/mfma_power.py, profiles/r05_mfma_power.txt:
# v_mfma_f32_32x32x16_f16 on all 1024 SIMDs, random f16 operands: the clock falls to 1.8 GHz at 1240-1310 W; with constant operands 2471 TFLOP/s at 1004 W)
F16_MFMA_MEASURED_RANDOM_OPERANDS_TFLOPS = 1603.0
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (about 6.3 TB/s measured with a float4 copy)
SPLIT_F16_KERNELS = ("layer_gemm_ws", "layer_dw", "dw_gemm_hx", "chain_sdf_value")   # 3 f16 MFMAs per fp32-equivalent product


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=4096, help="rays per step per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--rays-total", type=int, default=4096, help="rays per step over ALL GPUs (--scaling strong)")
    ap.add_argument("--no-optim", action="store_true", help="time fwd+loss+bwd only (skip clip + Adam)")
    ap.add_argument("--torch-optim", action="store_true", help="torch._foreach clip + torch fused Adam instead of the library's one-launch clip+Adam")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-small-batch", action="store_true")
    ap.add_argument("--no-torch-gpu-baseline", action="store_true")
    ap.add_argument("--no-inference", action="store_true")
    ap.add_argument("--no-c5", action="store_true")
    ap.add_argument("--no-loss-only", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=512)
    ap.add_argument("--cpu-threads", type=int, default=16,
                    help="torch threads of the contract CPU baseline (16 was the fastest of {8,16,32,64,128} on the 2x64-core bench host)")
    ap.add_argument("--torch-loss", action="store_true", help="evaluate the loss with torch ops instead of the library's fused loss kernels")
    ap.add_argument("--graph", action="store_true", help="one GPU: replay the step as ONE HIP graph (color-neus_amd/graph.py) instead of enqueuing its ~65 launches from "
                                                         "the host.  Bit-identical, and measured NO faster (round 6, same-box A/B: 174.4 k against 174.7 k rays/s at 4096 rays, "
                                                         "140.1 k / 140.5 k at 512, 157.0 k / 157.6 k at 1024: the enqueued step already keeps the GPU 98 % busy), hence opt-in")
    return ap.parse_args()


def clip_per_parameter_(params, max_norm=1.0):
    """clip_gradient (lib/utils/net_utils.py:174-184) with torch multi-tensor ops (--torch-optim)."""
    grads = [p.grad for p in params if p.grad is not None]
    coefs = torch._foreach_norm(grads)
    torch._foreach_add_(coefs, 1e-6)
    torch._foreach_reciprocal_(coefs)
    torch._foreach_mul_(coefs, max_norm)
    torch._foreach_clamp_max_(coefs, 1.0)
    torch._foreach_mul_(grads, coefs)


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks as a CHILD `torch.distributed.run` (never an exec: this process
    has made no GPU call, and stays that way), let the children inherit stdout / stderr (rank 0 prints the JSON line) and exit with
    their status.  Replaces the reference's single-GPU assert (train.py:111) with a launcher that cannot run as fewer ranks than asked."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:   # a record must never claim a GPU count it did not run on
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus (or plainly as "
                         "`python bench.py --gpus N`, which starts the N ranks itself)" % (args.gpus, world))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # CNR_BENCH_EMU_LIB=<tests/_build/libcolorneus_emu.so>: TEST INFRASTRUCTURE ONLY (tests/test_bench_harness.py).  The harness -- rank set-up, ray
    # sharding, collectives, the JSON line -- then runs on CPU tensors with the CPU emulation of the kernel layer and gloo, so that the code the
    # driver launches on an 8-GPU node is exercised where no GPU exists.  The line says "emulation": true and is never a measurement.
    emu = os.environ.get("CNR_BENCH_EMU_LIB")
    if emu:
        backend = "gloo"
        dev = torch.device("cpu")
        torch.set_num_threads(2)
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the render path has no CPU fallback)"
        # CNR_BENCH_BACKEND=gloo lets the multi-rank code path be exercised on a box with fewer GPUs than ranks (ranks then share
        # devices, collectives go through the host); the measured configuration is always one rank per GPU over RCCL ("nccl")
        backend = os.environ.get("CNR_BENCH_BACKEND", "nccl")
        if backend == "nccl" and torch.cuda.device_count() < world:
            raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s): one rank per GPU" % (world, torch.cuda.device_count()))
        dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import color_neus_amd as cn
    from color_neus_amd import synthetic, parallel

    cfg = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0)   # Color_NeuS_dtu.yml
    torch.manual_seed(0)
    renderer = synthetic.make_trained_like_(cn.ColorNeuSRenderer(cfg, library=emu) if emu else cn.ColorNeuSRenderer(cfg)).to(dev)
    params0 = params = list(renderer.parameters())
    lib = cn.load_library(emu) if emu else cn.load_library()
    assert emu or lib.backend == "hip-gfx950"
    if emu:   # every extra leg measures the GPU: off
        args.no_roofline = args.no_small_batch = args.no_torch_gpu_baseline = args.no_inference = args.no_c5 = args.no_cpu_baseline = args.no_loss_only = True
    if args.torch_optim:
        opt = torch.optim.Adam(params, lr=5e-4, betas=(0.9, 0.99), fused=True)
    else:   # config/Color_NeuS_dtu.yml: adam, LR 5e-4, GRAD_CLIP NORM 1.0 TYPE 2 per parameter tensor
        opt = cn.ClipAdam(renderer._ordered_params(), lr=5e-4, betas=(0.9, 0.99), eps=1e-8, max_norm=1.0, library=lib,
                          capturable=not emu and world == 1 and args.graph)
    opt0 = opt
    # --graph (one GPU): the step (ray generation -> forward -> fused loss -> backward -> clip + Adam: ~65 launches on fixed addresses) is captured ONCE per
    # batch size in a HIP graph and replayed (color-neus_amd/graph.py); per step the host refreshes the pixel indices and the jitter draw (the CPU
    # generator is consumed exactly like the enqueued step: torch.rand([R, 1])) in static device buffers.  Same launches, same arithmetic, bit-identical
    # parameters (tests/test_graph_step.py).  Not the default: it measured no faster than the enqueued step (see --graph's help).
    use_graph = (not emu and world == 1 and args.graph and not args.torch_optim and not args.torch_loss and not args.no_optim)
    graphs = {}

    if args.scaling == "strong":
        if args.rays_total % world:
            raise SystemExit("--rays-total must be divisible by the number of ranks")
        R = args.rays_total // world
    else:
        R = args.rays
    M = cfg.n_total
    strong = args.scaling == "strong"
    # weak scaling: every rank renders rays of its own view; strong scaling (BASELINE C4): ONE batch of --rays-total rays of one view, rank r
    # takes rows [r R, (r + 1) R) of it and of its jitter draw -- the N-rank step is then the single-process step on the same batch.
    # The rays of a step come from the library's ray generator (cnr_gen_rays: rays, near / far, colours and mask values of the chosen pixels
    # in ONE launch) -- the step the reference's trainer runs in front of the renderer (NeuS_Trainer.render, NeuS_Trainer.py:104-120:
    # get_rays_multicam + near_far_from_sphere), not a gather from a precomputed table of all 640 000 rays.
    from color_neus_amd import rays as raygen
    Hh = Ww = 800
    cams = {}

    def camera(strong_):   # weak: every rank its own view; strong: ONE view on every rank
        seed = 1 if strong_ else 1 + rank
        if seed not in cams:
            cams[seed] = synthetic.synthetic_camera(Hh, Ww, seed=seed, device=dev)
        return cams[seed]

    cam_c2w, cam_focal, cam_image, cam_mask = camera(strong)
    n_all = Hh * Ww
    perm = torch.randperm(n_all, generator=torch.Generator().manual_seed(7)).to(dev)

    def batch(i, r, strong_=None, solo=False):
        strong_ = strong if strong_ is None else strong_
        c2w_, focal_, image_, mask_ = camera(strong_)
        if strong_ and not solo:
            rg = r * world
            idx = perm[(i * rg) % (n_all - rg):(i * rg) % (n_all - rg) + rg][rank * r:(rank + 1) * r]
        else:
            idx = perm[(i * r) % (n_all - r):(i * r) % (n_all - r) + r]
        o, d, rgb, msel, near, far = raygen._generate(lib, idx, r, c2w_, focal_, Hh, Ww, True, False, image=image_, mask=mask_,
                                                      origin=None, radius=1.0, want_nearfar=True)
        return o, d, near, far, rgb, msel

    torch.manual_seed(2)   # jitter stream (CPU generator, like the reference)

    def graphed_step(i, r):
        g = graphs.get(r)
        if g is None:
            from color_neus_amd.graph import GraphedStep, PinnedStager
            static = {"idx": torch.empty(r, dtype=torch.int64, device=dev), "t": torch.empty(r, 1, dtype=torch.float32, device=dev)}
            state = {"i": i, "stager": PinnedStager()}
            c2w_, focal_, image_, mask_ = camera(False)

            def refresh():   # what changes from step to step: the chosen pixels and the jitter draw of the CPU generator
                ii = state["i"]
                static["idx"].copy_(perm[(ii * r) % (n_all - r):(ii * r) % (n_all - r) + r])
                static["t"].copy_(state["stager"].to_device(torch.rand([r, 1]), dev))

            def fn(idx, t):
                o, d, rgb, msel, near, far = raygen._generate(lib, idx, r, c2w_, focal_, Hh, Ww, True, False, image=image_, mask=mask_,
                                                              origin=None, radius=1.0, want_nearfar=True)
                out = renderer(o, d, near, far, t_rand=t)
                loss, _ = cn.compute_loss_fused(out, rgb, msel, library=lib)
                for p in params0:
                    p.grad = None
                loss.backward()
                opt0.step()
                return loss

            g = graphs[r] = (GraphedStep(fn, static, optimizer=opt0, warmup=2, before_each=refresh), state)
        gs, state = g
        state["i"] = i
        return gs.replay()

    def step(i, r=None, alt=None, outputs="dict", strong_=None, solo=False, eager=False):
        """one optimisation step of r rays on this rank.  strong_: the ranks split ONE batch of r * world rays (BASELINE C4) instead of rendering
        r rays of their own view each; solo: this rank alone, no collective (the one-GPU reference of the strong-scaling side leg)"""
        strong_ = strong if strong_ is None else strong_
        r = R if r is None else r
        if use_graph and not eager and alt is None and outputs == "dict" and not strong_:
            return graphed_step(i, r)
        nw = 1 if solo else world
        rg = r * nw
        o, d, near, far, gt, mask = batch(i, r, strong_, solo)
        rnd, params, opt = alt if alt is not None else (renderer, params0, opt0)
        M = rnd.rcfg.n_total
        if strong_ and nw > 1:   # the whole batch's jitter draw, this rank's rows (every rank consumes the CPU generator like one process would)
            out = rnd(o, d, near, far, training_outputs=outputs, t_rand=parallel.draw_jitter(rg, rank, world, "cpu"))
        else:
            out = rnd(o, d, near, far, training_outputs=outputs)
        if args.torch_loss:    # the torch restatement of compute_loss (and its sharded counterpart)
            if nw == 1:
                loss, _ = cn.compute_loss(out, gt, mask)
            else:
                loss, _ = parallel.sharded_loss(out, gt, mask, n_rays_global=rg, n_samples=M)
        else:                  # loss kernels of the render library; ray-sharded runs all-reduce 5 floats between their two phases
            loss, _ = cn.compute_loss_fused(out, gt, mask, n_rays_global=rg if nw > 1 else None, library=lib)
        for p in params:
            p.grad = None
        loss.backward()
        if nw > 1:
            parallel.allreduce_gradients(params)   # one in-place RCCL all-reduce of the flat gradient bucket
        if not args.no_optim:
            if args.torch_optim:
                clip_per_parameter_(params)
            if getattr(opt, "capturable", False):
                opt.prepare_step()
            opt.step()
        return loss

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            if dev.type == "cuda":
                torch.cuda.synchronize(dev)

    def timed(nsteps, warmup, r=None, alt=None, per_step=None, outputs="dict", strong_=None, solo=False, active=True):
        """active=False: this rank takes no step (a solo leg of another rank) but meets the others at the barriers and in the max-reduction"""
        loss = None
        for i in range(warmup if active else 0):
            step(i, r, alt, outputs, strong_, solo)
        sync()
        # per-step HIP events on the stream every kernel of the step is launched on (torch's current stream): nsteps + 1 marks
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(nsteps + 1)] if (per_step is not None and dev.type == "cuda") else None
        t0 = time.perf_counter()
        for i in range(nsteps if active else 0):
            if marks:
                marks[i].record()
            loss = step(warmup + i, r, alt, outputs, strong_, solo)
        if marks:
            marks[nsteps].record()
        sync()
        dt = time.perf_counter() - t0
        if marks:
            per_step.extend(marks[i].elapsed_time(marks[i + 1]) for i in range(nsteps))
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), loss

    def side_rate(nsteps, r, alt=None, outputs="dict"):
        """rays/s of a side leg (small batches, C2, loss_only): from the MEDIAN per-step event time on one GPU -- these legs are 0.1-0.4 s long,
        and one host hiccup in them moved a mean-based figure by 30 % between otherwise identical runs -- from the total time otherwise"""
        ts = []
        dts, _ = timed(nsteps, 5, r, alt, per_step=ts, outputs=outputs)
        if world == 1 and len(ts) == nsteps:
            ts.sort()
            return (r if r is not None else R) * 1e3 / ts[len(ts) // 2]
        return (r if r is not None else R) * world * nsteps / dts

    step_times = []
    dt, loss = timed(args.steps, args.warmup, per_step=step_times)
    Rg = R * world
    ms_per_step = dt / args.steps * 1e3
    value = Rg * args.steps / dt

    optim_name = "" if args.no_optim else ("+torch-clip+torch-adam" if args.torch_optim else "+fused-clip-adam")
    result = {
        "metric": "rays/sec (fwd+bwd) at 128 samples/ray", "value": round(value, 1), "unit": "rays/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
        "dtype_note": "fp32 tensors and fp32 accumulation; matrix products through error-free f16 hi/lo splits of the fp32 operands "
                      "(3 MFMAs per product, exact power-of-two scaling), parity gate 1e-4 relative",
        "data": "synthetic", **({"emulation": True} if emu else {}),
        "config": {"workload": "Color_NeuS_dtu.yml renderer block (SDF 8x256 + colour 4x256 + relight 4x256), synthetic 800x800 "
                               "view, %d rays/step/GPU x (64+64) samples, trained-like weights" % R,
                   "rays_per_step_per_gpu": R, "rays_per_step_total": Rg, "samples_per_ray": M, "parallelism": "ray-sharded dp%d" % world,
                   "step": "fwd+" + ("torch-loss" if args.torch_loss else "fused-loss") + "+bwd" + optim_name +
                           (("+rccl-allreduce" if backend == "nccl" else "+%s-allreduce" % backend) if world > 1 else "") +
                           ("; replayed as one HIP graph" if use_graph else ""),
                   "final_loss": float(loss.detach())},
    }

    # ---- N > 1: BOTH scaling figures from one invocation (the driver's command is fixed: `bench.py --gpus N`).  The timed leg above is the
    # line's `scaling` (weak by default: --rays per GPU, every rank its own view); this side leg is the other one.  For a weak main leg: BASELINE
    # config C4 -- ONE batch of --rays-total rays (4096) of one view and its jitter draw split over the ranks (512 rays per GPU at 8 GPUs), same
    # protocol (barrier + synchronise on both sides, max over ranks), next to the SAME batch as one step on one GPU of this node (rank 0 alone, the
    # other ranks wait at the barrier), so that the strong-scaling efficiency does not need a second invocation.
    if world > 1 and not strong and args.rays_total % world == 0:
        n_s = max(2, min(args.steps, 60))
        w_s = max(1, min(args.warmup, 10))
        rs = args.rays_total // world
        dts, loss_s = timed(n_s, w_s, r=rs, strong_=True)
        side = {"scaling": "strong", "rays_total": args.rays_total, "rays_per_gpu": rs, "steps": n_s, "warmup": w_s,
                "value": round(args.rays_total * n_s / dts, 1), "unit": "rays/s", "ms_per_step": round(dts / n_s * 1e3, 3),
                "final_loss": float(loss_s.detach())}
        # the same batch on ONE GPU of this node: rank 0 alone (no collective inside the step), everybody meets at the barriers of timed()
        dt1, _ = timed(n_s, w_s, r=args.rays_total, strong_=True, solo=True, active=rank == 0)
        side["one_gpu_same_batch"] = {"value": round(args.rays_total * n_s / dt1, 1), "ms_per_step": round(dt1 / n_s * 1e3, 3)}
        side["speedup_vs_one_gpu_same_batch"] = round(dt1 / dts, 3)
        side["efficiency_vs_n1_same_batch"] = round(dt1 / dts / world, 4)
        result["strong_scaling"] = side

    if step_times:
        st = sorted(step_times)
        pick = lambda q: st[min(len(st) - 1, int(q * len(st)))]
        result["step_ms"] = {"median": round(pick(0.5), 3), "p10": round(pick(0.1), 3), "p90": round(pick(0.9), 3), "n": len(st),
                             "rays_per_s_at_median": round(R / (pick(0.5) * 1e-3) * world, 1),
                             "note": "HIP events on the launch stream around each timed step (rank 0); includes the host's launch gaps"}

    # ---- roofline of the dominant kernel: event-instrumented pass over the same steps (rank 0)
    if not args.no_roofline:
        # every rank runs the same extra steps (they contain collectives when N > 1); only rank 0 records events
        lib.timing_enable(rank == 0)
        nrep = min(args.steps, 3)
        for i in range(nrep):
            step(args.warmup + args.steps + i, eager=True)   # (per-launch events: the enqueued form of the step)
        torch.cuda.synchronize(dev)
        recs = lib.timing_collect() if rank == 0 else []
        lib.timing_enable(False)
    if not args.no_roofline and rank == 0:
        agg = {}     # kernel name -> [ms, flop, launches, algorithmic HBM bytes]
        for name, kind, nt, P, N, K, pairs, ms, nbytes in recs:
            a = agg.setdefault(name, [0.0, 0.0, 0, 0.0])
            a[0] += ms
            a[1] += 2.0 * P * N * K * max(pairs, 1) if kind != 2 else 0.0
            a[2] += 1
            a[3] += nbytes
        tot_ms = sum(a[0] for a in agg.values())
        dom_name = max(agg, key=lambda k: agg[k][0])
        dom = agg[dom_name]
        desc = {"layer_gemm_ws": "layer_gemm_ws_kernel (weight-stationary layer GEMM: 256x256 layer held in registers as two f16 "
                                 "planes, 3 f16 MFMA 32x32x16 per product with exact power-of-two row scaling, points streamed "
                                 "HBM->LDS->MFMA->HBM with fused prologue/epilogue)",
                "layer_dw": "layer_dw_kernel (backward layer GEMM + the weight gradient of the same layer in one launch: column halves, 4 product waves "
                            "with the weights in registers + 4 waves that stage the input tile and accumulate dW from the on-chip operands, 3 f16 MFMA per product)",
                "layer_gemm": "layer_gemm_kernel (FP32 MFMA 32x32x2, 128-point tile)",
                "dw_gemm": "dw_gemm_kernel (FP32 MFMA weight-gradient GEMM, output-stationary)"}.get(dom_name, dom_name)
        gbs = dom[3] / (dom[0] * 1e-3) / 1e9
        tfl = dom[1] / (dom[0] * 1e-3) / 1e12
        f16_tfl = 3.0 * tfl if dom_name in SPLIT_F16_KERNELS else None
        # Every layer launch streams its operand matrices once (1-4 KB per point in, 1-2 KB out) against 131 kFLOP per point: the
        # launch is bound by HBM bytes, not by the matrix pipe -- both fractions are reported so that the claim can be checked.
        hbm_frac = gbs / HBM_PEAK_GBS
        mfma_frac = f16_tfl / F16_MFMA_PEAK_TFLOPS if f16_tfl else tfl / FP32_MFMA_PEAK_TFLOPS
        # the bound is whichever resource the launch uses the larger share of (computed, not assumed): frac / achieved / peak / unit follow it
        if hbm_frac >= mfma_frac:
            head = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4)}
        else:
            head = {"bound": "mfma", "achieved": round(f16_tfl or tfl, 1), "peak": F16_MFMA_PEAK_TFLOPS if f16_tfl else FP32_MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(mfma_frac, 4)}
        roof = {**head,
                "hbm_frac": round(hbm_frac, 4), "hbm_achieved_gbs": round(gbs, 1),
                "mfma_frac": round(mfma_frac, 4),
                "mfma_achieved_tflops": round(f16_tfl, 1) if f16_tfl else round(tfl, 1),
                "mfma_peak_tflops": F16_MFMA_PEAK_TFLOPS if f16_tfl else FP32_MFMA_PEAK_TFLOPS,
                **({"mfma_frac_of_measured_ceiling": round(f16_tfl / F16_MFMA_MEASURED_RANDOM_OPERANDS_TFLOPS, 4),
                    "mfma_measured_ceiling_note": "the matrix pipe alone, random f16 operands, whole chip: %.0f TFLOP/s at 1.8 GHz / 1240-1310 W "
                                                  "(tools/mfma_power.py; 2471 TFLOP/s only with constant operands)" % F16_MFMA_MEASURED_RANDOM_OPERANDS_TFLOPS} if f16_tfl else {}),
                "mfma_note": "f16 MFMA FLOP/s actually issued (3 per fp32-equivalent product) against the 2.5 PFLOP/s dense f16 peak" if f16_tfl
                             else "FP32 MFMA FLOP/s against the FP32 matrix peak",
                "fp32_equiv_tflops": round(tfl, 2),
                "hbm_note": "peak = 8 TB/s spec; a copy kernel reaches 6.3 TB/s on this chip and the layer kernel's own access pattern (1 KB rows in "
                            "16-byte pieces, reads and writes mixed) 5.2 TB/s with its MFMAs removed (tools/probes/ws_spec_ablate.hip, DESIGN.md 4.1)"}
        # HBM bytes per launch from the PMC counters cannot be collected inside this process; they come from the committed summary
        # of the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (tools/pmc_traffic.py)
        traffic, tfile = None, None
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            cand = os.path.join(ROOT, "profiles", "%s_pmc_hbm_traffic_%drays.json" % (rnd, R))
            if os.path.exists(cand):
                t = json.load(open(cand))["kernels"].get(dom_name, {}).get("hbm_bytes_per_launch")
                if t:
                    traffic, tfile = t, cand
                    break
        roof.update({"kernel": desc, "kernel_name": dom_name, "traffic": traffic, "traffic_source": os.path.relpath(tfile, ROOT) if traffic else None, "avg_launch_ms": round(dom[0] / dom[2], 4),
                     "algorithmic_bytes_per_launch": round(dom[3] / dom[2]), "launches_per_step": dom[2] // nrep,
                     "share_of_kernel_time": round(dom[0] / tot_ms, 3)})
        result["roofline"] = roof
        always = ("sampler_step", "upsample", "merge", "composite_fwd", "composite_bwd", "chain_sdf_value", "chain_fwd", "clip_adam")   # north_star evidence: sampler / compositor GB/s
        top = sorted(agg.items(), key=lambda kv: -kv[1][0])
        keep = [kv for i, kv in enumerate(top) if i < 8 or kv[0] in always]
        result["kernel_breakdown"] = [{"kernel": k, "ms_per_step": round(v[0] / nrep, 4),
                                       "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1) if v[1] else None,
                                       "gbs": round(v[3] / (v[0] * 1e-3) / 1e9, 1) if v[3] else None,
                                       "hbm_frac": round(v[3] / (v[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if v[3] else None,
                                       "launches": v[2] // nrep}
                                      for k, v in keep]
        result["kernel_ms_per_step"] = round(tot_ms / nrep, 3)
        result["launches_per_step"] = sum(v[2] for v in agg.values()) // nrep

    # ---- the same step at the reference's own batch (N_RAYS 1024, config/Color_NeuS_dtu.yml:13) and at C4's per-GPU share (512)
    if not args.no_small_batch:
        small = {}
        for r in (512, 1024):
            if r == R:
                small[str(r)] = round(value, 1)
                continue
            n = max(20, min(args.steps, 60))
            small[str(r)] = round(side_rate(n, r), 1)
        result["small_batch"] = {"unit": "rays/s", "rays_per_step_per_gpu": small}
        # launches of a 512-ray step (the fixed per-launch costs are what separates the small-batch rate from the 4096-ray rate)
        if rank == 0 and world == 1:
            lib.timing_enable(True)
            step(0, 512, eager=True)
            torch.cuda.synchronize(dev)
            result["small_batch"]["launches_per_step_512"] = len(lib.timing_collect())
            lib.timing_enable(False)
        # BASELINE config 2 as written: DTU network, 512 rays per batch x 64 samples, no importance sampling
        cfg2 = cn.RenderConfig(type="Color_NeuS", col_mode="no_view_dir", col_d_in=6, col_multires_view=0, n_samples=64, n_importance=0)
        r2 = cn.ColorNeuSRenderer(cfg2).to(dev)
        r2.load_state_dict(renderer.state_dict())
        opt2 = (torch.optim.Adam(list(r2.parameters()), lr=5e-4, betas=(0.9, 0.99), fused=True) if args.torch_optim
                else cn.ClipAdam(r2._ordered_params(), lr=5e-4, betas=(0.9, 0.99), eps=1e-8, max_norm=1.0, library=lib))
        alt = (r2, list(r2.parameters()), opt2)
        n = max(20, min(args.steps, 60))
        result["small_batch"]["c2_512rays_x_64samples_no_importance"] = round(side_rate(n, 512, alt), 1)
        result["small_batch"]["note"] = "side legs of 20-60 steps: rate at the median per-step event time (one GPU), total time over ranks otherwise"

    # ---- the same step with training_outputs="loss_only" (no [R][M][3] dict tensors, the relight term from per-ray sums of
    # the compositor); not the default because the reference's trainer consumes the dict
    if not args.no_loss_only and not args.torch_loss:
        lo = {}
        for r in sorted({R, 1024}):
            n = max(20, min(args.steps, 60))
            lo[str(r)] = round(side_rate(n, r, None, "loss_only"), 1)
        result["loss_only"] = {"unit": "rays/s", "rays_per_step_per_gpu": lo,
                               "note": "same step (fwd + fused loss + bwd + clip + Adam) with renderer(..., training_outputs='loss_only')"}

    # ---- inference use of the path (validate_image, NeuS_Trainer.py:216-277): forward only, EVAL-style chunks.  Under torch.no_grad() the module
    # takes the library's forward-only entry point (cnr_render_forward_only: bit-identical values, nothing kept for a backward pass); reported
    # next to it: the saving forward on the same rays (what an inference call cost before round 5) and the early-termination compaction
    # (colour / relight stacks only on samples with weight >= eps: ballot + popcount index list, chain-fused launch reads its rows through it)
    if not args.no_inference and rank == 0:
        Ri = 8192
        o, d, near, far, _, _ = batch(0, Ri)
        inf = {}
        with torch.no_grad():
            ref = renderer(o, d, near, far, perturb_overwrite=0, forward_only=False)
            for tag, kw in (("saving_forward", dict(forward_only=False)), ("prune_eps_0", dict()), ("prune_eps_0.0001", dict(prune_eps=1e-4))):
                for _ in range(2):
                    out = renderer(o, d, near, far, perturb_overwrite=0, **kw)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(5):
                    out = renderer(o, d, near, far, perturb_overwrite=0, **kw)
                torch.cuda.synchronize(dev)
                dti = (time.perf_counter() - t1) / 5
                eps = kw.get("prune_eps", 0.0)
                inf[tag] = {"rays_per_s": round(Ri / dti, 1), "ms": round(dti * 1e3, 2),
                            "kept_fraction": round(float((out["weights"] >= eps).float().mean()), 3) if eps > 0 else 1.0,
                            "max_abs_color_diff": float((out["color_fine"] - ref["color_fine"]).abs().max())}
            lib.timing_enable(True)
            renderer(o, d, near, far, perturb_overwrite=0)
            torch.cuda.synchronize(dev)
            recs = lib.timing_collect()
            lib.timing_enable(False)
            fam = {}
            for name, kind, nt, P_, N_, K_, pairs, ms, nbytes in recs:
                fam[name] = round(fam.get(name, 0.0) + ms, 4)
            inf["forward_only_kernel_ms"] = dict(sorted(fam.items(), key=lambda kv: -kv[1])[:8])
            # a whole 800 x 800 view (validate_image: all rays of one camera in chunks, colour and depth kept): 79 chunks of 8192 rays
            vo, vd = raygen.get_rays_at(cam_c2w[0], cam_focal, Hh, Ww, normalize=True, library=lib)
            vo, vd = vo.reshape(-1, 3), vd.reshape(-1, 3)
            vn, vf = raygen.near_far_from_sphere(vo, vd)
            # (chunks of 65536 rays: 113 GB of scratch -- what 288 GB of HBM are for: fewer, larger launches, +8 % over 8192-ray chunks)
            big = 65536 if torch.cuda.mem_get_info(dev)[0] > 170e9 else Ri
            imgs = {}
            for tag, ch, kw in ((("view_800x800_s", Ri, dict()), ("view_800x800_pruned_s", Ri, dict(prune_eps=1e-4)),
                                 ("view_800x800_chunk%d_s" % big, big, dict()), ("view_800x800_chunk%d_pruned_s" % big, big, dict(prune_eps=1e-4)))
                                if world == 1 else ()):   # (the chunk loop gathers over the process group)
                parallel.sharded_render_image(renderer, vo[:ch], vd[:ch], vn[:ch], vf[:ch], chunk=ch, perturb_overwrite=0, **kw)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                img = parallel.sharded_render_image(renderer, vo, vd, vn, vf, chunk=ch, perturb_overwrite=0, **kw)
                torch.cuda.synchronize(dev)
                inf[tag] = round(time.perf_counter() - t1, 3)
                # the 65536-ray chunks (P x 256 = 2^31 elements per tensor) must render the image of the 8192-ray chunks: per-ray arithmetic does not
                # depend on the chunking (no jitter here), so any difference would be an addressing fault at that size, not round-off
                imgs[tag] = {k: v.detach().clone() for k, v in img.items()}
            if "view_800x800_s" in imgs and "view_800x800_chunk%d_s" % big in imgs and big != Ri:
                # (bit patterns: a NaN pixel compares equal to the same NaN pixel)
                same = {k: bool(torch.equal(v.contiguous().view(torch.int32), imgs["view_800x800_chunk%d_s" % big][k].contiguous().view(torch.int32)))
                        for k, v in imgs["view_800x800_s"].items()}
                inf["chunk%d_image_equals_chunk%d_image" % (big, Ri)] = all(same.values())
                inf["view_nonfinite_pixels"] = int(sum((~torch.isfinite(v)).reshape(v.shape[0], -1).any(dim=1).sum() for v in imgs["view_800x800_s"].values()))
                if not all(same.values()):
                    a_, b_ = imgs["view_800x800_s"]["color_fine"], imgs["view_800x800_chunk%d_s" % big]["color_fine"]
                    bad = torch.nonzero((a_ != b_).reshape(a_.shape[0], -1).any(dim=1)).reshape(-1)
                    # (recorded, not raised: a side leg must not cost the run its bench line; tests/test_forward_only.py holds the property on the GPU suite)
                    inf["chunk_image_mismatch"] = ("the %d-ray chunks render a different image than the %d-ray chunks: %s; %d rays differ, first %s last %s, max abs %.3e"
                                                   % (big, Ri, same, len(bad), bad[:4].tolist(), bad[-4:].tolist(), float((a_ - b_).abs().max())))
                    print("bench.py: " + inf["chunk_image_mismatch"], file=sys.stderr)
            imgs.clear()
            torch.cuda.empty_cache()
            del vo, vd, vn, vf
            inf["scratch_GB"] = {"forward_only": round(lib.lib.cnr_infer_scratch_bytes(__import__("ctypes").byref(renderer._ccfg), Ri) / 1e9, 2),
                                 "saving": round(lib.lib.cnr_ctx_bytes(__import__("ctypes").byref(renderer._ccfg), Ri) / 1e9, 2)}
        result["inference"] = {"unit": "rays/s", "sample": "5 forward passes of %d rays x 128 samples, no jitter, DTU renderer block; prune_eps_0 = the forward-only "
                                                           "entry point (what a no_grad call takes), saving_forward = cnr_render_forward on the same rays" % Ri, **inf}

    # ---- BASELINE config 5 (evaluation.py -rr 512): dense SDF lattice + device marching cubes + vertex colours, one pass each
    if not args.no_c5 and rank == 0:
        res_, nv = 512, 500000
        renderer.extract_fields([-1.01] * 3, [1.01] * 3, dev, 64)   # warm-up
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        u = renderer.extract_fields([-1.01] * 3, [1.01] * 3, dev, res_)
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        mv, mt = renderer.marching_cubes(u, [-1.01] * 3, [1.01] * 3, 0.0)
        torch.cuda.synchronize(dev)
        t3 = time.perf_counter()
        gv = torch.Generator().manual_seed(3)
        v = torch.randn(nv, 3, generator=gv)
        v = (v / v.norm(dim=-1, keepdim=True) * (0.5 + 0.02 * torch.randn(nv, 1, generator=gv))).numpy()
        renderer.extract_color(v[:1000], dev)
        torch.cuda.synchronize(dev)
        t4 = time.perf_counter()
        renderer.extract_color(v, dev)
        t5 = time.perf_counter()
        result["c5"] = {"lattice": "%d^3" % res_, "lattice_s": round(t2 - t1, 3), "lattice_Mpts_per_s": round(res_ ** 3 / (t2 - t1) / 1e6, 1),
                        "marching_cubes_ms": round((t3 - t2) * 1e3, 2), "mesh_vertices": int(mv.shape[0]), "mesh_triangles": int(mt.shape[0]),
                        "vertex_colours": nv, "vertex_colour_ms": round((t5 - t4) * 1e3, 2),
                        "note": "extract_fields / extract_geometry / extract_color (NeuS.py:14-64) on the device; vertex colours include the H2D / D2H of the caller's numpy arrays"}
        del u, mv, mt

    # ---- the plain-PyTorch restatement on the same GPU (the 'reference single-GPU PyTorch' stand-in) on the SURVEY 8d protocol:
    # R in {512, 1024, 4096}, 3 warm-up + 20 timed iterations each, HIP events around every iteration on the launch stream, median
    if not args.no_torch_gpu_baseline and rank == 0:
        from oracle import colorneus_oracle as O
        ocfg = O.dtu_config()
        P = {k: v.detach().clone().requires_grad_(True) for k, v in renderer.state_dict().items()}
        nt_, per_r = 20, {}
        for Rt in (512, 1024, 4096):
            o, d, near, far, gt, mask = [x[:Rt] for x in batch(0, max(R, Rt))]

            def tstep():
                t_rand = torch.rand(Rt, 1).to(dev)
                out = O.render(P, ocfg, o, d, near, far, t_rand=t_rand, reference_ops=True)
                l, _ = O.compute_loss(out, gt, mask)
                for p in P.values():
                    p.grad = None
                l.backward()
            for _ in range(3):
                tstep()
            torch.cuda.synchronize(dev)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(nt_ + 1)]
            for i in range(nt_):
                ev[i].record()
                tstep()
            ev[nt_].record()
            torch.cuda.synchronize(dev)
            ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(nt_))
            per_r[str(Rt)] = {"rays_per_s": round(Rt / (ts[nt_ // 2] * 1e-3), 1), "median_ms": round(ts[nt_ // 2], 2),
                              "p10_ms": round(ts[nt_ // 10], 2), "p90_ms": round(ts[(9 * nt_) // 10], 2)}
            del o, d, near, far, gt, mask
            torch.cuda.empty_cache()
        tv = max(v["rays_per_s"] for v in per_r.values())      # the stand-in's best batch size is what the >= 10x target divides by
        sb = result.get("small_batch", {}).get("rays_per_step_per_gpu", {})
        native = {"4096": value / world if R == 4096 else None, "1024": sb.get("1024"), "512": sb.get("512")}
        result["torch_gpu_baseline"] = {"value": round(tv, 1), "unit": "rays/s", "per_batch": per_r,
                                        "sample": "3 warm-up + %d timed iterations at each of 512 / 1024 / 4096 rays x 128 samples (HIP events per iteration, median), plain "
                                                  "PyTorch-ROCm ops, the oracle restatement in its reference_ops mode (normals by a second SDF forward + "
                                                  "autograd.grad(create_graph=True) like fields.py:105-115), fwd+bwd, no optimiser" % nt_,
                                        "speedup_at_4096": round(value / world / tv, 2) if R == 4096 else None,
                                        "speedup_at_equal_batch": {k: round(native[k] / per_r[k]["rays_per_s"], 2) for k in per_r if native.get(k)}}

    # ---- CPU baseline: the oracle on the host cores, bounded samples (rank 0, N=1 only)
    if not args.no_cpu_baseline and rank == 0 and world == 1:
        from oracle import colorneus_oracle as O
        ocfg = O.dtu_config()
        P = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in renderer.state_dict().items()}
        ncpu = os.cpu_count() or 1

        def cpu_rate(threads, rays, budget_s, min_it, warm=True):
            torch.set_num_threads(threads)
            o, d, near, far, gt, mask = [x[:rays].cpu() for x in batch(0, max(R, rays))]

            def cstep():
                t_rand = torch.rand(rays, 1)
                out = O.render(P, ocfg, o, d, near, far, t_rand=t_rand, reference_ops=True)
                l, _ = O.compute_loss(out, gt, mask)
                for p in P.values():
                    p.grad = None
                l.backward()
            if warm:
                cstep()
            t1 = time.perf_counter()
            n_it = 0
            while n_it < min_it or (time.perf_counter() - t1 < budget_s and n_it < 40):
                cstep()
                n_it += 1
            return rays * n_it / (time.perf_counter() - t1), n_it

        cores = max(1, min(args.cpu_threads, ncpu))
        v, n_it = cpu_rate(cores, args.cpu_rays, 10.0, 3)
        result["cpu_baseline"] = {"value": round(v, 2), "unit": "rays/s", "cores": cores, "kind": "port",
                                  "sample": "%d iterations of %d rays x (64+64) samples, fwd+bwd, same network/weights, "
                                            "torch CPU ops with %d threads (host has %d logical CPUs)" % (n_it, args.cpu_rays, cores, ncpu)}
        v1, n1 = cpu_rate(1, 32, 8.0, 1)
        result["cpu_baseline_1thread"] = {"value": round(v1, 2), "unit": "rays/s", "cores": 1, "kind": "port",
                                          "sample": "%d iterations of 32 rays x (64+64) samples, one thread (the reference pins OMP/MKL to 1, train.py:4-8)" % n1}

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
