#!/usr/bin/env python3
"""Benchmark of the MV-KPConv hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic spheres whose RAW inputs are
already resident in HBM: first subsampling at dl + 5-level pyramid (HIP), for the fusion variants
2D encoder (PyTorch-ROCm) + depth unprojection + 3-NN + group_points + FeatureAggregation, KPFCNN
forward, loss, backward, gradient all-reduce (N > 1), clip, SGD step -- the step sequence of the
reference's trainer (utils/trainer.py:179-195) plus the input pyramid its DataLoader workers build.

    python bench.py [--gpus N --steps K --warmup W] [--workload early|baseline|middle|late]
                    [--spheres S] [--no-cpu-baseline]

N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL): spheres are sharded data
parallel (weak scaling: S spheres per GPU), gradients all-reduced in one flat bucket.
Rank 0 prints ONE JSON line (see DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="early", choices=["early", "baseline", "middle", "late"])
    ap.add_argument("--spheres", type=int, default=1, help="spheres per GPU per step")
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--in-radius", type=float, default=1.2, help="sphere radius (1.2 -> ~20 k points, 1.7 -> ~40 k)")
    ap.add_argument("--deformable", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dev", action="store_true",
                    help="development run: accept the knobs that change what a captured step contains (MVK_BENCH_DIAG, "
                         "MVK_BENCH_SKIP, MVK_BENCH_DUMMY_LAUNCHES); without it they are refused")
    ap.add_argument("--no-eager-line", action="store_true",
                    help="skip the eager drop-in step timing that goes to bench_detail.json")
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="replay the network part (fwd+loss+bwd+clip+SGD) as one hipGraph over capacity-padded levels")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    ap.add_argument("--cpu-baseline-steps", type=int, default=20,
                    help="timed iterations per CPU leg (each leg is also cut at a time bound)")
    return ap.parse_args()


def visible_gpu_count():
    """GPUs of this node as the kernel driver lists them (/sys/class/kfd topology nodes with SIMDs; CPUs are nodes
    without), cut to HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set; None when sysfs cannot tell (the ranks then
    find out themselves and fail loudly). No HIP or torch.cuda call: the launcher parent never touches the GPU runtime."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for path in nodes:
        try:
            with open(path) as f:
                props = dict(ln.split()[:2] for ln in f if len(ln.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None and v.strip() != "":
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process
    (python -m torch.distributed.run, one rank per GPU) before this process has touched the GPU, relay
    rank 0's JSON line and return the child's exit code. A process that has initialised HIP is never
    re-exec'ed."""
    import socket
    import subprocess
    one_gpu = os.environ.get("MVK_BENCH_ONE_GPU") == "1" or os.environ.get("MVK_BENCH_DRY") == "1"
    have = visible_gpu_count()                  # from sysfs: the parent makes no GPU-runtime call at all
    if not one_gpu and have is not None and have < args.gpus:
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, have), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print("bench.py: the %d-rank child run failed (exit %d)" % (args.gpus, proc.returncode), file=sys.stderr)
        return proc.returncode or 1
    res = json.loads(line)
    if res.get("n_gpus") != args.gpus or res.get("config", {}).get("ranks") != args.gpus:
        print("bench.py: asked for %d ranks, the run reports %s" % (args.gpus, res.get("n_gpus")), file=sys.stderr)
        return 3
    print(line)
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or "
                         "without a launcher: bench.py starts the ranks itself)" % (args.gpus, world, args.gpus))
    if os.environ.get("MVK_BENCH_DRY") == "1":
        return dry_run(args, world, rank)
    # rehearsal hook: MVK_BENCH_BACKEND=gloo + MVK_BENCH_ONE_GPU=1 runs N ranks on ONE card (development
    # only; the driver's multi-GPU runs use the defaults: RCCL, one rank per GPU)
    backend = os.environ.get("MVK_BENCH_BACKEND", "nccl")
    if os.environ.get("MVK_BENCH_ONE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # rehearsal hook: MVK_BENCH_FORCE_DP=1 takes the N > 1 code path (RCCL process group, eager all-reduce
    # between the two graphs) with a single rank, so that path can be exercised on a one-GPU box
    force_dp = world == 1 and os.environ.get("MVK_BENCH_FORCE_DP") == "1"
    if force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dp:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # everything runs on one non-default stream: autograd's AccumulateGrad nodes are bound to the stream
    # they were first used on, and nodes born on the legacy default stream cannot be captured later
    torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ.get("MVK_MAIN_PRIO", "0"))))

    if world > 1:
        # one MIOpen user database / kernel cache per rank: eight processes searching solvers at once otherwise
        # contend for the same SQLite files (lock warnings, serialised finds)
        os.environ.setdefault("MIOPEN_USER_DB_PATH", "/tmp/mvk_miopen_db_rank%d" % local)
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", "/tmp/mvk_miopen_cache_rank%d" % local)
        for k in ("MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR"):
            os.makedirs(os.environ[k], exist_ok=True)
    if os.environ.get("MVK_MIOPEN_DETERMINISTIC") == "1":      # development: only solvers without atomic split reductions
        torch.backends.cudnn.deterministic = True
    if os.environ.get("MVK_MIOPEN_BENCHMARK", "1") == "1":
        # MIOpen picks its fastest solver per convolution shape of the frozen 2D encoder, as the reference's own 2D
        # training scripts do (mvpnet/train_2d.py:17, train_mvpnet_3d.py:16: torch.backends.cudnn.benchmark = True);
        # side branches 2.16 -> 1.98 ms, step 4.72 -> 4.61 ms. 0 = the library's default heuristics.
        torch.backends.cudnn.benchmark = True
    import mvkpconv
    syn, ops, stepmod = mvkpconv.sub("synthetic"), mvkpconv.sub("ops"), mvkpconv.sub("step")
    torch.manual_seed(1234)           # same initial weights on every rank
    np.random.seed(1234)
    cfg = syn.make_config(args.workload, deformable=args.deformable, modulated=args.deformable and args.workload == "late")
    net = syn.build_model(cfg, dev)
    net.train()
    if hasattr(net, "net_2d"):
        for m in net.net_2d._modules.values():   # frozen 2D encoder stays in eval mode (architectures_sphere.py:234-237)
            m.train(False)
    params = [p for p in net.parameters() if p.requires_grad]
    deform = [p for n, p in net.named_parameters() if p.requires_grad and "offset" in n]
    other = [p for n, p in net.named_parameters() if p.requires_grad and "offset" not in n]
    groups = [{"params": other}, {"params": deform, "lr": cfg.learning_rate * cfg.deform_lr_factor}]   # trainer.py:72-79
    if os.environ.get("MVK_HIP_SGD", "1") == "1":
        # clip_grad_value_ + SGD (momentum, weight decay) of trainer.py:190-195 as ONE launch over all tensors
        opt = mvkpconv.sub("optim").FusedClipSGD(groups, lr=cfg.learning_rate, momentum=cfg.momentum,
                                                 weight_decay=cfg.weight_decay, clip_value=cfg.grad_clip_norm)
    else:
        opt = torch.optim.SGD(groups, lr=cfg.learning_rate, momentum=cfg.momentum, weight_decay=cfg.weight_decay,
                              fused=os.environ.get("MVK_FUSED_SGD", "1") == "1")
    reducer = stepmod.make_reducer(net, cfg, params, world) if (world > 1 or force_dp) else None

    # ---- synthetic raw inputs, staged in HBM once (data-parallel: different spheres per rank)
    spheres = [syn.raw_sphere(seed=1000 * rank + i, radius=args.in_radius) for i in range(args.spheres)]
    fusion = args.workload != "baseline"
    views = [syn.sphere_views(s, nv=args.views) for s in spheres] if fusion else None
    staged = syn.stage_spheres(spheres, dev, views)
    limits = syn.calibrate_limits(cfg, staged)

    def net_step(batch):
        opt.zero_grad(set_to_none=True)
        return stepmod.net_step_captured(net, batch, cfg, params, opt, reducer, begin=False)

    def eager_step():
        batch, lens = syn.build_batch(cfg, staged, limits, torch.int32)
        return lens, net_step(batch)

    def sync():
        torch.cuda.synchronize()
        if world > 1 or force_dp:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (eager). The gather launches of these steps are timed with HIP events on the
    #      launch stream for the roofline figure (a captured graph cannot carry timing events).
    use_graph = args.graph if args.graph is not None else True
    for _ in range(max(args.warmup, 1)):
        lens, loss = eager_step()
    ops.profile_reset(enabled=True)
    INSTRUMENTED = 16
    for _ in range(INSTRUMENTED):               # instrumented eager passes (same spheres, a fresh grid orientation each)
        lens, loss = eager_step()
    recs, rev_fill = ops._PROF["rec"], ops._PROF.get("h_eff", {})
    contraction = ops.profile_collect_contraction()
    ops.profile_reset(enabled=False)
    ops._PROF["rec"], ops._PROF["h_eff"] = recs, rev_fill      # (h_eff: mean fill of the reverse rows the [dx] gathers walked)

    step = eager_step
    graph_note = ("eager", "eager step (no graph)")
    if use_graph:
        # a capture that fails ends the run with a non-zero exit code: an eager run in its place would report a
        # different execution mode under the same command (--no-graph asks for the eager step explicitly)
        step = stepmod.GraphStep(net, cfg, opt, staged, limits, reducer, dev=args.dev)
        graph_note = step.note
    for _ in range(2 if use_graph else 0):
        lens, loss = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lens, loss = step()
    sync()
    dt = time.perf_counter() - t0
    if hasattr(step, "finish"):
        step.finish()                           # loud check of the input chain's status words
    eager_ms = None
    if rank == 0 and world == 1 and not force_dp and not args.no_eager_line:
        eager_ms = eager_dropin_ms(ops, eager_step, net_step, lambda: syn.build_batch(cfg, staged, limits, torch.int32)[0])
    if args.dev and os.environ.get("MVK_BENCH_DIAG") == "1" and getattr(step, "state", None):
        h = np.asarray(step.state.get("host", [(0, 0)]))[-args.steps:]
        print("DIAG host ms per step: replay enqueue %.2f | build_async %.2f" % tuple(h.mean(0) * 1e3), file=sys.stderr)
    # mean real-neighbour counts of every neighbour matrix of this (fixed) synthetic batch, computed
    # outside the timed region: (Nq, Ns, H) -> H_eff
    hb, _ = syn.build_batch(cfg, staged, limits, torch.int32)
    h_eff = {}
    for l in range(len(hb.points)):
        ns = hb.points[l].shape[0]
        for m in (hb.neighbors[l], hb.pools[l]):
            if m.shape[0] > 0:
                h_eff[(m.shape[0], ns, m.shape[1])] = float((m < ns).sum().item()) / m.shape[0]
    prof = ops.profile_collect(h_eff)
    ops.profile_reset(enabled=False)

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    pts = torch.tensor([float(sum(lens))], device=dev, dtype=torch.float64)
    ranks_seen, per_rank_ms = 1, [dt / args.steps * 1e3]
    if world > 1 or force_dp:
        ranks_seen = dist.get_world_size()      # what the process group (RCCL) actually spans
        every = [torch.zeros_like(t) for _ in range(ranks_seen)]
        dist.all_gather(every, t)
        per_rank_ms = [float(v.item()) / args.steps * 1e3 for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(pts, op=dist.ReduceOp.SUM)
    if ranks_seen != world:
        raise SystemExit("process group spans %d ranks, WORLD_SIZE=%d" % (ranks_seen, world))
    dt = t.item()
    total_points = pts.item() * args.steps

    if rank == 0:
        dp_on = world > 1 or force_dp
        tag, execution_text = graph_note
        res = {
            "metric": "input points/s through MV-KPConv KPFCNN forward+backward (pyramid + fusion + fwd + bwd + SGD)",
            "value": total_points / dt, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s_kpfcnn5_sphere%dk_x%d_per_gpu%s" % (
                args.workload + ("_fusion" if fusion else ""), int(round(sum(lens) / max(args.spheres, 1) / 1000.0)),
                args.spheres, "_deformable" if args.deformable else ""),
                "points_per_step_per_gpu": int(sum(lens)), "views": args.views if fusion else 0,
                "image_hw": [120, 160] if fusion else None, "parallelism": "dp%d" % world, "execution": tag,
                "ranks": ranks_seen, "backend": ("rccl" if backend == "nccl" else backend) if dp_on else None,
                "ms_per_step_per_rank": per_rank_ms,
                "final_loss": float(loss.item()),
                "input_lookahead_batches": int(getattr(step, "lookahead", 0)),
                "capacity_overflow": bool(getattr(step, "state", {}).get("overflow", False))},
            "roofline": roofline(prof),
            "contraction": mfma_report(contraction),
        }
        if not args.no_cpu_baseline:           # timed on rank 0 at N = 1 only (the other ranks would sit in the barrier)
            res["cpu_baseline"] = cpu_baseline(cfg, net, staged, limits, spheres, args) if world == 1 else None
        detail = {"execution": execution_text, "eager_dropin_ms_per_step": eager_ms, "roofline_detail": roofline_detail(prof),
                  "gather_launches": gather_by_level(prof), "cpu_baseline_detail": CPU_DETAIL.get("last")}
        write_detail(res, detail)
        print(result_line(res))
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()


def eager_dropin_ms(ops, eager_step, net_step, make_batch, n=10):
    """What the reference's loop body costs through the drop-in modules WITHOUT the graph executor (outside the timed
    region; bench_detail.json): utils/trainer.py:177-195 -- zero_grad, net(batch), loss, backward, clip_grad_value_,
    step, torch.cuda.synchronize -- (a) with the pyramid built eagerly in front of it (the DataLoader's share,
    datasets/common.py:779-900, host read-backs included), (b) on a batch that is already there. Host-launch bound:
    ~800 launches of ~13 us."""
    ops.set_row_counts(None)
    arena_was, ops._ARENA["on"] = ops._ARENA["on"], False
    try:
        out = {}
        batch = make_batch()
        for key, fn in (("pyramid_and_network", eager_step), ("network_only", lambda: net_step(batch))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
                torch.cuda.synchronize()            # trainer.py:195
            out[key] = (time.perf_counter() - t0) / n * 1e3
        return out
    finally:
        ops._ARENA["on"] = arena_was


def dry_run(args, world, rank):
    """Launch rehearsal without a GPU (CPU test of the --gpus N path): every rank joins a gloo group,
    takes part in the same barrier / MAX / SUM reductions as the real run and rank 0 prints the JSON
    skeleton. No hot-path work, value = 0."""
    dist.init_process_group("gloo")
    got = dist.get_world_size()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(got)]
    dist.all_gather(per_rank, t)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(result_line({"metric": "dry run (launch rehearsal, no GPU work)", "value": 0.0, "unit": "points/s",
                          "n_gpus": got, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t.item(),
                          "config": {"ranks": got, "backend": "gloo", "parallelism": "dp%d" % got,
                                     "ms_per_step_per_rank": [float(v.item()) for v in per_rank]}}))
    dist.barrier()
    dist.destroy_process_group()
    return 0 if got == args.gpus else 3


LINE_LIMIT = 4096          # bytes: the driver's consumer keeps a bounded tail of stdout (BENCH_r02: a 22 KB line was not parsed)


def _short(v, digits=6):
    """Floats to `digits` significant digits, recursively: the line is read by a machine, not diffed bit for bit."""
    if isinstance(v, float):
        return float("%.*g" % (digits, v))
    if isinstance(v, dict):
        return {k: _short(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_short(x, digits) for x in v]
    return v


def result_line(res):
    """The ONE JSON line of the contract, bounded to LINE_LIMIT bytes: everything per-launch lives in the side file
    (write_detail). Raises instead of printing a line its consumer cannot read."""
    line = json.dumps(_short(res), separators=(",", ":"))
    if len(line) >= LINE_LIMIT or "\n" in line:
        raise SystemExit("bench.py: result line is %d bytes (limit %d): move detail to bench_detail.json" % (len(line), LINE_LIMIT))
    return line


def write_detail(res, detail):
    """Per-level gather table, the long execution description and the CPU-baseline protocol go to
    gpurun_out/bench_detail.json (MVK_BENCH_DETAIL overrides the path; failures to write are reported, never fatal)."""
    path = os.environ.get("MVK_BENCH_DETAIL", os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"line": _short(res), "detail": _short(detail)}, f, indent=1)
    except OSError as e:
        print("bench.py: could not write %s: %s" % (path, e), file=sys.stderr)


def gather_by_level(prof):
    """Gather launches of the instrumented steps aggregated per (kernel, H, Cin, K): the row counts of levels >= 1
    move with the random grid orientation, H (the level's neighbour limit) and Cin identify the layer class."""
    agg = {}
    for r in prof.values():
        sh = r["shape"]
        a = agg.setdefault((r["kernel"], sh["H"], sh["Cin"], sh["K"]),
                           {"kernel": r["kernel"], "H": sh["H"], "Cin": sh["Cin"], "K": sh["K"], "launches": 0,
                            "total_ms": 0.0, "bytes": 0.0, "Nq_min": sh["Nq"], "Nq_max": sh["Nq"]})
        a["launches"] += r["launches"]
        a["total_ms"] += r["total_ms"]
        a["bytes"] += r["bytes_per_launch"] * r["launches"]
        a["Nq_min"], a["Nq_max"] = min(a["Nq_min"], sh["Nq"]), max(a["Nq_max"], sh["Nq"])
    out = []
    for a in sorted(agg.values(), key=lambda a: -a["total_ms"]):
        a["avg_us"] = a["total_ms"] / a["launches"] * 1e3
        a["GBps"] = a.pop("bytes") / (a["total_ms"] * 1e-3) / 1e9
        a["frac_of_hbm_peak"] = a["GBps"] / HBM_PEAK_GBS          # every class, not only the one `roofline` reports
        out.append(a)
    return out


def _dominant(prof):
    """The FORWARD gather class with the largest total time (the gather-form feature gradient runs on the same kernel
    over the reverse lists: listed per level in bench_detail.json, tagged [dx], not a candidate for `roofline`)."""
    fwd = [r for r in prof.values() if not r["kernel"].endswith("[dx]")]
    return max(fwd, key=lambda r: r["total_ms"]) if fwd else None


def roofline(prof):
    """Dominant KPConv gather launch class: achieved = algorithmic gathered bytes / average launch
    duration (HIP events), against the HBM peak. Bytes per launch:
        B_g = Nq*H_eff*(Cin*4 + 12 + 4) + Nq*12 + Nq*K*Cin*4."""
    best = _dominant(prof)
    if best is None:
        return None
    traffic, _ = pmc_traffic(best)
    avg_ms = best["total_ms"] / best["launches"]
    achieved = best["bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
    sh = best["shape"]
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "kernel": best["kernel"].split("(")[0],
            "launch": {"Nq": sh["Nq"], "H": sh["H"], "H_eff": sh["H_eff"], "Cin": sh["Cin"], "K": sh["K"]},
            "avg_launch_us": avg_ms * 1e3, "launches": best["launches"],
            "algorithmic_bytes_per_launch": best["bytes_per_launch"]}


def roofline_detail(prof):
    best = _dominant(prof)
    if best is None:
        return None
    traffic, note = pmc_traffic(best)
    avg_ms = best["total_ms"] / best["launches"]
    each = best.get("each_ms") or [avg_ms]
    return {"kernel": best["kernel"], "launch": best["shape"], "traffic_source": note,
            "hbm_frac_measured": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "launch_us_min_median_max": [min(each) * 1e3, float(np.median(each)) * 1e3, max(each) * 1e3],
            "measured_in": "HIP events on the launch stream around every gather launch of the instrumented eager steps "
                           "run between warm-up and the timed region"}


MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD (= f32 vector peak)


def mfma_report(contraction):
    """The dense K x Cin x Cout contraction (forward, gemm_f32_mfma NN / gemm_f32_stream): all launches of the
    instrumented steps together, and the largest one, against the f32 MFMA peak."""
    if not contraction:
        return None
    peak = MFMA_F32_PEAK_TFLOPS
    tot_f = sum(r["flops_per_launch"] * r["launches"] for r in contraction.values())
    tot_t = sum(r["total_ms"] for r in contraction.values()) * 1e-3
    (M, Kd, N), big = max(contraction.items(), key=lambda kv: kv[1]["flops_per_launch"])
    big_tf = big["flops_per_launch"] / (big["total_ms"] / big["launches"] * 1e-3) / 1e12
    return {"bound": "mfma", "dtype": "f32", "achieved": tot_f / tot_t / 1e12, "peak": peak,
            "unit": "TFLOP/s", "frac": tot_f / tot_t / 1e12 / peak,
            "largest": {"M": M, "K": Kd, "N": N, "TFLOP/s": big_tf, "avg_launch_us": big["total_ms"] / big["launches"] * 1e3}}


def pmc_traffic(best):
    """HBM bytes per launch of the dominant gather launch from the newest committed PMC profile
    (profiles/rNN_pmc_gather.json: rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this same command, gfx950
    correction 2*FETCH + WRITE; PMC counters cannot be read from inside the process). Returns
    (bytes or None, note): the value is reported only when the profile was taken with the csrc/kpconv.hip
    of this tree (sha256 recorded in the file) and the launch geometry matches."""
    import glob
    import hashlib
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_gather.json")))
    if not files:
        return None, "no profiles/rNN_pmc_gather.json"
    path = files[-1]
    prof = json.load(open(path))
    src = os.path.join(ROOT, "enhancing-3d-point-cloud-segmentation-using-multi-modal-fusion-with-2d-images_amd", "csrc", "kpconv.hip")
    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()
    if prof.get("kpconv_hip_sha256") != sha:
        return None, "%s was taken with another csrc/kpconv.hip (sha mismatch): re-run the --pmc passes" % os.path.basename(path)
    sh = best["shape"]
    import mvkpconv
    grid_threads = mvkpconv.sub("ops").kpconv_gather_plan(sh["Nq"], sh["Ns"], sh["H"], sh["Cin"])["grid_threads"]
    for l in prof["launches"]:
        if l["kernel"].startswith(("kpconv_gather_vec", "kpconv_gather_mfma")) and l["grid_threads"] == grid_threads:
            return l["traffic_bytes"], "%s (%s)" % (os.path.basename(path), prof.get("command", ""))
    return None, "%s holds no launch with this run's grid" % os.path.basename(path)


CPU_DETAIL = {}


def cpu_baseline(cfg, net, staged, limits, spheres, args):
    """The oracle timed on the host cores (rank 0, N = 1 only): C restatement of the pyramid (single
    thread, like the reference extension) + unfused PyTorch-ops port of the network on all cores.
    Bounded sample: `cpu_baseline_steps` step(s) over ONE sphere of the same workload."""
    import mvkpconv
    from oracle import cport, torch_port
    syn = mvkpconv.sub("synthetic")
    cores = min(os.cpu_count() or 1, 16)      # the 1-GPU box's CPU share; more threads only add OpenMP overhead
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    net2d = None
    if cfg.variant != "baseline":
        net2d = syn.build_model(cfg, torch.device("cpu")).net_2d
        net2d.load_state_dict({k[len("net_2d."):]: v for k, v in sd.items() if k.startswith("net_2d.")})
        for m in net2d._modules.values():
            m.train(False)
    # GPU-built batch gives the fusion inputs; the CPU leg rebuilds the pyramid itself
    one = {k: (v[:1] if isinstance(v, list) else v) for k, v in staged.items()}
    batch, lens = syn.build_batch(cfg, one, limits, torch.int64)
    cb = torch_port.batch_to_cpu(batch)
    sub0 = (staged["points"][0] - staged["center"][0]).cpu().numpy()      # level-0 cloud (scene-load subsampling is not per step)
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.dtype == torch.float32 and
            ("weight" in k or "bias" in k) and not k.startswith("net_2d.") and "running" not in k}

    impl = "ref" if cport.ref() else "oracle"
    from oracle import pyramid_workers

    def pyramid_leg():
        t0 = time.perf_counter()
        pyramid_workers.pyramid(cport, sub0, cfg.first_subsampling_dl, cfg.conv_radius, impl)
        return time.perf_counter() - t0

    def network_leg():
        t0 = time.perf_counter()
        sdl = dict(sd)
        sdl.update(leaf)
        out, reg = torch_port.forward(sdl, cfg, cb, net2d, training=True)
        loss = torch_port.loss_fn(out, cb["labels"], reg, cfg)
        loss.backward()
        for v in leaf.values():
            v.grad = None
        return time.perf_counter() - t0

    def timed(leg, warm, want, budget_s):
        """SURVEY 8d protocol: median of `want` (>= 20) iterations after `warm` (5) warm-ups, cut at the time bound
        that keeps the default bench run within minutes (the cut is reported)."""
        t_end = time.perf_counter() + budget_s
        first = leg()
        for _ in range(warm - 1):
            if time.perf_counter() + first > t_end - want * first * 0.5:
                break
            leg()
        ts = []
        while len(ts) < want and (not ts or time.perf_counter() + first < t_end):
            ts.append(leg())
        return float(np.median(ts)), len(ts)

    want = max(args.cpu_baseline_steps, 1)
    tp, n_p = timed(pyramid_leg, 5, want, 8.0)
    tn, n_n = timed(network_leg, 5, want, 16.0)
    # x P worker processes on the pyramid (the reference's DataLoader workers, train_ScanNet_sphere.py:58,365-377),
    # in a child process that never sees the GPU
    multi = None
    try:
        import subprocess
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".npy", delete=False) as f:
            np.save(f, sub0)
        per = max(2, min(10, int(4.0 / max(tp, 1e-3))))
        r = subprocess.run([sys.executable, "-m", "oracle.pyramid_workers", f.name, str(cores), str(per),
                            repr(float(cfg.first_subsampling_dl)), repr(float(cfg.conv_radius))],
                           cwd=ROOT, capture_output=True, text=True, timeout=120)
        os.unlink(f.name)
        if r.returncode == 0:
            multi = json.loads(r.stdout.strip().splitlines()[-1])
        else:
            print("bench.py: pyramid worker run failed: %s" % r.stderr[-400:], file=sys.stderr)
    except Exception as e:                      # the single-process figures stand on their own
        print("bench.py: pyramid worker run skipped: %r" % (e,), file=sys.stderr)
    n0 = lens[0]
    CPU_DETAIL["last"] = {
        "protocol": "SURVEY 8d: median of >= 20 timed iterations after 5 warm-ups per leg, each leg cut at its time "
                    "bound (pyramid 8 s, network 16 s) so the default run stays within minutes",
        "pyramid": {"impl": "compiled reference core (oracle/_ref)" if cport.ref() else "C oracle", "threads": 1,
                    "median_ms": tp * 1e3, "timed_iterations": n_p},
        "pyramid_worker_processes": multi,
        "network": {"impl": "oracle/torch_port.py (unfused PyTorch-CPU ops), fwd+loss+bwd, no optimizer step",
                    "threads": cores, "median_ms": tn * 1e3, "timed_iterations": n_n},
        "points": n0}
    amort = (1.0 / multi["spheres_per_s"]) if multi else tp
    return {"value": n0 / (tp + tn), "unit": "points/s", "cores": cores, "kind": "port",
            "sample": "one %d-pt sphere: pyramid %s 1 thread median %.0f ms (n=%d) + PyTorch-CPU net fwd+bwd %d threads "
                      "median %.0f ms (n=%d), 5 warm-ups each; pyramid x%d worker procs: %s spheres/s" % (
                          n0, "oracle/_ref" if cport.ref() else "C oracle", tp * 1e3, n_p, cores, tn * 1e3, n_n, cores,
                          ("%.1f" % multi["spheres_per_s"]) if multi else "n/a"),
            "value_pyramid_on_workers": n0 / (amort + tn)}


if __name__ == "__main__":
    main()
