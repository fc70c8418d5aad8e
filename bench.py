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
    ap.add_argument("--features", default="f32", choices=["f32", "f16"],
                    help="f16: BASELINE config 5's fp16-feature mode of every KPConv layer (fp16 features / aggregate / "
                         "weights, fp16 MFMA contraction, f32 accumulation and f32 everything else)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    syn, ops = mvkpconv.sub("synthetic"), mvkpconv.sub("ops")
    ops.set_feature_dtype(torch.float16 if args.features == "f16" else torch.float32)
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
    reducer = make_reducer(mvkpconv.sub("dp"), net, cfg, params, world) if (world > 1 or force_dp) else None

    # ---- synthetic raw inputs, staged in HBM once (data-parallel: different spheres per rank)
    spheres = [syn.raw_sphere(seed=1000 * rank + i, radius=args.in_radius) for i in range(args.spheres)]
    fusion = args.workload != "baseline"
    views = [syn.sphere_views(s, nv=args.views) for s in spheres] if fusion else None
    staged = syn.stage_spheres(spheres, dev, views)
    limits = syn.calibrate_limits(cfg, staged)

    def net_step(batch):
        opt.zero_grad(set_to_none=True)
        return net_step_captured(net, batch, cfg, params, opt, reducer, begin=False)

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
    recs = ops._PROF["rec"]
    contraction = ops.profile_collect_contraction()
    ops.profile_reset(enabled=False)
    ops._PROF["rec"] = recs

    step = eager_step
    graph_note = ("eager", "eager step (no graph)")
    if use_graph:
        # a capture that fails ends the run with a non-zero exit code: an eager run in its place would report a
        # different execution mode under the same command (--no-graph asks for the eager step explicitly)
        step, graph_note = make_graph_step(syn, ops, cfg, net, staged, limits, params, opt, reducer)
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
    if os.environ.get("MVK_BENCH_DIAG") == "1" and state_ref:
        h = np.asarray(state_ref[-1].get("host", [(0, 0)]))[-args.steps:]
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
            "vs_baseline": None, "dtype": "f32" if args.features == "f32" else "f16 features / f32 accumulate",
            "data": "synthetic",
            "config": {"workload": "%s_kpfcnn5_sphere%dk_x%d_per_gpu%s" % (
                args.workload + ("_fusion" if fusion else ""), int(round(sum(lens) / max(args.spheres, 1) / 1000.0)),
                args.spheres, "_deformable" if args.deformable else ""),
                "points_per_step_per_gpu": int(sum(lens)), "views": args.views if fusion else 0,
                "image_hw": [120, 160] if fusion else None, "parallelism": "dp%d" % world, "execution": tag,
                "ranks": ranks_seen, "backend": ("rccl" if backend == "nccl" else backend) if dp_on else None,
                "ms_per_step_per_rank": per_rank_ms,
                "final_loss": float(loss.item()),
                "capacity_overflow": bool(state_ref and state_ref[-1].get("overflow", False))},
            "roofline": roofline(prof),
            "contraction": mfma_report(contraction, args.features),
        }
        if not args.no_cpu_baseline:           # timed on rank 0 at N = 1 only (the other ranks would sit in the barrier)
            res["cpu_baseline"] = cpu_baseline(cfg, net, staged, limits, spheres, args) if world == 1 else None
        detail = {"execution": execution_text, "roofline_detail": roofline_detail(prof),
                  "gather_launches": gather_by_level(prof), "cpu_baseline_detail": CPU_DETAIL.get("last")}
        write_detail(res, detail)
        print(result_line(res))
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()


state_ref = []


def make_graph_step(syn, ops, cfg, net, staged, limits, params, opt, reducer):
    """Capacity-padded static batches + captured hipGraphs for forward, loss, backward, gradient
    all-reduce (N>1), clip and SGD.

    Two static sets / two graph instances: the graph of set k % 2 runs the network on batch k and prepares
    batch k+1 in the other set, so nothing of batch k+1 touches memory the network reads.
    Measured on this runtime (tools/overlap_probe.py): a graph replay does not overlap with work of another
    stream or another graph launch -- only branches inside ONE graph run concurrently. Hence, per graph:
      branch 1  network forward + loss + backward + clip + SGD on static set k % 2;
      branch 2  the frozen eval-mode 2D encoder (architectures_sphere.py:232-237: a pure function of the
                images) on the views of batch k+1 (MVK_ENCODER_AHEAD=0: in line inside the forward);
      branch 3  the input chain of batch k+1 -- pyramid, unprojection, 3-NN -- as a sync-free launch
                sequence with device-side counts (synthetic.DeviceInputChain). MVK_DEVICE_CHAIN=0 builds
                it eagerly on a second stream instead (its shapes then follow the data; only the host round
                trips of the chain are hidden, and a batch that outgrows a capacity falls back to an eager step).
    The host draws the random grid orientations, copies them to the device and launches the graph."""
    has_2d = hasattr(net, "net_2d") and os.environ.get("MVK_ENCODER_AHEAD", "1") == "1"
    dev = staged['points'][0].device
    # Stream rule of the capture (cause of the abort recorded in round 1, gpurun_out/bench7.log: segfault in
    # capture_end): autograd binds every parameter's AccumulateGrad node to the stream of the backward that created
    # it, and the node lives as long as any autograd graph that references it (e.g. a loss tensor of a warm-up step
    # that is still held). torch.cuda.graph() captures on its OWN side stream unless told otherwise, so a backward
    # under capture met nodes of another stream: a cross-stream wait inside the capture -- on the legacy default
    # stream that is not capturable and the runtime aborted. Hence: (1) refuse the default stream, (2) capture on
    # the CURRENT stream, the one every eager warm-up step ran on, so no node ever changes stream.
    main_stream = torch.cuda.current_stream()
    if main_stream == torch.cuda.default_stream():
        raise RuntimeError("make_graph_step must run under a non-default stream (torch.cuda.set_stream(torch.cuda.Stream())): "
                           "autograd nodes created on the legacy default stream cannot take part in a graph capture")
    # all three streams at the same priority: on this driver a priority difference between queues that
    # are busy at the same time costs far more (2-3x the step) than any ordering it buys
    build_stream = torch.cuda.Stream(priority=int(os.environ.get("MVK_BUILD_PRIO", "0")))
    enc_stream = build_stream if os.environ.get("MVK_ONE_SIDE_BRANCH") == "1" else torch.cuda.Stream()
    status = [torch.zeros(2, dtype=torch.int32, device=dev) for _ in range(2)]   # neighbour-search status words

    enc = None
    if has_2d:
        enc_in = torch.stack(staged['images'], 0).clone()          # (b, nv, 3, h, w)

        def encode(images):
            b, nv = images.shape[:2]
            with torch.no_grad():
                return net.net_2d({'image': images.reshape([-1] + list(images.shape[2:]))})['feature']

        for _ in range(2):
            encode(enc_in)
        enc = enc_in

    batch0, lens = syn.build_batch(cfg, staged, limits, torch.int32)
    if enc is not None:
        batch0.feature_2d = encode(enc_in)
    # FeatureAggregation of batch k+1 beside step k as well, where the network detaches its output (early / middle
    # fusion: no trainable state upstream of that point): 12 launches / 0.17 ms off the network's chain
    fa_ahead = enc is not None and getattr(net, "fa_output_detached", False) and hasattr(syn, "DeviceInputChain") \
        and os.environ.get("MVK_DEVICE_CHAIN", "1") == "1" and os.environ.get("MVK_FA_AHEAD", "1") == "1"
    # The frozen encoder for SEVERAL upcoming batches in one call (MVK_ENCODER_PAIR=0: one batch per step): where the
    # network reads the lifted features that were made ahead (fa_ahead) nothing in step k touches
    # statics[k % 2].feature_2d, so with a cycle of 2 the replay of static set 0 encodes the views of batches k+1 AND k+2
    # (into set 1's and its own feature map) and the replay of set 1 encodes nothing; with a cycle of 4 the first step of
    # four encodes batches k+1 .. k+4 (the last two into holding buffers that steps 3 and 4 copy into place). Same work per
    # batch, a half / a quarter of the library launches per step and larger convolutions: stand-alone 1.17 ms for 3 views,
    # 1.73 for 6, 2.71 for 12, 4.65 for 25 (tools/encoder_probe.py). Default: a cycle of 2 (MVK_ENCODER_CYCLE) while the call
    # stays within MVK_ENCODER_PAIR_MAX_VIEWS (12) views: 3.91-3.97 -> 3.80-3.87 ms per step with 6 views per call; 12.8 ->
    # 13.0 ms at 5 spheres x 5 views with 50 (the long call sits beside one step only), which keeps one batch per step.
    enc_cycle = 1
    if fa_ahead and os.environ.get("MVK_ENCODER_PAIR", "1") == "1" and os.environ.get("MVK_DEVICE_CHAIN", "1") == "1":
        per_batch = int(enc_in.shape[0]) * int(enc_in.shape[1])
        most = int(os.environ.get("MVK_ENCODER_PAIR_MAX_VIEWS", "12"))
        want = int(os.environ.get("MVK_ENCODER_CYCLE", "2"))      # (4: measured slower, 4.04 against 3.80-3.85 ms -- twelve views in one
                                                                   # call make their step longer than the three others save)
        for c in (4, 2):
            if c <= want and c * per_batch <= most:
                enc_cycle = c
                break
    enc_pair = enc_cycle > 1
    if enc_pair:
        enc_in2 = torch.cat([enc_in] * enc_cycle, 0).clone()         # (cycle * b, nv, 3, h, w): batches k+1 .. k+cycle
        for _ in range(2):
            encode(enc_in2)
        enc_hold = [torch.empty_like(batch0.feature_2d) for _ in range(enc_cycle - 2)]
    if fa_ahead:
        lift = sys.modules[type(net).__module__].lift_2d_features

        def aggregate(batch):
            """FeatureAggregation on the batch's encoder features (BatchNorm in the module's own mode), no autograd."""
            held, batch.feature_2d3d = getattr(batch, "feature_2d3d", None), None
            try:
                with torch.no_grad():
                    return lift(net, batch)
            finally:
                batch.feature_2d3d = held

        batch0.feature_2d3d = aggregate(batch0)
        # early fusion's forward starts with cat(feature_3d, feature_2d3d) (:290-291): made on the branch as well
        stack_ahead = cfg.variant == "early" and os.environ.get("MVK_STACK_AHEAD", "1") == "1"
        if stack_ahead:
            batch0.stacked_features = torch.cat((batch0.feature_3d, batch0.feature_2d3d), dim=1)
    # Row capacities of levels 1..: the level sizes move by about +-8 % with the random grid orientation, so one
    # batch is not a safe yardstick -- take the largest of a few draws, plus 10 %, rounded up to 64 rows
    # (distinct per level: the masked BatchNorm finds its row-count word by capacity).
    sizes = np.array([[int(p.shape[0]) for p in batch0.points]] +
                     [[int(p.shape[0]) for p in syn.build_batch(cfg, staged, limits, torch.int32)[0].points]
                      for _ in range(int(os.environ.get("MVK_CAPACITY_DRAWS", "12")))])
    caps, used = [], set()
    for l, m in enumerate(sizes.max(0)):
        c = int(m) if l == 0 else int(-(-int(m * 1.10 + 8) // 64) * 64)
        while c in used:
            c += 64
        used.add(c)
        caps.append(c)
    statics = [syn.StaticBatch(batch0, limits, caps=caps), syn.StaticBatch(batch0, limits, caps=caps)]
    # MVK_DEVICE_CHAIN=1 (default): the input side is a sync-free launch sequence with device-side counts,
    # captured as one more parallel branch of the graph (chain s fills static set s)
    use_chain = os.environ.get("MVK_DEVICE_CHAIN", "1") == "1" and hasattr(syn, "DeviceInputChain")
    chains = [syn.DeviceInputChain(cfg, staged, limits, s) for s in statics] if use_chain else None
    if use_chain:
        for c, s in zip(chains, statics):       # warm the workspaces of the chain outside the capture
            c.draw_rotations()
            c.build(s)

    ops.set_row_counts(statics[0].valid)
    ops.zero_arena_high_water(reset=True)
    for it in range(2):                         # momentum buffers, MIOpen / hipBLASLt plans for the padded shapes
        opt.zero_grad(set_to_none=True)
        if it == 1 and os.environ.get("MVK_BENCH_DIAG") == "arena":   # development: who asks for zero-filled memory
            ops._ARENA["log"] = []
        net_step_captured(net, statics[0], cfg, params, opt, reducer)
    torch.cuda.synchronize()
    if ops._ARENA.get("log"):
        import collections
        agg = collections.Counter()
        for nbytes, shape, who in ops._ARENA["log"]:
            agg[(who, shape)] += nbytes
        for (who, shape), nb in agg.most_common(40):
            print("DIAG arena %8.2f MB  %-28s %s" % (nb / 1e6, who, shape), file=sys.stderr)
        ops._ARENA["log"] = None
    ops.step_begin()
    if os.environ.get("MVK_ZERO_ARENA", "1") == "1":
        # one fill per replay instead of ~100 (split-K outputs, scatter targets): sized from the warm-up
        ops.zero_arena_enable(int(ops.zero_arena_high_water() * 1.05) + (1 << 20), dev)
        if os.environ.get("MVK_BENCH_DIAG") == "1":
            print("DIAG zero arena %.1f MB" % (ops.zero_arena_high_water() / 1e6), file=sys.stderr)
    opt.zero_grad(set_to_none=True)
    if os.environ.get("MVK_BENCH_DIAG") == "fills":
        # development: which host call sites still launch zero-fill / copy kernels in a step (arena enabled)
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True,
                     experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
            net_step_captured(net, statics[0], cfg, params, opt, reducer)
        torch.cuda.synchronize()
        import collections
        agg = collections.Counter()
        for ev in prof.events():
            if ev.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::copy_", "aten::clone",
                           "aten::contiguous", "aten::add", "aten::add_", "aten::mul", "aten::cat"):
                st = [f for f in ev.stack if "/torch/" not in f][:4] or list(ev.stack)[:4]
                shape = str(getattr(ev, "input_shapes", ""))[:40]
                agg[(ev.name, shape + " " + " <- ".join(x.split("/")[-1] for x in st))] += 1
        for (name, st), n in agg.most_common(60):
            print("DIAG %3d %-18s %s" % (n, name, st), file=sys.stderr)
        opt.zero_grad(set_to_none=True)

    # the per-step inputs (views, grid orientations) as nodes of the side branches (MVK_INPUTS_IN_GRAPH=0: eager launches
    # on the network's stream between two replays, as up to round 3: ~45 us in front of every step's first kernel)
    in_graph_inputs = use_chain and os.environ.get("MVK_INPUTS_IN_GRAPH", "1") == "1"

    def capture(static, phase=0):
        ops.set_row_counts(static.valid)
        opt.zero_grad(set_to_none=True)         # every graph instance produces its own .grad tensors
        graph = torch.cuda.CUDAGraph()
        other = statics[1 - statics.index(static)]

        side_first = os.environ.get("MVK_NET_FIRST") != "1"      # development: 1 = capture the network's nodes first

        def fork_encoder(work=True):
            # parallel branches of the SAME graph (separate graph launches do not overlap on this runtime,
            # branches of one graph do): features of the NEXT batch's views, and the NEXT batch's pyramid /
            # unprojection / 3-NN, into the other static set.
            # Measured (MVK_BENCH_DIAG=noside / onlyside): network alone 3.33 ms, side branches alone 2.16 ms, together
            # 4.74 ms -- kernel time of the three branches is 6.5 ms, i.e. the wide kernels of the branches time-share
            # the chip. Capturing the network's nodes BEFORE the side branches' (work=False here, side_work() after the
            # network) makes it worse (5.04 ms): the side branches then start late and end after the network.
            cur = torch.cuda.current_stream()
            if enc is not None:
                enc_stream.wait_stream(cur)
            if use_chain:
                build_stream.wait_stream(cur)
            if work:
                side_work()

        skip = os.environ.get("MVK_BENCH_SKIP", "").split(",")   # development (timing only, the step's inputs go stale):
                                                                   # leave "enc" / "chain" / "fa" out of the side branches

        def side_work():
            if enc is not None and "enc" not in skip and enc_pair:
                with torch.cuda.stream(enc_stream):
                    if phase == 0:      # this replay: the views of batches k+1 (other set), k+2 (this set again), k+3, k+4 (held)
                        if in_graph_inputs:
                            views = torch.stack(staged['images'], 0)
                            enc_in2.copy_(torch.cat([views] * enc_cycle, 0))
                        every = encode(enc_in2)
                        one = every.shape[0] // enc_cycle
                        other.feature_2d.copy_(every[:one])
                        static.feature_2d.copy_(every[one:2 * one])
                        for h, hold in enumerate(enc_hold):
                            hold.copy_(every[(2 + h) * one:(3 + h) * one])
                    elif phase >= 2:    # the features of batch k+1 were made two or three steps ago
                        other.feature_2d.copy_(enc_hold[phase - 2])
            elif enc is not None and "enc" not in skip:
                with torch.cuda.stream(enc_stream):
                    if in_graph_inputs:       # the views of batch k+1 enter on this branch, not by eager launches on the network's queue
                        enc_in.copy_(torch.stack(staged['images'], 0))
                    other.feature_2d.copy_(encode(enc_in))
            if use_chain and "chain" not in skip:
                with torch.cuda.stream(build_stream):
                    if in_graph_inputs:       # this step's grid orientations: a copy node reading the pinned draw of the host
                        chains[1 - statics.index(static)].upload_rotations()
                    chains[1 - statics.index(static)].build(other)
                    for _ in range(int(os.environ.get("MVK_BENCH_DUMMY_LAUNCHES", "0"))):   # development: what is one more
                        _DUMMY.setdefault(dev, torch.zeros(64, device=dev)).add_(1.0)        # tiny launch on a side branch worth?
            if fa_ahead and "fa" not in skip:        # needs both: the encoder's features and the chain's 3-NN pixels of batch k+1
                enc_stream.wait_stream(build_stream)
                with torch.cuda.stream(enc_stream):
                    if stack_ahead:     # the network's input features in one go: [feature_3d | lifted features]
                        torch.cat((other.feature_3d, aggregate(other)), dim=1, out=other.stacked_features)
                    else:
                        other.feature_2d3d.copy_(aggregate(other))

        def join_encoder():
            if enc is not None:
                torch.cuda.current_stream().wait_stream(enc_stream)
            if use_chain:
                torch.cuda.current_stream().wait_stream(build_stream)

        if reducer is None or getattr(reducer, "capturable", False):
            diag = os.environ.get("MVK_BENCH_DIAG", "")    # development: "noside" / "onlyside" time the branches apart
            with torch.cuda.graph(graph, stream=main_stream, capture_error_mode="thread_local"):
                if diag != "noside":
                    fork_encoder(work=side_first)
                if diag != "onlyside":
                    loss = net_step_captured(net, static, cfg, params, opt, reducer)
                else:
                    loss = torch.zeros((), device=dev)
                if diag != "noside":
                    if not side_first:
                        side_work()
                    join_encoder()
            return graph.replay, loss
        # N > 1: the RCCL all-reduces stay eager calls between graphs (capturing them was tried with a one-rank process
        # group: the group's watchdog thread queries an event recorded in the capturing stream and aborts with
        # hipErrorCapturedEvent). With the staged exchange (dp.two_stage_backward) the step is three graphs:
        #   G1 forward + loss + backward above the cut + pack bucket 0   | eager: start all-reduce 0 (asynchronous)
        #   G2 backward below the cut + pack bucket 1                   | eager: start all-reduce 1, wait for both
        #   G3 unpack + clip + SGD
        # so the ring of bucket 0 (~95 % of the bytes) runs on RCCL's stream while G2 computes.
        # thread_local capture mode: the process group's watchdog thread may query events meanwhile.
        staged_exchange = hasattr(reducer, "cut_block")
        net.backward_cut = reducer.cut_block if staged_exchange else None
        scope = backward_scope(ops)
        with torch.cuda.graph(graph, stream=main_stream, capture_error_mode="thread_local"):
            fork_encoder()
            ops.step_begin()
            loss = net.loss(net(static, cfg), static.labels)
            if staged_exchange:
                orig, leaves = net.cut_tensors
                with scope():
                    loss.backward(reducer.seed(loss), retain_graph=reducer.dp.deformable_below(cfg.architecture, reducer.cut_block))
                reducer.pack(0)
            else:
                backward(ops, loss)
            join_encoder()
        if not staged_exchange:
            graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_b, stream=main_stream, pool=graph.pool(), capture_error_mode="thread_local"):
                clip_and_step(params, opt, cfg)
            grads = [p.grad for p in reducer.params if p.grad is not None]

            def replay():
                graph.replay()
                reducer(grads)
                graph_b.replay()
            return replay, loss
        graph_2, graph_3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph_2, stream=main_stream, pool=graph.pool(), capture_error_mode="thread_local"):
            pairs = [(t, l.grad) for t, l in zip(orig, leaves) if t.requires_grad and l.grad is not None]
            with scope():
                torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])
            reducer.pack(1)
        with torch.cuda.graph(graph_3, stream=main_stream, pool=graph.pool(), capture_error_mode="thread_local"):
            reducer.unpack(0)
            reducer.unpack(1)
            clip_and_step(params, opt, cfg)

        def replay():
            graph.replay()
            reducer.launch(0)
            graph_2.replay()
            reducer.launch(1)
            reducer.wait()
            graph_3.replay()
        return replay, loss

    _SEED.setdefault(dev, torch.ones((), device=dev))      # (before the captures: see backward())
    _DUMMY.setdefault(dev, torch.zeros(64, device=dev))
    replays = [capture(statics[ph % 2], ph) for ph in range(max(2, enc_cycle))]      # (phase of the encoder's cycle; set = phase % 2)
    state = {"next": None, "k": 0, "free": [None, None]}

    if use_chain:
        lens0 = [int(p.shape[0]) for p in staged['points']]
        pending = []                            # (event, pinned copy of a chain's status word)

        def check_status(block=False):
            while pending and (block or pending[0][0].query()):
                ev, host = pending.pop(0)
                ev.synchronize()
                if (int(host[1]) or int(host[3])) and not state.get("overflow"):
                    # never silent: the step ran on a level cut at its capacity (or a neighbour row cut at the
                    # list size); reported on stderr and in the JSON line, the run goes on
                    state["overflow"] = True
                    print("WARNING input chain: a level outgrew its captured capacity, a query its neighbour list or a "
                          "support its reverse-list width (status %s, capacities %s)" % (host.tolist(), statics[0].caps),
                          file=sys.stderr)

        chains[0].draw_rotations()
        chains[0].build(statics[0])             # batch 0; every later batch is built by the graph before it
        state["slot"] = 0

        replayed = [None, None]                 # per static set: event after the last replay that built into it

        def step_chain():
            slot = state["slot"]
            check_status()
            if replayed[slot ^ 1] is not None:  # bound the host's run-ahead to two steps: the pinned staging
                replayed[slot ^ 1].synchronize()   # buffer of this chain must not be rewritten before its copy ran
            # host draw of batch k+1's grid orientations into pinned memory (the replay's copy node reads it: the previous
            # replay of this chain is two steps back and has long read its own) + the views of batch k+1
            chains[slot ^ 1].draw_rotations(upload=not in_graph_inputs)
            phase = state.get("phase", 0)
            if enc is not None and not in_graph_inputs:
                if enc_pair:
                    if phase == 0:
                        views = torch.stack(staged['images'], 0)
                        enc_in2.copy_(torch.cat([views] * enc_cycle, 0))  # the views of batches k+1 .. k+cycle
                else:
                    enc_in.copy_(torch.stack(staged['images'], 0))      # the views of batch k+1
            ta = time.perf_counter()
            replays[phase][0]()
            host = torch.empty(4, dtype=torch.int32).pin_memory() if len(pending) < 4 else None
            if host is not None:
                host.copy_(chains[slot ^ 1].status4, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pending.append((ev, host))
            replayed[slot ^ 1] = torch.cuda.Event()
            replayed[slot ^ 1].record()
            state.setdefault("host", []).append((time.perf_counter() - ta, 0.0))
            state["slot"] = slot ^ 1
            state["phase"] = (phase + 1) % len(replays)
            return lens0, replays[phase][1]

        step_chain.finish = lambda: check_status(block=True)
        if os.environ.get("MVK_BENCH_DIAG") == "1":
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                replays[0][0]()
                replays[1][0]()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(10):
                chains[0].build(statics[0])
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print("DIAG graph replay (network + encoder + input chain branches) %.2f ms | input chain alone, eager "
                  "%.2f ms" % ((t1 - t0) * 50, (t2 - t1) * 100), file=sys.stderr)
        state_ref.append(state)
        tag = "hipGraph[net|chain%s]%s" % ((("|enc2d(x%d every %s step)+fa" % (enc_cycle, "2nd" if enc_cycle == 2 else "4th") if enc_pair else "|enc2d+fa") if fa_ahead else "|enc2d")
                                           if enc is not None else "",
                                           "" if reducer is None else
                                           "+rccl-in-graph" if getattr(reducer, "capturable", False) else "+eager-rccl(3 graphs)")
        return step_chain, (tag, "hipGraph with %s branches per step: network fwd+loss+bwd+clip+SGD on static set k%%2 | "
                            "sync-free input chain (pyramid, unprojection, 3-NN; device-side counts) of batch k+1"
                            % ("three" if enc is not None else "two")
                            + (((" | frozen 2D encoder: the views of batches k+1 .. k+%d in one call on every %s step "
                                 "(none on the steps between)" % (enc_cycle, "second" if enc_cycle == 2 else "fourth")
                                 if enc_pair else " | frozen 2D encoder of batch k+1")
                                + (", then its FeatureAggregation (the network "
                                                                       "detaches that output: nothing trainable is "
                                                                       "upstream of it)" if fa_ahead else ""))
                               if enc is not None else
                               (" (2D encoder in line with the network)" if hasattr(net, "net_2d") else ""))
                            + ("" if reducer is None else
                               " | gradient all-reduce (RCCL, two buckets) captured as a branch of the same graph"
                               if getattr(reducer, "capturable", False) else
                               " | gradient all-reduce: eager RCCL calls between the graphs of the step (backward "
                               "above the cut | below the cut | unpack+clip+SGD), bucket 0 overlapped with the second"))

    def build_async(slot):
        """Enqueues batch k+1 on the build / encode streams and pads it into static set `slot`."""
        free = state["free"][slot]
        if free is not None:                       # (the searches of the batch this set held are long done)
            with torch.cuda.stream(build_stream):
                ops.check_neighbor_status(status[slot])
                if state.get("rev_status", [None, None])[slot] is not None:
                    ops.check_reverse_status(state["rev_status"][slot])
        if free is not None:                       # the replay that last read this set has finished
            build_stream.wait_event(free)
        with torch.cuda.stream(build_stream):
            status[slot].zero_()
            batch, lens = syn.build_batch(cfg, staged, limits, torch.int32, status=status[slot])
            state.setdefault("rev_status", [None, None])[slot] = getattr(batch, "rev_status", None)
            fits = True
            try:
                statics[slot].load(batch)
            except RuntimeError as e:           # a level outgrew its captured capacity
                fits = False
                state["fallbacks"] = state.get("fallbacks", 0) + 1
                if os.environ.get("MVK_BENCH_DIAG") == "1":
                    print("DIAG eager fallback:", str(e)[:200], file=sys.stderr)
            ev = torch.cuda.Event()
            ev.record(build_stream)
        return batch, lens, ev, fits, slot

    state["next"] = build_async(0)

    def step():
        batch, lens, ev, fits, slot = state["next"]
        main = torch.cuda.current_stream()
        main.wait_event(ev)
        if not fits:                            # run this step eagerly on the exact-size batch
            ops.set_row_counts(None)
            ops._ARENA["on"], arena_was = False, ops._ARENA["on"]
            opt.zero_grad(set_to_none=True)     # fresh .grad tensors: the captured ones are slices of the zero arena
            if enc is not None:                 # keep the encoder pipeline going: features for the next set
                statics[slot ^ 1].feature_2d.copy_(encode(enc_in))
            loss = net_step_captured(net, batch, cfg, params, opt, reducer)   # plain eager step on the exact-size batch
            ops._ARENA["on"] = arena_was
            main.synchronize()                  # rare path: `batch` lives in the build stream's pool
            state["next"] = build_async(slot)
            return lens, loss
        ops.set_row_counts(statics[slot].valid)
        if enc is not None:
            enc_in.copy_(torch.stack(staged['images'], 0))      # the views of batch k+1
        ta = time.perf_counter()
        replays[slot][0]()
        done = torch.cuda.Event()
        done.record(main)
        state["free"][slot] = done
        state["keep"] = batch                   # alive until its padding copies are ordered before `ev`
        tb = time.perf_counter()
        state["next"] = build_async(slot ^ 1)
        tc = time.perf_counter()
        state.setdefault("host", []).append((tb - ta, tc - tb))
        return lens, replays[slot][1]

    if os.environ.get("MVK_BENCH_DIAG") == "1":        # development aid: the chains in isolation
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            replays[0][0]()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            bb, _ = syn.build_batch(cfg, staged, limits, torch.int32)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(10):
            statics[1].load(batch0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(10 if enc is not None else 0):
            encode(enc_in)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        for _ in range(10):
            replays[1][0]()
        torch.cuda.synchronize()
        t5 = time.perf_counter()
        for _ in range(5):
            replays[0][0]()
            replays[1][0]()
        torch.cuda.synchronize()
        t6 = time.perf_counter()
        print("DIAG graph replay %.2f ms (set 1: %.2f, alternating: %.2f) | build_batch %.2f ms | static.load %.2f ms | "
              "2D encoder (eager) %.2f ms" % ((t1 - t0) * 100, (t5 - t4) * 100, (t6 - t5) * 100, (t2 - t1) * 100,
                                            (t3 - t2) * 100, (t4 - t3) * 100), file=sys.stderr)

    state_ref.append(state)
    return step, ("hipGraph[net%s]+eager-input-stream" % ("|enc2d" if enc is not None else ""),
                  "hipGraph(network step: fwd+loss+bwd+clip+SGD over capacity-padded levels, two static sets) | "
                  "second stream: pyramid + unprojection + 3-NN of the next batch" +
                  (" | frozen 2D encoder of the next batch as a parallel branch of the same graph" if enc is not None else ""))


def make_reducer(dp, net, cfg, params, world):
    """Gradient exchange of the N > 1 path: two buckets around a cut of the backward at the entry of encoder level 2
    (dp.py: bucket 0 = head + decoder + levels >= 2 = ~95 % of the bytes, reduced while the backward of levels 0-1
    still runs). MVK_DP_OVERLAP=0: one flat bucket after the whole backward."""
    cut = dp.cut_block_of_layer(cfg.architecture, 2) if os.environ.get("MVK_DP_OVERLAP", "1") == "1" else None
    if cut is None or not hasattr(net, "encoder_blocks"):
        return dp.FlatAllReduce(params, world)
    late, early = dp.split_parameters_at(net, cut)
    # MVK_DP_GRAPH_COLLECTIVES=1 (opt-in: verified with a one-rank communicator only, this pool gives one GPU per
    # box): the all-reduces go straight to librccl and are captured as a branch of the step's single graph
    comm = None
    if os.environ.get("MVK_DP_GRAPH_COLLECTIVES", "0") == "1" and torch.distributed.get_backend() == "nccl":
        comm = dp.RcclCommunicator(params[0].device)
    red = dp.BucketedAllReduce([late, early], world, comm=comm, prescaled=os.environ.get("MVK_DP_PRESCALE", "1") == "1")
    red.cut_block = cut
    red.dp = dp
    return red


def net_step_captured(net, static, cfg, params, opt, reducer, begin=True):
    """One network step (no host sync inside): forward, loss, backward, gradient exchange (N > 1), clip, SGD --
    the body of the captured graph for N = 1, and the eager step of every configuration."""
    import mvkpconv
    ops = mvkpconv.sub("ops")
    if begin:
        ops.step_begin()
    staged_exchange = reducer is not None and hasattr(reducer, "cut_block")
    tail = split_tail_plan(net, cfg, opt) if reducer is None else None
    net.backward_cut = reducer.cut_block if staged_exchange else (tail["cut"] if tail else None)
    out = net(static, cfg)
    loss = net.loss(out, static.labels)
    if staged_exchange:
        def between():
            reducer.pack(0)
            reducer.launch(0)
        reducer.dp.two_stage_backward(loss, net.cut_tensors, between=between, backward_scope=backward_scope(ops),
                                      seed=reducer.seed(loss),
                                      retain_graph=reducer.dp.deformable_below(cfg.architecture, reducer.cut_block))
        reducer.pack(1)
        reducer.launch(1)
        reducer.wait()
        reducer.unpack(0)
        reducer.unpack(1)
    elif tail:
        split_tail_backward(ops, net, loss, opt, tail)
        return loss
    else:
        backward(ops, loss)
        if reducer is not None:
            reducer()
    clip_and_step(params, opt, cfg)
    return loss


_TAIL = {}


def split_tail_plan(net, cfg, opt):
    """N = 1: the tail of a step -- the grouped weight-gradient launch (0.23 ms) and clip + SGD (0.075 ms) -- sits
    behind the whole backward although 95 % of its bytes belong to the head, the decoder and encoder levels >= 2, whose
    gradients are complete when the backward reaches the cut at the entry of level 2 (dp.py: the same cut the N > 1
    exchange uses). With the cut, that part of the tail runs on a side stream (a branch of the captured graph) beside
    the backward of levels 0-1. MVK_SPLIT_TAIL=0: everything at the end, on the chain."""
    if os.environ.get("MVK_SPLIT_TAIL", "0") != "1" or not hasattr(net, "encoder_blocks") or not hasattr(opt, "clip"):
        return None
    if os.environ.get("MVK_DEFER_DW", "1") != "1" or os.environ.get("MVK_OVERLAP_DW", "0") == "1":
        return None
    key = id(net)
    if key not in _TAIL:
        import mvkpconv
        dp = mvkpconv.sub("dp")
        cut = dp.cut_block_of_layer(cfg.architecture, int(os.environ.get("MVK_TAIL_CUT_LAYER", "2")))
        if cut is None:
            _TAIL[key] = None
        else:
            late, early = dp.split_parameters_at(net, cut)
            _TAIL[key] = {"cut": cut, "late": late, "early": early, "stream": torch.cuda.Stream()}
    return _TAIL[key]


def split_tail_backward(ops, net, loss, opt, tail):
    """Backward in two stages around net.backward_cut; the deferred weight gradients and the optimiser step of the
    parameters above the cut on the side stream while stage 2 runs; those below the cut at the end."""
    main, side = torch.cuda.current_stream(), tail["stream"]
    if os.environ.get("MVK_TAIL_SIDE", "1") != "1":       # development: the same two pieces, both on the chain
        side = main
    orig, leaves = net.cut_tensors
    with ops.defer_weight_grads(flush=False) as scope:
        loss.backward()
    if side is not main:
        side.wait_stream(main)
    ops.flush_deferred(scope.take(), side)
    with torch.cuda.stream(side):
        opt.step(only=tail["late"])
    pairs = [(t, l.grad) for t, l in zip(orig, leaves) if t.requires_grad and l.grad is not None]
    if pairs:
        with ops.defer_weight_grads():
            torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])
    opt.step(only=tail["early"])
    if side is not main:
        main.wait_stream(side)


def backward_scope(ops):
    """Context manager factory for a backward pass: all weight-gradient products of the pass as one grouped launch at
    its end (ops.defer_weight_grads, default), on a side branch (MVK_OVERLAP_DW=1), or in line (MVK_DEFER_DW=0)."""
    import contextlib
    if os.environ.get("MVK_OVERLAP_DW", "0") == "1":
        return ops.overlap_weight_grads
    return ops.defer_weight_grads if os.environ.get("MVK_DEFER_DW", "1") == "1" else contextlib.nullcontext


_SEED = {}
_DUMMY = {}


def backward(ops, loss):
    """loss.backward() with the weight-gradient products on a side branch (ops.overlap_weight_grads): nothing reads
    a gradient before the optimiser (or the all-reduce), which run after the scope has joined."""
    seed = _SEED.get(loss.device)
    if seed is None:        # the seed of loss.backward() -- a tensor of ones, one fill launch per call -- made once
        seed = _SEED[loss.device] = torch.ones((), device=loss.device, dtype=loss.dtype)
    with backward_scope(ops)():
        loss.backward(seed)


def clip_and_step(params, opt, cfg):
    """utils/trainer.py:190-195: clip_grad_value_(grad_clip_norm) then optimizer.step()."""
    if hasattr(opt, "clip"):            # FusedClipSGD clips inside its single launch
        opt.step()
    else:
        torch.nn.utils.clip_grad_value_(params, cfg.grad_clip_norm)
        opt.step()


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


MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA peak of the MI355X (the vendor's headline figure includes 2:1 sparsity)


def mfma_report(contraction, features="f32"):
    """The dense K x Cin x Cout contraction (forward, gemm_f32_mfma / gemm_f16_mfma NN): all launches of the
    instrumented steps together, and the largest one, against the MFMA peak of the operand type."""
    if not contraction:
        return None
    peak = MFMA_F32_PEAK_TFLOPS if features == "f32" else MFMA_F16_PEAK_TFLOPS
    tot_f = sum(r["flops_per_launch"] * r["launches"] for r in contraction.values())
    tot_t = sum(r["total_ms"] for r in contraction.values()) * 1e-3
    (M, Kd, N), big = max(contraction.items(), key=lambda kv: kv[1]["flops_per_launch"])
    big_tf = big["flops_per_launch"] / (big["total_ms"] / big["launches"] * 1e-3) / 1e12
    return {"bound": "mfma", "dtype": features, "achieved": tot_f / tot_t / 1e12, "peak": peak,
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
        if l["kernel"].startswith("kpconv_gather_vec") and l["grid_threads"] == grid_threads:
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
