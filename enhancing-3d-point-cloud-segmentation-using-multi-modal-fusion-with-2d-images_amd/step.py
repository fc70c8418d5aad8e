"""The captured training step of the hot path (round 5: moved here from bench.py).

The reference's loop body (KPConv-PyTorch/utils/trainer.py:177-195)

    self.optimizer.zero_grad(); outputs = net(batch, config); loss = net.loss(outputs, batch.labels)
    loss.backward(); torch.nn.utils.clip_grad_value_(net.parameters(), config.grad_clip_norm); self.optimizer.step()
    torch.cuda.synchronize(self.device)

run through the drop-in modules eagerly is ~800 host launches of ~13 us each: host bound (bench_detail.json:
`eager_dropin_ms_per_step`). `GraphStep(net, cfg, optimizer, staged)` is the same step as ONE hipGraph replay with
three parallel branches -- the network (forward, loss, backward, clip, SGD on capacity-padded levels), the sync-free input
chain of the NEXT batch (pyramid, unprojection, 3-NN: synthetic.DeviceInputChain) and the frozen 2D encoder +
FeatureAggregation of the next batch(es) -- see DESIGN.md 4.6, 4.10, 4.11 for every measured choice below.
INTEGRATION.md section 5 shows the three lines that replace the loop body above.

Development knobs (MVK_BENCH_DIAG, MVK_BENCH_SKIP, MVK_BENCH_DUMMY_LAUNCHES) change what a captured step contains; they
are refused unless the step is built with dev=True (bench.py --dev): a judged run cannot pick them up from the
environment by accident.
"""
import os
import sys
import time

import numpy as np
import torch

from . import dp, ops, synthetic as syn

_DEV = {"on": False}
_DEV_KNOBS = ("MVK_BENCH_DIAG", "MVK_BENCH_SKIP", "MVK_BENCH_DUMMY_LAUNCHES")


def _knob(name, default=None):
    """os.environ.get for the development knobs: set without dev mode -> loud refusal, never a silently different step."""
    v = os.environ.get(name)
    if v is None or v == "":
        return default
    if not _DEV["on"]:
        raise RuntimeError("%s=%s is a development knob of the captured step (it changes what a step contains): pass "
                           "dev=True to GraphStep / --dev to bench.py, or unset it" % (name, v))
    return v


class GraphStep:
    """step = GraphStep(net, cfg, optimizer, staged [, limits, reducer]); lens, loss = step(); ...; step.finish()

    net: a KPFCNN of dropin.models (train mode; frozen 2D encoder in eval mode), cfg: its config, optimizer:
    optim.FusedClipSGD (clip + SGD in one launch) or a torch optimizer, staged: the raw inputs resident in HBM
    (synthetic.stage_spheres layout: points, colours, labels, views). Must be created under a non-default current stream
    after at least one eager step on that stream (MIOpen / allocator warm-up), as bench.py does.
    step() replays the graph for batch k (building batch k+1 beside it) and returns (level-0 lengths, loss tensor);
    step.finish() checks the input chain's status words (capacity / list overflow) loudly.
    `lookahead`: how many upcoming batches' views the frozen encoder takes per call (1 or 2): the input
    pipeline must be that many batches ahead -- the synthetic loader is; a loader that is not passes max_lookahead=1."""

    def __init__(self, net, cfg, optimizer, staged, limits=None, reducer=None, dev=False, max_lookahead=2):
        _DEV["on"] = bool(dev)
        params = [p for p in net.parameters() if p.requires_grad]
        if limits is None:
            limits = syn.calibrate_limits(cfg, staged)
        self._state = []
        self._step, self.note = make_graph_step(syn, ops, cfg, net, staged, limits, params, optimizer, reducer,
                                                state_out=self._state, max_lookahead=max(int(max_lookahead), 1))
        self.state = self._state[-1] if self._state else {}
        self.lookahead = int(self.state.get("lookahead", 1))

    def __call__(self):
        return self._step()

    def finish(self):
        if hasattr(self._step, "finish"):
            self._step.finish()


def make_graph_step(syn, ops, cfg, net, staged, limits, params, opt, reducer, state_out=None, max_lookahead=2):
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
            if c <= want and c <= max_lookahead and c * per_batch <= most:
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
        if it == 1 and _knob("MVK_BENCH_DIAG") == "arena":   # development: who asks for zero-filled memory
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
        if _knob("MVK_BENCH_DIAG") == "1":
            print("DIAG zero arena %.1f MB" % (ops.zero_arena_high_water() / 1e6), file=sys.stderr)
    opt.zero_grad(set_to_none=True)
    if _knob("MVK_BENCH_DIAG") == "fills":
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

        def fork_encoder(work=True, parts=("enc", "chain", "fa")):
            # parallel branches of the SAME graph (separate graph launches do not overlap on this runtime,
            # branches of one graph do): features of the NEXT batch's views, and the NEXT batch's pyramid /
            # unprojection / 3-NN, into the other static set.
            # Measured (MVK_BENCH_DIAG=noside / onlyside): network alone 3.33 ms, side branches alone 2.16 ms, together
            # 4.74 ms -- kernel time of the three branches is 6.5 ms, i.e. the wide kernels of the branches time-share
            # the chip. Capturing the network's nodes BEFORE the side branches' (work=False here, side_work() after the
            # network) makes it worse (5.04 ms): the side branches then start late and end after the network.
            cur = torch.cuda.current_stream()
            if enc is not None and "enc" in parts:
                enc_stream.wait_stream(cur)
            if use_chain and "chain" in parts:
                build_stream.wait_stream(cur)
            if work:
                side_work(parts)

        skip = _knob("MVK_BENCH_SKIP", "").split(",")   # development (timing only, the step's inputs go stale):
                                                                   # leave "enc" / "chain" / "fa" out of the side branches

        def side_work(parts=("enc", "chain", "fa")):
            if "enc" not in parts:
                pass
            elif enc is not None and "enc" not in skip and enc_pair:
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
            if use_chain and "chain" not in skip and "chain" in parts:
                with torch.cuda.stream(build_stream):
                    if in_graph_inputs:       # this step's grid orientations: a copy node reading the pinned draw of the host
                        chains[1 - statics.index(static)].upload_rotations()
                    chains[1 - statics.index(static)].build(other)
                    for _ in range(int(_knob("MVK_BENCH_DUMMY_LAUNCHES", "0"))):   # development: what is one more
                        _DUMMY.setdefault(dev, torch.zeros(64, device=dev)).add_(1.0)        # tiny launch on a side branch worth?
            if fa_ahead and "fa" not in skip and "fa" in parts:        # needs both: the encoder's features and the chain's 3-NN pixels of batch k+1
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

        def fork_side_branches():
            # MVK_SIDE_AFTER_BLOCK = k >= 0: the side branches fork from the network's chain AFTER encoder block k instead of at
            # the step's start -- their wide kernels then overlap the latency-bound coarse levels, not the wide level-0 kernels
            # that open the step. Measured on six boxes, one sphere per step (DESIGN 4.12 k): 3.52-3.60 ms against 3.58-3.64
            # (-0.4 .. -2.6 %); baseline net -1.5 %, configs[3] / [4] per rank -0.8 / -1.2 %; with 5-8 spheres per step the side
            # branches are as long as the network and starting them late costs 0.4-0.7 %: default 3 for one or two spheres per
            # step (two: 5.42 against 5.49 ms), off otherwise.
            late = int(os.environ.get("MVK_SIDE_AFTER_BLOCK", "3" if len(staged['points']) <= 2 else "-1"))
            late_enc = int(os.environ.get("MVK_ENC_AFTER_BLOCK", str(late)))        # the frozen encoder separately (-1: at the start)
            if late < 0 and late_enc < 0:
                fork_encoder(work=side_first)
                return []
            blocks = getattr(net, "encoder_blocks", None) or net.encoder_blocks_3d      # (middle fusion: the 3D tower runs first)
            hooks = []
            late, late_enc = min(late, len(blocks) - 1), min(late_enc, len(blocks) - 1)      # (short architectures)

            def at(block, parts):          # fork `parts` at the step's start (block < 0) or after encoder block `block`
                if block < 0:
                    fork_encoder(work=True, parts=parts)
                else:
                    hooks.append(blocks[block].register_forward_hook(lambda *_: fork_encoder(work=True, parts=parts)))
            # FeatureAggregation runs on the encoder's branch and waits for the input chain: it goes with the LATER of the two
            if late_enc == late:
                at(late, ("enc", "chain", "fa"))
            elif late_enc > late:
                at(late, ("chain",))
                at(late_enc, ("enc", "fa"))
            else:
                at(late_enc, ("enc",))
                at(late, ("chain", "fa"))
            return hooks

        if reducer is None or getattr(reducer, "capturable", False):
            diag = _knob("MVK_BENCH_DIAG", "")    # development: "noside" / "onlyside" time the branches apart
            hooks = []
            with torch.cuda.graph(graph, stream=main_stream, capture_error_mode="thread_local"):
                if diag != "noside":
                    if diag != "onlyside":
                        hooks = fork_side_branches()
                    else:
                        fork_encoder(work=side_first)
                if diag != "onlyside":
                    loss = net_step_captured(net, static, cfg, params, opt, reducer)
                else:
                    loss = torch.zeros((), device=dev)
                for h in hooks:
                    h.remove()
                if diag != "noside":
                    if not side_first and not hooks:
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
            hooks = fork_side_branches()
            ops.step_begin()
            loss = net.loss(net(static, cfg), static.labels)
            for h in hooks:
                h.remove()
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
    if ops.capture_table_slots_left(dev) < max(2, enc_cycle):      # (an error inside a capture leaves the stream unusable)
        raise RuntimeError("GraphStep: this process has captured too many steps (the pre-pinned launch tables of "
                           "ops.defer_weight_grads are used up: 32 captured passes per process)")
    replays = [capture(statics[ph % 2], ph) for ph in range(max(2, enc_cycle))]      # (phase of the encoder's cycle; set = phase % 2)
    # "lookahead": the input pipeline must hold the views of this many upcoming batches when a step starts
    state = {"next": None, "k": 0, "free": [None, None], "lookahead": enc_cycle}

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

        status_every = max(2, int(os.environ.get("MVK_STATUS_EVERY", "16")))
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
            # the chains' status words are sticky (atomicMax / atomicOr, never reset): reading them back on every
            # MVK_STATUS_EVERY-th step (and in finish()) reports an overflow as surely as reading them after each replay,
            # without a copy launch between every two graph launches
            state["k"] += 1
            host = (torch.empty(4, dtype=torch.int32).pin_memory()
                    if len(pending) < 4 and state["k"] % status_every in (0, 1) else None)
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

        def finish():
            for c in chains:                     # both chains' words, whatever the step count was
                host = torch.empty(4, dtype=torch.int32).pin_memory()
                host.copy_(c.status4, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pending.append((ev, host))
            check_status(block=True)

        step_chain.finish = finish
        if _knob("MVK_BENCH_DIAG") == "1":
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
        if state_out is not None:
            state_out.append(state)
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
                if _knob("MVK_BENCH_DIAG") == "1":
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

    if _knob("MVK_BENCH_DIAG") == "1":        # development aid: the chains in isolation
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

    if state_out is not None:
        state_out.append(state)
    return step, ("hipGraph[net%s]+eager-input-stream" % ("|enc2d" if enc is not None else ""),
                  "hipGraph(network step: fwd+loss+bwd+clip+SGD over capacity-padded levels, two static sets) | "
                  "second stream: pyramid + unprojection + 3-NN of the next batch" +
                  (" | frozen 2D encoder of the next batch as a parallel branch of the same graph" if enc is not None else ""))


def make_reducer(net, cfg, params, world):
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
