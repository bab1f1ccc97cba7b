#!/usr/bin/env python3
"""Headline benchmark: batched 7-DoF RNEA + gradient evaluations per second (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path over one batch: a single ``rbd_rnea_grad_f32`` launch that
maps ``B`` rows ``(q, qd, qdd)`` (already resident in HBM) to ``(c, dc_du)`` -- the work of ``B``
calls of the reference's ``rnea_grad`` (``/root/reference/RBDReference.py:1345-1368``).  Workload:
BASELINE.json configs[3] ("7-DoF iiwa rnea_grad batch=1M sharded across 8xMI355X"): a GLOBAL batch of
``--batch`` (1 048 576) rows, rank r evaluating rows ``shard_bounds(B, world, r)`` -- strong scaling,
the configuration as written, no data-path collective (rows are independent).  With
``--scaling weak`` every rank evaluates its own ``--batch`` rows instead.  Under torchrun the other mode
is measured in the same run and reported as a second object (``"weak"`` / ``"strong"``); at N = 1 the
two coincide.  Every step uses the next of ``--rotate`` (default 4) input/output buffer sets, so
that more input bytes are in flight (4 x 88 MB) than the 256 MB Infinity Cache holds.  configs[1]
(B = 4096) is reported beside it under "extra" -- at that size a launch is latency-bound and says
nothing about the roofline.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
``roofline`` (algorithmic HBM bytes per launch / measured kernel time vs the 8 TB/s peak; the
kernel is FP32-VALU-bound, so the VALU fraction is reported too) and ``cpu_baseline`` (the CPU
oracle timed on this host's cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
FP32_VALU_PEAK_TFLOPS = 157.3  # spec vector peak
N_DOF = 7
BYTES_PER_EVAL = (4 * N_DOF + 2 * N_DOF * N_DOF) * 4   # in 3n + out (n + 2n^2), fp32 = 504 B


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1 << 20,
                    help="rows per step: the global batch (strong scaling) / rows per GPU (weak scaling)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--rotate", type=int, default=4, help="input/output buffer sets cycled through by the steps")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="time the K steps as ONE replay of a HIP graph of K launches (auto: strong scaling on more than one GPU, "
                         "where a rank's shard is small enough for the launch path to show)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    return ap.parse_args()


def make_inputs(B, n, seed, device, dtype=torch.float32):
    """Synthetic inputs of SURVEY.md §8d: q ~ U(-pi, pi), qd, qdd ~ U(-1, 1); numpy PCG64 on the
    host, uploaded before the timed region (device "cpu": left on the host)."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(-np.pi, np.pi, (B, n)).astype(np.float32 if dtype == torch.float32 else np.float64)
    qd = rng.uniform(-1, 1, (B, n)).astype(q.dtype)
    qdd = rng.uniform(-1, 1, (B, n)).astype(q.dtype)
    return tuple(torch.from_numpy(x).to(device) for x in (q, qd, qdd))


class GradStep:
    """Pre-allocated buffers + a direct C-ABI launch on torch's current stream.  `sets` input/output
    buffer sets are used round-robin (each call takes the next one)."""

    def __init__(self, rbd, inputs):
        self.rbd = rbd
        self.inputs = inputs                      # list of (q, qd, qdd)
        q = inputs[0][0]
        B, n = q.shape
        self.B = B
        self.outs = [(torch.empty((B, n), device=q.device, dtype=q.dtype),
                      torch.empty((B, n, 2 * n), device=q.device, dtype=q.dtype)) for _ in inputs]
        self.fn = rbd._fn("rbd_rnea_grad", q.dtype)
        self.stream = torch.cuda.current_stream(q.device).cuda_stream
        self.args = [(a.data_ptr(), b.data_ptr(), c_.data_ptr(), -9.81, 0, B, o[0].data_ptr(), o[1].data_ptr(), self.stream)
                     for (a, b, c_), o in zip(inputs, self.outs)]
        self.k = 0

    @property
    def dc(self):
        return self.outs[0][1]

    def __call__(self):
        a = self.args[self.k]
        self.k = (self.k + 1) % len(self.args)
        rc = self.fn(*a)
        if rc != 0:
            self.rbd._lib.check(rc)

    def on_stream(self, st):
        """The next launch on stream `st` (a raw hipStream_t) instead of the stream this object was built on."""
        a = self.args[self.k]
        self.k = (self.k + 1) % len(self.args)
        rc = self.fn(*a[:-1], st)
        if rc != 0:
            self.rbd._lib.check(rc)


class GraphSteps:
    """K launches of a `GradStep` captured ONCE into a HIP graph; calling the object replays all K.  For shards so small that
    the gap between eager launches is a fifth of the step (131 072 rows per rank at N = 8, 20.6 us per
    eager launch around an 18 us kernel) the graph takes the launch path out of the step; the work is the same K launches
    on the same rotating buffer sets."""

    def __init__(self, step, K, dev):
        self.K = K
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            step.on_stream(side.cuda_stream)            # everything the launch path sets up lazily happens outside the capture
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            st = torch.cuda.current_stream(dev).cuda_stream
            for _ in range(K):
                step.on_stream(st)

    def __call__(self):
        self.graph.replay()


def time_kernel_ms(fn, steps, warmup):
    """Average duration of one launch from HIP events on the launch stream."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def time_extra_ms(fn, steps, warmup, reps=3, ramp_s=0.05):
    """Extras only: best of `reps` timed groups (one stalled launch on a shared box otherwise
    dominates a 20-launch average).  Like the headline (main(): "clock ramp"), every extra is timed at the GPU's
    working clock: its set-up (allocations, a library load) leaves the GPU idle for milliseconds, and a group of a hundred
    20 us launches is over before the clock is back -- the 30-body minv measured 22.9 us this way against 20.4 us in
    a loop that runs for seconds (tools/exp_minv_abi.py).  `ramp_s` seconds of the same launches, untimed, come first."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < ramp_s:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    return min(time_kernel_ms(fn, steps, warmup if r == 0 else 1) for r in range(reps))


def cpu_baseline(robot, seed, budget_s=15.0):
    """CPU baseline on a bounded sample of the same workload: the C restatement of the oracle
    (oracle/rbd_oracle.c: the reference's passes, dense 6x6 arithmetic, float64, one configuration
    at a time per thread, OpenMP over rows on every host core); the numpy oracle's rates are
    reported beside it."""
    from oracle import rbd_oracle as orc
    om = orc.model_from_robot(robot)
    rng = np.random.default_rng(seed)
    extra = {}
    try:
        from oracle.c_oracle import COracle
        co = COracle(robot)
        threads = co.max_threads
        chunk = 1 << 16
        q = rng.uniform(-np.pi, np.pi, (chunk, om.n)); qd = rng.uniform(-1, 1, (chunk, om.n)); qdd = rng.uniform(-1, 1, (chunk, om.n))
        outb = (np.zeros((chunk, om.n)), np.zeros((chunk, om.n, 2 * om.n)))
        co.rnea_grad(q, qd, qdd, out=outb)                            # warm-up (thread pool, pages)
        done = 0
        t0 = time.perf_counter()
        while True:
            co.rnea_grad(q, qd, qdd, out=outb)
            done += chunk
            el = time.perf_counter() - t0
            if el > budget_s or done >= (1 << 24):
                break
        t1 = time.perf_counter(); co.rnea_grad(q[:8192], qd[:8192], qdd[:8192], threads=1); one = 8192 / (time.perf_counter() - t1)
        main = {"value": done / el, "unit": "evals/s", "cores": threads, "kind": "port",
                "sample": f"{done} rows of the same workload (chunks of {chunk}) through oracle/rbd_oracle.c "
                          f"(float64, dense 6x6 restatement of the reference's passes, OpenMP x{threads}); "
                          f"1 thread: {one:.0f} evals/s"}
    except Exception as e:           # no gcc on the box: fall back to the numpy oracle as the baseline
        main = None
        extra["c_oracle_error"] = repr(e)
    chunk = 4096
    q = rng.uniform(-np.pi, np.pi, (chunk, om.n)); qd = rng.uniform(-1, 1, (chunk, om.n)); qdd = rng.uniform(-1, 1, (chunk, om.n))
    orc.rnea_grad(om, q[:64], qd[:64], qdd[:64], return_c=True)
    t0 = time.perf_counter(); done = 0
    while time.perf_counter() - t0 < (3.0 if main else budget_s):
        orc.rnea_grad(om, q, qd, qdd, return_c=True); done += chunk
    np_rate = done / (time.perf_counter() - t0)
    t1 = time.perf_counter(); k = 0
    while time.perf_counter() - t1 < 2.0:       # the reference's own style: one configuration per call
        orc.rnea_grad(om, q[k % chunk], qd[k % chunk], qdd[k % chunk]); k += 1
    per_call = k / (time.perf_counter() - t1)
    if main is None:
        main = {"value": np_rate, "unit": "evals/s", "cores": 1, "kind": "port",
                "sample": f"{done} rows through oracle/rbd_oracle.py (numpy fp64, batch-vectorised, 1 thread)"}
    main["numpy_oracle_batched_evals_per_s_1thread"] = np_rate
    main["numpy_oracle_one_config_per_call_evals_per_s"] = per_call
    main["host_cpus"] = os.cpu_count()
    main.update(extra)
    return main


def sources_digest():
    """sha256 over the kernel sources + C-ABI header: profiles/hbm_traffic.json is only attached to a
    bench line produced by the same code."""
    import hashlib
    h = hashlib.sha256()
    cs = os.path.join(ROOT, "rbdreference_amd", "csrc")
    for f in sorted(os.listdir(cs)):
        h.update(f.encode()); h.update(open(os.path.join(cs, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "rbd_hip.h"), "rb").read())
    return h.hexdigest()[:16]


def pick_graph(step, steps, dist, dev):
    """--graph auto, strong scaling on several GPUs: an UNTIMED calibration decides how the K timed launches are issued.
    The same K launches, eager and as one replay of a HIP graph, are timed once each (after a replay for the graph's
    upload and 0.2 s of replays: both candidates at the working clock, interleaved, best of three); the graph is used when
    it is at least 3 % faster on the slowest rank.  Why not always: measured this way the two are within 1-2 % of each other
    at every shard size on most boxes (19.8 vs 20.1 us per launch at 131 072 rows, 37.3 vs 37.1 at 262 144, 69.3 vs 68.3 at
    524 288), on one box of this round the replay was 18.7 against 20.5 us at 131 072 -- so the choice is measured on the box
    that runs.  (A replay timed right after the capture, with the GPU back at its idle clock, reads 10-25 % slow: the first
    version of this bench did that.)  Returns (use_graph, GraphSteps or None, the two times).""" 
    n = max(5, min(steps, 20))
    gs = GraphSteps(step, steps, dev)

    def spin(fn, seconds):                   # both candidates are timed at the working clock (the capture left the GPU idle)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            fn()
            torch.cuda.synchronize()

    def timed(fn, per):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / per

    def eager_n():
        for _ in range(n):
            step()
    spin(gs, 0.2)
    t_eager, t_graph = 1e30, 1e30
    for _ in range(3):                       # interleaved, best of three each
        t_graph = min(t_graph, timed(gs, steps))
        t_eager = min(t_eager, timed(eager_n, n))
    if dist is not None:
        t = torch.tensor([t_eager, t_graph], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        t_eager, t_graph = t[0].item(), t[1].item()
    use = t_graph < 0.97 * t_eager
    return use, (gs if use else None), {"eager_ms_per_launch": t_eager, "graph_ms_per_launch": t_graph}


def timed_steps(step, steps, warmup, dist, dev, graph=False, prebuilt=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize on both sides.
    Returns (wall seconds, HIP-event ms per step on the launch stream): this rank's.
    graph: the K steps are captured (untimed) into one HIP graph and the timed region is its replay -- the same K launches."""
    run = None
    if graph:
        gs = prebuilt if prebuilt is not None else GraphSteps(step, steps, dev)
        run = gs
    for _ in range(warmup):
        step()
    if run is not None:
        run()                                    # one untimed replay (graph upload)
        torch.cuda.synchronize()
        t_spin = time.perf_counter()             # capture + instantiation left the GPU idle: untimed replays back to the working clock
        while time.perf_counter() - t_spin < 0.2:
            run()
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    if run is not None:
        run()
    else:
        for _ in range(steps):
            step()
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0          # this rank's K steps, device-synchronised; MAX over ranks below
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    kern_ms = ev0.elapsed_time(ev1) / steps
    if dist is not None:
        t = torch.tensor([wall, kern_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kern_ms = t[0].item(), t[1].item()
    return wall, kern_ms


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback)")
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("RBD_BENCH_FORCE_DIST") == "1":   # (forcing lets a 1-GPU box rehearse the RCCL path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from rbdreference_amd import RBDReference, iiwa_like
    from rbdreference_amd._lib import RBD_OP_RNEA_GRAD
    from rbdreference_amd.dist import shard_bounds
    robot = iiwa_like()
    rbd = RBDReference(robot, build=False)     # prebuilt by __graft_entry__.build(); raises if missing
    B = args.batch
    nsets = max(1, args.rotate)

    def make_step(mode):
        """strong: this rank's rows of the GLOBAL batch (seed per set, identical on every rank, sliced
        with shard_bounds); weak: `B` rows of this rank's own."""
        sets = []
        for k in range(nsets):
            if mode == "strong":
                a, b = shard_bounds(B, world, rank)
                full = make_inputs(B, N_DOF, 3 + 17 * k, "cpu")
                sets.append(tuple(x[a:b].contiguous().to(dev) for x in full))
            else:
                sets.append(make_inputs(B, N_DOF, 3 + 17 * k + 1000 * rank, dev))
        return GradStep(rbd, sets)

    step = make_step(args.scaling)
    rows_rank = step.B
    # strong scaling = shards of ONE global batch: kernel selection is pinned to the global row count, as
    # rbdreference_amd.dist.ShardedRBD does, so that every rank runs the kernel the unsharded call would run
    from rbdreference_amd._lib import RBD_OPT_SELECT_BATCH
    rbd._lib.set_option(RBD_OPT_SELECT_BATCH, B if (args.scaling == "strong" and world > 1) else 0)

    # Clock ramp (untimed, before the W warm-up steps): from idle the GPU needs ~25 ms of sustained
    # load to reach its working clock -- with only 10 warm-up launches (3 ms) the timed launches
    # measured 287 us instead of 243 us on the same box.
    if dist is not None:
        dist.barrier()                       # builds the RCCL communicator (100s of ms): before the clock ramp
        torch.cuda.synchronize()
    def ramp():
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.3:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
    ramp()
    use_graph, prebuilt, graph_cal = args.graph == "on", None, None
    if args.graph == "auto" and (world > 1 or os.environ.get("RBD_BENCH_CALIBRATE")) and args.scaling == "strong":   # (env: rehearse the calibration on one GPU)
        use_graph, prebuilt, graph_cal = pick_graph(step, args.steps, dist, dev)
        ramp()                               # (the capture and instantiation left the GPU idle: back to the working clock)
    wall, kern_ms = timed_steps(step, args.steps, args.warmup, dist, dev, graph=use_graph, prebuilt=prebuilt)

    other = None
    if world > 1:                            # the other scaling mode, same run, reported beside the headline
        omode = "weak" if args.scaling == "strong" else "strong"
        ostep = make_step(omode)
        rbd._lib.set_option(RBD_OPT_SELECT_BATCH, B if omode == "strong" else 0)
        og, opre = args.graph == "on", None
        if args.graph == "auto" and omode == "strong":
            og, opre, _ = pick_graph(ostep, args.steps, dist, dev)
        ow, ok = timed_steps(ostep, args.steps, args.warmup, dist, dev, graph=og, prebuilt=opre)
        rbd._lib.set_option(RBD_OPT_SELECT_BATCH, 0)
        orows = B if omode == "strong" else world * B
        other = {"scaling": omode, "value": orows * args.steps / ow, "unit": "evals/s", "ms_per_step": ow / args.steps * 1e3,
                 "kernel_ms": ok, "rows_per_gpu": ostep.B, "global_batch": orows}
        del ostep

    rbd._lib.set_option(RBD_OPT_SELECT_BATCH, 0)
    # parity spot-check of what was just timed (first 256 rows of buffer set 0 vs the fp64 oracle), rank 0
    parity = None
    if rank == 0:
        from oracle import rbd_oracle as orc
        om = orc.model_from_robot(robot)
        k = min(256, rows_rank)
        q, qd, qdd = step.inputs[0]
        c_ref, dc_ref = orc.rnea_grad(om, q[:k].double().cpu().numpy(), qd[:k].double().cpu().numpy(),
                                      qdd[:k].double().cpu().numpy(), return_c=True)
        got = step.dc[:k].double().cpu().numpy().reshape(k, -1)
        ref = dc_ref.reshape(k, -1)
        parity = float(np.max(np.max(np.abs(got - ref), 1) / np.max(np.abs(ref), 1)))

    if rank == 0:
        rows_global = B if args.scaling == "strong" else world * B
        evals = rows_global * args.steps
        value = evals / wall
        achieved = BYTES_PER_EVAL * rows_rank / (kern_ms * 1e-3) / 1e9       # one launch of the slowest rank
        kernel = rbd._lib.kernel_name(RBD_OP_RNEA_GRAD, 4, rows_rank)
        digest = sources_digest()
        traffic = None
        traffic_note = "no profiles/hbm_traffic.json for this kernel / batch / source digest"
        valu = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("batch") == rows_rank and tj.get("kernel") == kernel and tj.get("sources_digest") == digest:
                    traffic = tj.get("hbm_bytes_per_launch")
                    valu = tj.get("valu")
                    traffic_note = tj.get("note")
            except Exception:
                traffic = None
        # the vector-pipe side of the roofline, from the SQ counters committed with the traffic file (same kernel, same
        # sources): flops and instruction counts are per launch and do not depend on the box; the time is this run's
        valu_frac = valu_tflops = valu_issue = None
        if valu and valu.get("flops_per_launch"):
            valu_tflops = valu["flops_per_launch"] / (kern_ms * 1e-3) / 1e12
            valu_frac = valu_tflops / FP32_VALU_PEAK_TFLOPS
            valu_issue = valu.get("simd_valu_busy_frac")
        hbm_frac = achieved / HBM_PEAK_GBS
        bound = "valu" if (valu_frac is not None and valu_frac > hbm_frac) else "hbm"
        out = {
            "metric": "batched RNEA+grad evals/s (7-DoF)", "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: 7-DoF iiwa-like rnea_grad (c + dc_du), fp32, "
                                   f"global batch {rows_global} rows per step, {rows_rank} rows per GPU "
                                   f"({args.scaling} scaling, batch-sharded), inputs resident in HBM, "
                                   f"{nsets} buffer sets rotated",
                       "robot": "iiwa_like", "batch_per_gpu": rows_rank, "global_batch": rows_global,
                       "buffer_sets": nsets,
                       "launch": (f"one replay of a HIP graph of the {args.steps} launches" if use_graph else "eager launches"),
                       "launch_calibration": graph_cal,
                       "parallelism": f"batch-shard x{world} (no data-path collective)"},
            "roofline": {"bound": bound, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                         "frac_from_ms_per_step": (rows_rank * BYTES_PER_EVAL / (wall / args.steps) / 1e9) / HBM_PEAK_GBS,
                         "frac_uses": "kernel_ms (HIP events around the K launches on the launch stream); frac_from_ms_per_step divides by the wall-clock step instead",
                         "algorithmic_bytes_per_eval": BYTES_PER_EVAL, "kernel": kernel,
                         "kernel_ms": kern_ms, "sources_digest": digest, "valu": valu,
                         "valu_frac_of_peak": valu_frac, "valu_tflops": valu_tflops,
                         "valu_peak_tflops": FP32_VALU_PEAK_TFLOPS, "valu_issue_frac": valu_issue,
                         "note": "frac = algorithmic bytes (SURVEY.md §8d: 504 B per evaluation) / this run's launch time / "
                                 "8 TB/s.  valu_frac_of_peak = fp32 flops per launch (2 x FMA + MUL + ADD + TRANS wave-"
                                 "instructions x 64 lanes, from the committed SQ profile of this kernel and these "
                                 "sources) / this run's launch time / 157.3 TFLOP/s; valu_issue_frac = all VALU "
                                 "wave-instructions x 2 cycles per SIMD / launch cycles (what the vector pipe is busy); "
                                 "`bound` names the larger of frac and valu_frac_of_peak"},
            "parity_max_rel_err_first_256_rows": parity,
        }
        if other is not None:
            out[other["scaling"]] = other
        q, qd, qdd = step.inputs[0]
        if not args.no_extra and world == 1:
            extra = {}
            # BASELINE configs[1]: B = 4096, rnea + rnea_grad back to back
            q4, qd4, qdd4 = make_inputs(4096, N_DOF, 1, dev)
            s4 = GradStep(rbd, [(q4, qd4, qdd4)])
            ms = time_extra_ms(s4, 200, 20)
            extra["cfg1_iiwa_rnea_grad_B4096_f32"] = {"ms_per_launch": ms, "evals_per_s": 4096 / (ms * 1e-3)}
            ms = time_extra_ms(lambda: rbd.rnea(q4, qd4, qdd4), 100, 10)
            extra["cfg1_iiwa_rnea_cvaf_B4096_f32_api"] = {"ms_per_call": ms, "evals_per_s": 4096 / (ms * 1e-3)}
            # configs[1] as one call: rnea + rnea_grad -> (c, v, a, f, dc_du), eager launches back to back and
            # the same launches replayed from a HIP graph (launch-amortised)
            try:
                from rbdreference_amd._lib import RBD_OP_RNEA_GRAD as _OP
                o4 = [torch.empty(sh, device=dev, dtype=torch.float32) for sh in
                      ((4096, 7), (4096, 6, 7), (4096, 6, 7), (4096, 6, 7), (4096, 7, 14))]
                fn4 = rbd._fn("rbd_rnea_with_grad", torch.float32)

                def fused():
                    st = torch.cuda.current_stream(dev).cuda_stream
                    rc = fn4(q4.data_ptr(), qd4.data_ptr(), qdd4.data_ptr(), -9.81, 0, 4096,
                             *[t.data_ptr() for t in o4], st)
                    if rc != 0:
                        rbd._lib.check(rc)
                ms = time_extra_ms(fused, 200, 20)
                ent = {"ms_per_launch_eager": ms, "evals_per_s_eager": 4096 / (ms * 1e-3),
                       "kernel": rbd._lib.kernel_name(_OP, 4, 4096), "outputs": "c, v, a, f, dc_du",
                       "alg_GBps_eager": 4096 * (22 * 7 + 2 * 49) * 4 / (ms * 1e-3) / 1e9}
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    fused()
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                NG = 50
                with torch.cuda.graph(gr):
                    for _ in range(NG):
                        fused()
                msg = time_extra_ms(gr.replay, 20, 3) / NG
                ent["ms_per_launch_graph"] = msg
                ent["evals_per_s_graph"] = 4096 / (msg * 1e-3)
                extra["cfg1_iiwa_rnea+rnea_grad_one_call_B4096_f32"] = ent
            except Exception as e:
                extra["cfg1_one_call_error"] = repr(e)
            Bq = q.shape[0]
            ms = time_extra_ms(lambda: rbd.minv(q), 10, 2)
            extra["iiwa_minv_B%d_f32_api" % Bq] = {"ms_per_call": ms, "evals_per_s": Bq / (ms * 1e-3),
                                                    "alg_GBps": Bq * (7 + 49) * 4 / (ms * 1e-3) / 1e9}
            ms = time_extra_ms(lambda: rbd.rnea(q, qd, qdd), 10, 2)
            extra["iiwa_rnea_cvaf_B%d_f32_api" % Bq] = {"ms_per_call": ms, "evals_per_s": Bq / (ms * 1e-3),
                                                         "alg_GBps": Bq * 22 * 7 * 4 / (ms * 1e-3) / 1e9}
            ms = time_extra_ms(lambda: rbd.aba(q, qd, qdd), 10, 2)
            extra["iiwa_aba_B%d_f32_api" % Bq] = {"ms_per_call": ms, "evals_per_s": Bq / (ms * 1e-3),
                                                   "alg_GBps": Bq * 4 * 7 * 4 / (ms * 1e-3) / 1e9}
            # forward_dynamics_grad (what MPC consumers call): algorithmic bytes (3 n + 2 n^2) s per evaluation
            # (q, qd, u in; [qdd_dq | qdd_dqd] out); two launches for a chain: fd_pre_kernel + the chain gradient kernel with the
            # -Minv epilogue (rbd_fd_chain.h)
            ms = time_extra_ms(lambda: rbd.forward_dynamics_grad(q, qd, qdd), 10, 2)
            extra["iiwa_forward_dynamics_grad_B%d_f32_api" % Bq] = {"ms_per_call": ms, "evals_per_s": Bq / (ms * 1e-3),
                                                                    "alg_GBps": Bq * (3 * 7 + 2 * 49) * 4 / (ms * 1e-3) / 1e9,
                                                                    "kernels": "fd_pre_kernel<float> + rnea_grad_idsva_pipe_kernel<float,true,true>"}
            try:
                # the reference's own arithmetic (fp64) on the headline robot, and the first-use path (the model-handle
                # library, include/rbd_generic.h: no per-robot compilation) on one rank's share of the batch
                q64, qd64, qdd64 = q.double(), qd.double(), qdd.double()
                dc64 = torch.empty((Bq, 7, 14), dtype=torch.float64, device=dev)
                ms = time_extra_ms(lambda: rbd.rnea_grad(q64, qd64, qdd64, out=dc64), 5, 2)
                extra["iiwa_rnea_grad_B%d_f64_api" % Bq] = {"ms_per_call": ms, "evals_per_s": Bq / (ms * 1e-3),
                                                             "alg_GBps": Bq * (21 + 98) * 8 / (ms * 1e-3) / 1e9,
                                                             "kernel": rbd._lib.kernel_name(RBD_OP_RNEA_GRAD, 8, Bq)}
                ms = time_extra_ms(lambda: rbd.forward_dynamics_grad(q64, qd64, qdd64), 5, 2)
                extra["iiwa_forward_dynamics_grad_B%d_f64_api" % Bq] = {"ms_per_call": ms, "evals_per_s": Bq / (ms * 1e-3),
                                                                        "alg_GBps": Bq * (3 * 7 + 2 * 49) * 8 / (ms * 1e-3) / 1e9,
                                                                        "kernels": "fd_pre_kernel<double> + rnea_grad_idsva_kernel<double,true,true>"}
                del q64, qd64, qdd64, dc64
                gen = RBDReference(robot, build=False, generic="only")
                Bg = 131072
                dcg = torch.empty((Bg, 7, 14), dtype=torch.float32, device=dev)
                ms = time_extra_ms(lambda: gen.rnea_grad(q[:Bg], qd[:Bg], qdd[:Bg], out=dcg), 3, 1)
                extra["first_use_model_handle_library_iiwa_rnea_grad_B%d_f32" % Bg] = {
                    "ms_per_call": ms, "evals_per_s": Bg / (ms * 1e-3), "kernel": gen._lib.kernel_name(RBD_OP_RNEA_GRAD, 4, Bg),
                    "note": "librbd_generic.so: the robot is a run-time table; serves a never-built robot until its own library is ready"}
                del gen, dcg
            except Exception as e:
                extra["f64_or_generic_error"] = repr(e)
            try:
                # the same kernel on a 7-chain with DENSE joint frames and inertias (random_chain_n7): what
                # the iiwa-like robot's structural zeros / unit entries are worth
                from __graft_entry__ import test_robots
                rc7 = RBDReference([r for r in test_robots() if r.name == "random_chain_n7"][0], build=False)
                sd = GradStep(rc7, [(q, qd, qdd)])
                ms = time_extra_ms(sd, 10, 2)
                extra["dense_random_chain_n7_rnea_grad_B%d_f32" % Bq] = {
                    "ms_per_launch": ms, "evals_per_s": Bq / (ms * 1e-3), "alg_GBps": Bq * BYTES_PER_EVAL / (ms * 1e-3) / 1e9,
                    "kernel": rc7._lib.kernel_name(RBD_OP_RNEA_GRAD, 4, Bq)}
                del sd
            except Exception as e:
                extra["dense_chain_error"] = repr(e)
            try:
                from rbdreference_amd import atlas_like, quadruped_like
                atlas = atlas_like()
                ra = RBDReference(atlas, build=False)
                qa, qda, qdda = make_inputs(16384, 30, 2, dev)
                ms = time_extra_ms(lambda: ra.minv(qa), 20, 3)
                extra["cfg2_atlas_minv_B16384_f32"] = {"ms_per_call": ms, "evals_per_s": 16384 / (ms * 1e-3),
                                                       "alg_GBps": 16384 * 3720 / (ms * 1e-3) / 1e9}
                # measured parity of that launch (the error itself, not only a pass/fail)
                from oracle import rbd_oracle as orc
                oma = orc.model_from_robot(atlas)
                Mi = ra.minv(qa[:256]).double().cpu().numpy().reshape(256, -1)
                Mr = orc.minv(oma, qa[:256].double().cpu().numpy()).reshape(256, -1)
                extra["cfg2_atlas_minv_B16384_f32"]["max_rel_err_first_256_rows"] = float(
                    np.max(np.max(np.abs(Mi - Mr), 1) / np.max(np.abs(Mr), 1)))
                ms = time_extra_ms(lambda: ra.rnea(qa, qda, qdda), 20, 3)
                extra["cfg2_atlas_rnea_cvaf_B16384_f32"] = {"ms_per_call": ms, "evals_per_s": 16384 / (ms * 1e-3),
                                                            "alg_GBps": 16384 * 2640 / (ms * 1e-3) / 1e9}
                # the same two entry points through the C-ABI with pre-allocated outputs (what a host that owns
                # its buffers pays: the per-call numbers above include four / one torch allocations)
                Ba, na = 16384, 30
                ca = torch.empty((Ba, na), device=dev, dtype=torch.float32)
                va = torch.empty((Ba, 6, na), device=dev, dtype=torch.float32); aa = torch.empty_like(va); fa = torch.empty_like(va)
                Ma = torch.empty((Ba, na, na), device=dev, dtype=torch.float32)
                wsb = int(ra._lib.lib.rbd_minv_workspace_bytes(Ba, 4))
                wsa = torch.empty((max(wsb, 16),), device=dev, dtype=torch.uint8)
                st = torch.cuda.current_stream(dev).cuda_stream
                f_rnea = ra._fn("rbd_rnea", torch.float32); f_minv = ra._fn("rbd_minv", torch.float32)
                ms = time_extra_ms(lambda: f_rnea(qa.data_ptr(), qda.data_ptr(), qdda.data_ptr(), -9.81, Ba, ca.data_ptr(),
                                                  va.data_ptr(), aa.data_ptr(), fa.data_ptr(), st), 100, 10)
                extra["cfg2_atlas_rnea_cvaf_B16384_f32"].update(ms_per_launch_abi=ms, alg_GBps_abi=Ba * 2640 / (ms * 1e-3) / 1e9,
                                                                kernel=ra._lib.kernel_name(0, 4, Ba))
                ms = time_extra_ms(lambda: f_minv(qa.data_ptr(), Ba, 1, Ma.data_ptr(), wsa.data_ptr(), wsb, st), 100, 10)
                extra["cfg2_atlas_minv_B16384_f32"].update(ms_per_launch_abi=ms, alg_GBps_abi=Ba * 3720 / (ms * 1e-3) / 1e9,
                                                           kernel=ra._lib.kernel_name(2, 4, Ba))
                del ca, va, aa, fa, Ma, wsa
                ms = time_extra_ms(lambda: ra.rnea_grad(qa, qda, qdda, return_c=True), 20, 3)
                extra["atlas_rnea_grad_B16384_f32"] = {"ms_per_call": ms, "evals_per_s": 16384 / (ms * 1e-3),
                                                       "alg_GBps": 16384 * (4 * 30 + 2 * 900) * 4 / (ms * 1e-3) / 1e9}
                # the same robot in the reference's own precision: the workspace tree kernel (rbd_idsva_tree_ws.h)
                qa8, qda8, qdda8 = make_inputs(16384, 30, 2, dev, torch.float64)
                ms = time_extra_ms(lambda: ra.rnea_grad(qa8, qda8, qdda8, return_c=True), 10, 2)
                extra["atlas_rnea_grad_B16384_f64"] = {"ms_per_call": ms, "evals_per_s": 16384 / (ms * 1e-3),
                                                       "alg_GBps": 16384 * (4 * 30 + 2 * 900) * 8 / (ms * 1e-3) / 1e9,
                                                       "kernel": ra._lib.kernel_name(1, 8, 16384)}
                del qa8, qda8, qdda8
                ms = time_extra_ms(lambda: ra.forward_dynamics_grad(qa, qda, qdda), 10, 2)
                extra["atlas_forward_dynamics_grad_B16384_f32_api"] = {
                    "ms_per_call": ms, "evals_per_s": 16384 / (ms * 1e-3), "alg_GBps": 16384 * (3 * 30 + 2 * 900) * 4 / (ms * 1e-3) / 1e9,
                    "kernels": "rnea_kernel + minv_fused_kernel + rnea_grad_tree_kernel + neg_mm_kernel<float,30>"}
                rq = RBDReference(quadruped_like(), build=False)
                qq, qdq, qddq = make_inputs(65536, 12, 4, dev, torch.float64)
                ms1 = time_extra_ms(lambda: rq.rnea_grad(qq, qdq, qddq, return_c=True), 10, 2)
                ms2 = time_extra_ms(lambda: rq.minv(qq), 10, 2)
                extra["cfg4_quadruped_rnea_grad+minv_B65536_f64"] = {
                    "ms_rnea_grad": ms1, "ms_minv": ms2, "evals_per_s": 65536 / ((ms1 + ms2) * 1e-3),
                    "alg_GBps": 65536 * 3840 / ((ms1 + ms2) * 1e-3) / 1e9}
                ms = time_extra_ms(lambda: rq.forward_dynamics_grad(qq, qdq, qddq), 10, 2)
                extra["quadruped_forward_dynamics_grad_B65536_f64_api"] = {
                    "ms_per_call": ms, "evals_per_s": 65536 / (ms * 1e-3), "alg_GBps": 65536 * (3 * 12 + 2 * 144) * 8 / (ms * 1e-3) / 1e9,
                    "kernels": "rnea_kernel + minv_lane_kernel + rnea_grad_kernel<double,true,true> (one block per leg, -Minv epilogue)"}
                # floating base: a 13-body trunk + four legs, nv = 18, fp32
                from rbdreference_amd import floating_quadruped_like
                rf = RBDReference(floating_quadruped_like(), build=False)
                Bf = 65536
                qf, qdf, qddf = make_inputs(Bf, rf.nv, 5, dev)
                nvf, nbf = rf.nv, rf.n
                t1 = time_extra_ms(lambda: rf.rnea(qf, qdf, qddf), 10, 2)
                t2 = time_extra_ms(lambda: rf.minv(qf), 10, 2)
                t3 = time_extra_ms(lambda: rf.rnea_grad(qf, qdf, qddf, return_c=True), 5, 1)
                extra["floating_quadruped_B65536_f32"] = {
                    "ms_rnea_cvaf": t1, "alg_GBps_rnea": Bf * (4 * nvf + 18 * nbf) * 4 / (t1 * 1e-3) / 1e9,
                    "ms_minv": t2, "alg_GBps_minv": Bf * (nvf + nvf * nvf) * 4 / (t2 * 1e-3) / 1e9,
                    "ms_rnea_grad": t3, "alg_GBps_rnea_grad": Bf * (4 * nvf + 2 * nvf * nvf) * 4 / (t3 * 1e-3) / 1e9}
                t4 = time_extra_ms(lambda: rf.forward_dynamics_grad(qf, qdf, qddf), 5, 1)
                extra["floating_quadruped_forward_dynamics_grad_B65536_f32_api"] = {
                    "ms_per_call": t4, "evals_per_s": Bf / (t4 * 1e-3), "alg_GBps": Bf * (3 * nvf + 2 * nvf * nvf) * 4 / (t4 * 1e-3) / 1e9,
                    "kernels": "minv_fbm (bias force, Minv, qdd in one launch) + rnea_grad_fbw + neg_mm_kernel<float,18>"}
            except Exception as e:  # the headline line must still be printed
                extra["error"] = repr(e)
            try:
                # the per-rank launch of configs[3] at N = 8: 131 072 rows, kernel selection pinned to the global batch
                # as ShardedRBD does (a prediction the first real 8-GPU SCALE run can be checked against)
                Bs = (1 << 20) // 8
                qs, qds, qdds = (x[:Bs].contiguous() for x in (q, qd, qdd))
                with rbd.shard_of(1 << 20):
                    ss = GradStep(rbd, [(qs, qds, qdds)])
                    ms = time_extra_ms(ss, 50, 10)
                    kn = rbd._lib.kernel_name(RBD_OP_RNEA_GRAD, 4, Bs)
                    NG = 50
                    gs = GraphSteps(ss, NG, dev)
                    msg = time_extra_ms(gs, 20, 3) / NG
                    # the same independent launches issued on TWO streams alternately (two rotating buffer sets): the row stores
                    # of one launch overlap the arithmetic of the next.  Information only: the headline and --gpus N use one stream
                    # (one kernel at a time is what the roofline fields and the rocprof stats describe).
                    ss2 = GradStep(rbd, [(qs, qds, qdds), (qs, qds, qdds)])
                    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

                    def two_streams(n=50):
                        for k in range(n):
                            ss2.on_stream((sa if k & 1 else sb).cuda_stream)
                        sa.synchronize(); sb.synchronize()

                    def wall_ms(fn, n):
                        fn(n); best = 1e30
                        for _ in range(5):
                            t0 = time.perf_counter(); fn(n); best = min(best, (time.perf_counter() - t0) / n * 1e3)
                        return best
                    ms2 = wall_ms(two_streams, 200)
                extra["cfg3_per_rank_shard_B131072_of_1M_f32"] = {
                    "ms_per_launch": ms, "evals_per_s": Bs / (ms * 1e-3), "alg_GBps": Bs * BYTES_PER_EVAL / (ms * 1e-3) / 1e9,
                    "kernel": kn, "predicted_8gpu_strong_scaling_evals_per_s": (1 << 20) / (ms * 1e-3),
                    "ms_per_launch_graph": msg, "alg_GBps_graph": Bs * BYTES_PER_EVAL / (msg * 1e-3) / 1e9,
                    "predicted_8gpu_strong_scaling_evals_per_s_graph": (1 << 20) / (msg * 1e-3),
                    "ms_per_launch_two_streams": ms2, "alg_GBps_two_streams": Bs * BYTES_PER_EVAL / (ms2 * 1e-3) / 1e9,
                    "predicted_8gpu_strong_scaling_evals_per_s_two_streams": (1 << 20) / (ms2 * 1e-3),
                    "note": "one rank's share of the 1 048 576-row global batch; 8 ranks run it concurrently with no "
                            "data-path collective, so the N = 8 strong-scaling value is bounded by (1 M rows) / this time.  "
                            "`_graph`: the same launches replayed from one HIP graph of 50 (what bench.py --gpus N > 1 times "
                            "in strong mode: the launch path leaves the step).  `_two_streams`: the same independent launches "
                            "issued alternately on two streams (wall clock over 200 launches): a launch's row stores overlap the "
                            "next launch's arithmetic -- what a consumer with independent batches can have; not what the headline "
                            "or --gpus N measure"}
                del gs
                del ss
            except Exception as e:
                extra["cfg3_shard_error"] = repr(e)
            for ent in extra.values():            # every algorithmic rate also as a fraction of the 8 TB/s HBM peak
                if isinstance(ent, dict):
                    for k in [k for k in ent if k.startswith("alg_GBps")]:
                        ent["frac" + k[len("alg_GBps"):]] = ent[k] / HBM_PEAK_GBS
            # PMC traffic of the non-headline kernels (profiles/config_traffic.json, tools/make_config_traffic.py): attached
            # when it was measured on these sources; traffic / algorithmic bytes per launch beside each `frac`
            try:
                ct = json.load(open(os.path.join(ROOT, "profiles", "config_traffic.json")))
                if ct.get("sources_digest") == digest:
                    kern = ct["kernels"]
                    nvq, nbq = 18, 13                    # floating quadruped: velocities, bodies
                    plan = {   # extra entry -> (suffix, [kernels of one call], algorithmic bytes of one call)
                        "cfg2_atlas_minv_B16384_f32": ("", ["minv_fused_kernel<float>"], 16384 * 3720),
                        "cfg2_atlas_rnea_cvaf_B16384_f32": ("", ["rnea_segments_kernel<float,true>"], 16384 * 2640),
                        "atlas_rnea_grad_B16384_f32": ("", ["rnea_grad_tree_kernel<float,true>"], 16384 * (4 * 30 + 2 * 900) * 4),
                        "atlas_rnea_grad_B16384_f64": ("", ["rnea_grad_tree_ws_kernel<double,true>"], 16384 * (4 * 30 + 2 * 900) * 8),
                        "cfg4_quadruped_rnea_grad+minv_B65536_f64": ("", ["rnea_grad_kernel<double,true,false>", "minv_lane_kernel<double>"], 65536 * 3840),
                        "iiwa_forward_dynamics_grad_B1048576_f32_api": ("", ["fd_pre_kernel<float>", "rnea_grad_idsva_pipe_kernel<float,true,true>"], (1 << 20) * (3 * 7 + 2 * 49) * 4),
                        "atlas_forward_dynamics_grad_B16384_f32_api": ("", ["rnea_kernel<float,false,false>", "minv_fused_kernel<float>", "rnea_grad_tree_kernel<float,true>", "neg_mm_kernel<float,30>"], 16384 * (3 * 30 + 2 * 900) * 4),
                        "floating_quadruped_forward_dynamics_grad_B65536_f32_api": ("", ["minv_fbm_kernel<float>", "rnea_grad_fbw_kernel<float,true>", "neg_mm_kernel<float,18>"], 65536 * (3 * nvq + 2 * nvq * nvq) * 4),
                    }
                    fbq = {"_rnea": (["rnea_fbw_kernel<float,true,0>"], 65536 * (4 * nvq + 18 * nbq) * 4),
                           "_minv": (["minv_fbm_kernel<float>"], 65536 * (nvq + nvq * nvq) * 4),
                           "_rnea_grad": (["rnea_grad_fbw_kernel<float,true>"], 65536 * (4 * nvq + 2 * nvq * nvq) * 4)}
                    for key, (sfx, ks, alg) in plan.items():
                        if key in extra and all(k in kern for k in ks):
                            tb = sum(kern[k][0]["hbm_bytes_per_launch"] for k in ks)
                            extra[key]["traffic_bytes_per_call" + sfx] = tb
                            extra[key]["traffic_over_algorithmic" + sfx] = tb / alg
                    if "floating_quadruped_B65536_f32" in extra:
                        for sfx, (ks, alg) in fbq.items():
                            if all(k in kern for k in ks):
                                tb = sum(kern[k][0]["hbm_bytes_per_launch"] for k in ks)
                                extra["floating_quadruped_B65536_f32"]["traffic_over_algorithmic" + sfx] = tb / alg
                    extra["traffic_note"] = ct.get("note")
            except Exception as e:
                extra["traffic_error"] = repr(e)
            out["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(robot, 3)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
