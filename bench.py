#!/usr/bin/env python3
"""Headline benchmark: batched 7-DoF RNEA + gradient evaluations per second (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One *step* = one pass of the hot path over one batch: a single ``rbd_rnea_grad_f32`` launch that
maps ``B`` rows ``(q, qd, qdd)`` (already resident in HBM) to ``(c, dc_du)`` -- the work of ``B``
calls of the reference's ``rnea_grad`` (``/root/reference/RBDReference.py:1345-1368``).  Workload:
BASELINE.json configs[3] ("7-DoF iiwa rnea_grad batch=1M", the configuration the metric's
1/2/4/8-GPU series is quoted on); the full 1 048 576-row batch fits one MI355X, so every rank
evaluates its own 1 048 576 rows (weak scaling, no data-path collective: rows are independent,
SURVEY.md §8e).  configs[1] (B = 4096) is reported beside it under "extra" -- at that size a
launch is latency-bound and says nothing about the roofline.

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
    ap.add_argument("--batch", type=int, default=1 << 20, help="rows per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    return ap.parse_args()


def make_inputs(B, n, seed, device, dtype=torch.float32):
    """Synthetic inputs of SURVEY.md §8d: q ~ U(-pi, pi), qd, qdd ~ U(-1, 1); numpy PCG64 on the
    host, uploaded before the timed region."""
    rng = np.random.default_rng(seed)
    q = rng.uniform(-np.pi, np.pi, (B, n)).astype(np.float32 if dtype == torch.float32 else np.float64)
    qd = rng.uniform(-1, 1, (B, n)).astype(q.dtype)
    qdd = rng.uniform(-1, 1, (B, n)).astype(q.dtype)
    return tuple(torch.from_numpy(x).to(device) for x in (q, qd, qdd))


class GradStep:
    """Pre-allocated buffers + a direct C-ABI launch on torch's current stream."""

    def __init__(self, rbd, q, qd, qdd):
        self.rbd, self.q, self.qd, self.qdd = rbd, q, qd, qdd
        B, n = q.shape
        self.B = B
        self.c = torch.empty((B, n), device=q.device, dtype=q.dtype)
        self.dc = torch.empty((B, n, 2 * n), device=q.device, dtype=q.dtype)
        self.fn = rbd._fn("rbd_rnea_grad", q.dtype)
        self.stream = torch.cuda.current_stream(q.device).cuda_stream

    def __call__(self):
        rc = self.fn(self.q.data_ptr(), self.qd.data_ptr(), self.qdd.data_ptr(), -9.81, 0, self.B,
                     self.c.data_ptr(), self.dc.data_ptr(), self.stream)
        if rc != 0:
            self.rbd._lib.check(rc)


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


def time_extra_ms(fn, steps, warmup, reps=3):
    """Extras only: best of `reps` timed groups (one stalled launch on a shared box otherwise
    dominates a 20-launch average)."""
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
    robot = iiwa_like()
    rbd = RBDReference(robot, build=False)     # prebuilt by __graft_entry__.build(); raises if missing
    B = args.batch
    q, qd, qdd = make_inputs(B, N_DOF, 3 + rank, dev)
    step = GradStep(rbd, q, qd, qdd)

    # Clock ramp (untimed, before the W warm-up steps): from idle the GPU needs ~25 ms of sustained
    # load to reach its working clock -- with only 10 warm-up launches (3 ms) the timed launches
    # measured 287 us instead of 243 us on the same box.
    if dist is not None:
        dist.barrier()                       # builds the RCCL communicator (100s of ms): before the clock ramp
        torch.cuda.synchronize()
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.3:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0          # this rank's K steps, device-synchronised; MAX over ranks below
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    kern_ms = ev0.elapsed_time(ev1) / args.steps
    if dist is not None:
        t = torch.tensor([wall, kern_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, kern_ms = t[0].item(), t[1].item()

    # parity spot-check of what was just timed (first 256 rows vs the fp64 oracle), rank 0
    parity = None
    if rank == 0:
        from oracle import rbd_oracle as orc
        om = orc.model_from_robot(robot)
        k = min(256, B)
        c_ref, dc_ref = orc.rnea_grad(om, q[:k].double().cpu().numpy(), qd[:k].double().cpu().numpy(),
                                      qdd[:k].double().cpu().numpy(), return_c=True)
        got = step.dc[:k].double().cpu().numpy().reshape(k, -1)
        ref = dc_ref.reshape(k, -1)
        parity = float(np.max(np.max(np.abs(got - ref), 1) / np.max(np.abs(ref), 1)))

    if rank == 0:
        evals = world * B * args.steps
        value = evals / wall
        achieved = BYTES_PER_EVAL * B / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("batch") == B:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "batched RNEA+grad evals/s (7-DoF)", "value": value, "unit": "evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[3]: 7-DoF iiwa-like rnea_grad (c + dc_du), fp32, "
                                   f"B={B} rows per GPU per step, batch-sharded (weak), inputs resident in HBM",
                       "robot": "iiwa_like", "batch_per_gpu": B, "global_batch": world * B,
                       "parallelism": f"batch-shard x{world} (no data-path collective)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_eval": BYTES_PER_EVAL, "kernel": "rnea_grad_idsva_kernel<float,true,false>",
                         "kernel_ms": kern_ms,
                         "note": "kernel is FP32-VALU-bound (SURVEY.md §8d); HBM fraction from algorithmic bytes"},
            "parity_max_rel_err_first_256_rows": parity,
        }
        if not args.no_extra and world == 1:
            extra = {}
            # BASELINE configs[1]: B = 4096, rnea + rnea_grad back to back
            q4, qd4, qdd4 = make_inputs(4096, N_DOF, 1, dev)
            s4 = GradStep(rbd, q4, qd4, qdd4)
            ms = time_extra_ms(s4, 200, 20)
            extra["cfg1_iiwa_rnea_grad_B4096_f32"] = {"ms_per_launch": ms, "evals_per_s": 4096 / (ms * 1e-3)}
            ms = time_extra_ms(lambda: rbd.rnea(q4, qd4, qdd4), 100, 10)
            extra["cfg1_iiwa_rnea_cvaf_B4096_f32_api"] = {"ms_per_call": ms, "evals_per_s": 4096 / (ms * 1e-3)}
            ms = time_extra_ms(lambda: rbd.minv(q), 10, 2)
            extra["iiwa_minv_B%d_f32_api" % B] = {"ms_per_call": ms, "evals_per_s": B / (ms * 1e-3),
                                                   "alg_GBps": B * (7 + 49) * 4 / (ms * 1e-3) / 1e9}
            ms = time_extra_ms(lambda: rbd.rnea(q, qd, qdd), 10, 2)
            extra["iiwa_rnea_cvaf_B%d_f32_api" % B] = {"ms_per_call": ms, "evals_per_s": B / (ms * 1e-3),
                                                        "alg_GBps": B * 22 * 7 * 4 / (ms * 1e-3) / 1e9}
            ms = time_extra_ms(lambda: rbd.aba(q, qd, qdd), 10, 2)
            extra["iiwa_aba_B%d_f32_api" % B] = {"ms_per_call": ms, "evals_per_s": B / (ms * 1e-3),
                                                  "alg_GBps": B * 4 * 7 * 4 / (ms * 1e-3) / 1e9}
            ms = time_extra_ms(lambda: rbd.forward_dynamics_grad(q, qd, qdd), 10, 2)
            extra["iiwa_forward_dynamics_grad_B%d_f32_api" % B] = {"ms_per_call": ms, "evals_per_s": B / (ms * 1e-3)}
            try:
                from rbdreference_amd import atlas_like, quadruped_like
                ra = RBDReference(atlas_like(), build=False)
                qa, qda, qdda = make_inputs(16384, 30, 2, dev)
                ms = time_extra_ms(lambda: ra.minv(qa), 20, 3)
                extra["cfg2_atlas_minv_B16384_f32"] = {"ms_per_call": ms, "evals_per_s": 16384 / (ms * 1e-3),
                                                       "alg_GBps": 16384 * 3720 / (ms * 1e-3) / 1e9}
                ms = time_extra_ms(lambda: ra.rnea(qa, qda, qdda), 20, 3)
                extra["cfg2_atlas_rnea_cvaf_B16384_f32"] = {"ms_per_call": ms, "evals_per_s": 16384 / (ms * 1e-3),
                                                            "alg_GBps": 16384 * 2640 / (ms * 1e-3) / 1e9}
                ms = time_extra_ms(lambda: ra.rnea_grad(qa, qda, qdda, return_c=True), 20, 3)
                extra["atlas_rnea_grad_B16384_f32"] = {"ms_per_call": ms, "evals_per_s": 16384 / (ms * 1e-3),
                                                       "alg_GBps": 16384 * (4 * 30 + 2 * 900) * 4 / (ms * 1e-3) / 1e9}
                rq = RBDReference(quadruped_like(), build=False)
                qq, qdq, qddq = make_inputs(65536, 12, 4, dev, torch.float64)
                ms1 = time_extra_ms(lambda: rq.rnea_grad(qq, qdq, qddq, return_c=True), 10, 2)
                ms2 = time_extra_ms(lambda: rq.minv(qq), 10, 2)
                extra["cfg4_quadruped_rnea_grad+minv_B65536_f64"] = {
                    "ms_rnea_grad": ms1, "ms_minv": ms2, "evals_per_s": 65536 / ((ms1 + ms2) * 1e-3),
                    "alg_GBps": 65536 * 3840 / ((ms1 + ms2) * 1e-3) / 1e9}
            except Exception as e:  # the headline line must still be printed
                extra["error"] = repr(e)
            out["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(robot, 3)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
