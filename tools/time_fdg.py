#!/usr/bin/env python3
"""forward_dynamics_grad of the BASELINE robots + the floating quadruped through the API (pooled outputs), HIP events,
with the algorithmic HBM rate ((3 n + 2 n^2) s bytes per evaluation: q, qd, u in, dqdd_du out) next to each.

    python tools/time_fdg.py            # on the GPU box
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, builtin_robot, floating_quadruped_like

CASES = [("iiwa_like", 1 << 20, torch.float32), ("atlas_like", 16384, torch.float32), ("quadruped_like", 65536, torch.float64),
         ("fb_quadruped_like", 65536, torch.float32), ("iiwa_like", 1 << 20, torch.float64)]
for name, B, dt in CASES:
    robot = floating_quadruped_like() if name.startswith("fb_") else builtin_robot(name)
    r = RBDReference(robot, build=False, generic="never")
    nv = r.nv
    rng = np.random.default_rng(0)
    q, qd, u = (torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-1, 1, (B, nv)), rng.uniform(-1, 1, (B, nv)), rng.uniform(-2, 2, (B, nv))))
    fn = lambda: r.forward_dynamics_grad(q, qd, u)      # noqa: E731
    best = 1e9
    for _ in range(4):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
    esz = 4 if dt == torch.float32 else 8
    alg = (3 * nv + 2 * nv * nv) * esz * B
    print(f"{name:20s} B={B:8d} {str(dt):14s} forward_dynamics_grad {best:8.1f} us   {B / best / 1e3:6.2f} G evals/s   alg {alg / best / 1e3:7.0f} GB/s  frac {alg / best / 1e3 / 8000:.3f}")
