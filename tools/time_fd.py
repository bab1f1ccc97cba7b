#!/usr/bin/env python3
"""Time forward_dynamics / forward_dynamics_grad / aba (env ROBOT, DTYPE=f32|f64, B; default iiwa fp32 1M) through the API."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference
from rbdreference_amd.robot import BUILTIN_ROBOTS
name = os.environ.get("ROBOT", "iiwa_like")
r = RBDReference(BUILTIN_ROBOTS[name](), build=False)
B = int(os.environ.get("B", 1 << 20))
dt = torch.float64 if os.environ.get("DTYPE", "f32") == "f64" else torch.float32
n = r.n
rng = np.random.default_rng(0)
q, qd, u = (torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-3, 3, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-5, 5, (B, n))))
print(f"{name} n={n} B={B} {dt}")
for name, fn in (("forward_dynamics", lambda: r.forward_dynamics(q, qd, u)), ("forward_dynamics_grad", lambda: r.forward_dynamics_grad(q, qd, u)),
                 ("aba", lambda: r.aba(q, qd, u)),
                 ("rnea_grad", lambda: r.rnea_grad(q, qd, u)), ("minv", lambda: r.minv(q))):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{name:24s} {ms*1e3:8.1f} us per call   {B/(ms*1e-3)/1e9:6.2f} G evals/s")
