#!/usr/bin/env python3
"""Atlas fp64 rnea_grad / minv at B = 16 384: the robot's own library against the model-handle library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, atlas_like
from tools.time_generic import t
robot = atlas_like(); B = 16384; dt = torch.float64
spec = RBDReference(robot, build=False, generic="never"); gen = RBDReference(robot, build=False, generic="only")
n = spec.n; rng = np.random.default_rng(3)
q, qd, qdd = (torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-3, 3, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n))))
dc = torch.empty((B, n, 2 * n), dtype=dt, device="cuda"); M = torch.empty((B, n, n), dtype=dt, device="cuda")
for nm, fn in (("rnea_grad", lambda r: r.rnea_grad(q, qd, qdd, out=dc)), ("minv", lambda r: r.minv(q, out=M)), ("rnea", lambda r: r.rnea(q, qd, qdd))):
    print(f"atlas fp64 B={B} {nm:10s} specialised {t(lambda: fn(spec), 5):9.1f} us   model-handle {t(lambda: fn(gen), 5):9.1f} us")
