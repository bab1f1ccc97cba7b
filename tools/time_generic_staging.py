#!/usr/bin/env python3
"""Model-handle library with and without output staging over batch sizes (picks the AUTO threshold)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, atlas_like, iiwa_like, quadruped_like
from rbdreference_amd.generic import load_generic_library
from tools.time_generic import t
lib = load_generic_library()
for mk, dt, Bs in ((iiwa_like, torch.float32, (16384, 65536, 131072, 262144, 1 << 20)), (atlas_like, torch.float32, (16384, 65536, 262144)),
                   (quadruped_like, torch.float64, (65536, 262144))):
    robot = mk(); gen = RBDReference(robot, build=False, generic="only"); n = gen.n
    for B in Bs:
        rng = np.random.default_rng(3)
        q, qd, qdd = (torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-3, 3, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n))))
        dc = torch.empty((B, n, 2 * n), dtype=dt, device="cuda"); M = torch.empty((B, n, n), dtype=dt, device="cuda")
        c = torch.empty((B, n), dtype=dt, device="cuda"); v = torch.empty((B, 6, n), dtype=dt, device="cuda"); a = torch.empty_like(v); f = torch.empty_like(v)
        out = []
        for nm, fn in (("rnea", lambda: gen.rnea(q, qd, qdd, out=(c, v, a, f))), ("rnea_grad", lambda: gen.rnea_grad(q, qd, qdd, out=dc)), ("minv", lambda: gen.minv(q, out=M))):
            ts = []
            for mode in (1, 2):
                lib.rbd_g_set_output_staging(mode); ts.append(t(fn, 3))
            out.append(f"{nm} direct {ts[0]:9.1f} staged {ts[1]:9.1f} us")
        lib.rbd_g_set_output_staging(0)
        print(f"{robot.name:15s} B={B:8d} waves={B // 64:6d} {str(dt)[6:]:8s} " + " | ".join(out))
