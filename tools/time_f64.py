#!/usr/bin/env python3
"""fp64 (the reference's own arithmetic: numpy inputs land here) next to fp32, API calls with pre-allocated outputs.

    python tools/time_f64.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, atlas_like, iiwa_like, quadruped_like
from rbdreference_amd._lib import RBD_OP_RNEA_GRAD
from tools.time_generic import t

for mk, B in ((iiwa_like, 1 << 20), (quadruped_like, 65536), (atlas_like, 16384)):
    robot = mk(); rbd = RBDReference(robot, build=False, generic="never"); n = rbd.n
    for dt in (torch.float32, torch.float64):
        esz = 4 if dt == torch.float32 else 8
        rng = np.random.default_rng(3)
        q, qd, qdd = (torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-3, 3, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n))))
        dc = torch.empty((B, n, 2 * n), dtype=dt, device="cuda"); M = torch.empty((B, n, n), dtype=dt, device="cuda")
        c = torch.empty((B, n), dtype=dt, device="cuda"); v = torch.empty((B, 6, n), dtype=dt, device="cuda"); a = torch.empty_like(v); f = torch.empty_like(v)
        for nm, fn, nbytes in (("rnea (c,v,a,f)", lambda: rbd.rnea(q, qd, qdd, out=(c, v, a, f)), 22 * n * esz),
                               ("rnea_grad", lambda: rbd.rnea_grad(q, qd, qdd, out=dc), (3 * n + 2 * n * n) * esz),
                               ("minv", lambda: rbd.minv(q, out=M), (n + n * n) * esz)):
            us = t(fn, 20)
            kn = rbd._lib.kernel_name(RBD_OP_RNEA_GRAD, esz, B) if nm == "rnea_grad" else ""
            print(f"{robot.name:16s} B={B:8d} {str(dt)[6:]:8s} {nm:16s} {us:9.1f} us  {B * nbytes / us / 1e3:8.1f} GB/s ({B * nbytes / us / 8e4:5.1f} % of 8 TB/s) {kn}")
