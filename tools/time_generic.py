#!/usr/bin/env python3
"""HIP-event timings of the model-handle library (include/rbd_generic.h) next to the robot's own library, through
the Python API with pre-allocated outputs where the API offers them.

    python tools/time_generic.py            # iiwa B = 1 048 576, Atlas B = 16 384, quadruped fp64 B = 65 536
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, atlas_like, iiwa_like, quadruped_like


def t(fn, iters=10):
    best = 1e9
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


if __name__ == "__main__":
    for mk, B, dt in ((iiwa_like, 1 << 20, torch.float32), (atlas_like, 16384, torch.float32), (quadruped_like, 65536, torch.float64)):
        robot = mk()
        spec = RBDReference(robot, build=False, generic="never"); gen = RBDReference(robot, build=False, generic="only")
        n = spec.n; rng = np.random.default_rng(3)
        q, qd, qdd = (torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-3, 3, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n))))
        dc = torch.empty((B, n, 2 * n), dtype=dt, device="cuda"); M = torch.empty((B, n, n), dtype=dt, device="cuda")
        c = torch.empty((B, n), dtype=dt, device="cuda"); v = torch.empty((B, 6, n), dtype=dt, device="cuda"); a = torch.empty_like(v); f = torch.empty_like(v)
        for nm, fn in (("rnea (c,v,a,f)", lambda r: r.rnea(q, qd, qdd, out=(c, v, a, f))),
                       ("rnea_grad", lambda r: r.rnea_grad(q, qd, qdd, out=dc)),
                       ("minv", lambda r: r.minv(q, out=M)),
                       ("forward_dynamics_grad", lambda r: r.forward_dynamics_grad(q, qd, qdd))):
            ts, tg = t(lambda: fn(spec)), t(lambda: fn(gen), 5)
            print(f"{robot.name:16s} n={n:2d} B={B:8d} {str(dt)[6:]:8s} {nm:24s} specialised {ts:9.1f} us   model-handle {tg:10.1f} us   x{tg / ts:5.1f}")
