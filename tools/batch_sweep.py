#!/usr/bin/env python3
"""The headline path over batch sizes: `rnea_grad` (c and dc_du in one launch) of the 7-DoF arm in fp32 through the C-ABI with
pre-allocated buffers, B = 2^10 .. 2^22; eager launches and the same launches replayed from a HIP graph.  Columns: us per launch,
G evals/s, algorithmic GB/s (504 B per evaluation), fraction of 8 TB/s, kernel the library picked.

    python tools/batch_sweep.py [iiwa_like|atlas_like|quadruped_like] [f32|f64]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, builtin_robot
from rbdreference_amd import _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "iiwa_like"
dt = torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32
esz = 8 if dt == torch.float64 else 4
r = RBDReference(builtin_robot(name), build=False, generic="never")
n = r.n
bytes_per_eval = (4 * n + 2 * n * n) * esz
rng = np.random.default_rng(3)
lo, hi = (10, 22) if n <= 12 else (8, 19)
print(f"# {name} n = {n} {str(dt)[6:]}  rnea_grad (c, dc_du): {bytes_per_eval} algorithmic bytes per evaluation")
for lg in range(lo, hi + 1):
    B = 1 << lg
    q, qd, qdd = [torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-np.pi, np.pi, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n)))]
    c = torch.empty((B, n), dtype=dt, device="cuda"); d = torch.empty((B, n, 2 * n), dtype=dt, device="cuda")
    f = lambda: r.rnea_grad(q, qd, qdd, return_c=True, out=(c, d))
    for _ in range(5): f()
    torch.cuda.synchronize()
    reps = max(20, min(400, int(2e-2 / (max(B, 4096) * 1.3e-10))))
    def timed(fn):
        best = []
        for _ in range(5):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record(); torch.cuda.synchronize()
            best.append(a.elapsed_time(b) / reps * 1e3)
        return sorted(best)[2]
    def eager():
        for _ in range(reps): f()
    us_e = timed(eager)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): f()
    torch.cuda.synchronize()
    us_g = timed(lambda: g.replay())
    gb = lambda us: B * bytes_per_eval / us / 1e3
    print(f"B = 2^{lg:<2d} = {B:8d}   eager {us_e:9.2f} us  {B / us_e / 1e3:7.3f} G evals/s  {gb(us_e):7.1f} GB/s  frac {gb(us_e) / 8000:5.3f}   "
          f"graph {us_g:9.2f} us  {B / us_g / 1e3:7.3f} G evals/s  frac {gb(us_g) / 8000:5.3f}   {r._lib.kernel_name(L.RBD_OP_RNEA_GRAD, esz, B)}")
