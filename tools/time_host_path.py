#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer path: numpy (fp64) in -> upload -> HIP kernels -> numpy out.
Reported in DESIGN.md beside the HBM-resident headline; never the bench `value`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, iiwa_like
r = RBDReference(iiwa_like(), build=False)
for B in (4096, 1 << 17, 1 << 20):
    rng = np.random.default_rng(0)
    q, qd, qdd = rng.uniform(-3, 3, (B, 7)), rng.uniform(-1, 1, (B, 7)), rng.uniform(-1, 1, (B, 7))
    for _ in range(2): r.rnea_grad(q, qd, qdd)
    n = 5
    t0 = time.perf_counter()
    for _ in range(n): out = r.rnea_grad(q, qd, qdd)
    dt = (time.perf_counter() - t0) / n
    nbytes = 3 * q.nbytes + out.nbytes
    print(f"B={B:8d}  {dt*1e3:9.3f} ms per call  {B/dt/1e6:8.2f} M evals/s  ({nbytes/dt/1e9:5.1f} GB/s over PCIe incl. host copies, fp64)")
