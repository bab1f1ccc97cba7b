#!/usr/bin/env python3
"""HIP-event timings of direct C-ABI launches (pre-allocated buffers, no per-call allocation), per kernel option.

    ROBOT=atlas_like B=16384 python tools/time_abi.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, builtin_robot
from rbdreference_amd._lib import (RBD_OPT_RNEA_KERNEL, RBD_RNEA_KERNEL_AUTO, RBD_RNEA_KERNEL_BATCH, RBD_RNEA_KERNEL_GROUPS,
                                   RBD_OPT_MINV_PHASE_A, RBD_MINV_PHASE_A_AUTO, RBD_MINV_PHASE_A_LANE, RBD_MINV_PHASE_A_IA8,
                                   RBD_MINV_PHASE_A_FUSED)
name = os.environ.get("ROBOT", "atlas_like"); B = int(os.environ.get("B", "16384"))
dt = torch.float64 if os.environ.get("DTYPE", "f32") == "f64" else torch.float32
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import make_robot
rbd = RBDReference(make_robot(name), build=False); n = rbd.n
rng = np.random.default_rng(2)
q, qd, qdd = (torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-np.pi, np.pi, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n))))
c = torch.empty((B, n), dtype=dt, device="cuda"); v = torch.empty((B, 6, n), dtype=dt, device="cuda"); a = torch.empty_like(v); f = torch.empty_like(v)
M = torch.empty((B, n, n), dtype=dt, device="cuda")
esz = 4 if dt == torch.float32 else 8
wsb = int(rbd._lib.lib.rbd_minv_workspace_bytes(B, esz)); ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
frnea = rbd._fn("rbd_rnea", dt); fminv = rbd._fn("rbd_minv", dt)


def t(fn, iters=200):
    best = 1e9
    for _ in range(4):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for nm, opt in (("auto", RBD_RNEA_KERNEL_AUTO), ("batch", RBD_RNEA_KERNEL_BATCH), ("groups", RBD_RNEA_KERNEL_GROUPS)):
    rbd._lib.set_option(RBD_OPT_RNEA_KERNEL, opt)
    us = t(lambda: frnea(q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), -9.81, B, c.data_ptr(), v.data_ptr(), a.data_ptr(), f.data_ptr(), st))
    print(f"{name} rnea (c,v,a,f) B={B} {nm:7s}: {us:7.2f} us   {B * 22 * n * esz / us / 1e3:7.1f} GB/s")
rbd._lib.set_option(RBD_OPT_RNEA_KERNEL, RBD_RNEA_KERNEL_AUTO)
us = t(lambda: frnea(q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), -9.81, B, c.data_ptr(), None, None, None, st))
print(f"{name} rnea (c only)   B={B}        : {us:7.2f} us")
for nm, opt in (("auto", RBD_MINV_PHASE_A_AUTO), ("lane", RBD_MINV_PHASE_A_LANE), ("ia8", RBD_MINV_PHASE_A_IA8), ("fused", RBD_MINV_PHASE_A_FUSED)):
    rbd._lib.set_option(RBD_OPT_MINV_PHASE_A, opt)
    us = t(lambda: fminv(q.data_ptr(), B, 1, M.data_ptr(), ws.data_ptr(), wsb, st), 100)
    print(f"{name} minv dense      B={B} {nm:7s}: {us:7.2f} us   {B * (n + n * n) * esz / us / 1e3:7.1f} GB/s")
