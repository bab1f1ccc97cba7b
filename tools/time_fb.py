#!/usr/bin/env python3
"""HIP-event timings of the floating-base entry points through the C-ABI (pre-allocated buffers), with the
algorithmic HBM rate next to each.

    B=65536 DTYPE=f32 python tools/time_fb.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, floating_quadruped_like
from rbdreference_amd._lib import RBD_OPT_GRAD_KERNEL, RBD_GRAD_KERNEL_AUTO, RBD_GRAD_KERNEL_COLS

B = int(os.environ.get("B", "65536"))
dt = torch.float64 if os.environ.get("DTYPE", "f32") == "f64" else torch.float32
esz = 4 if dt == torch.float32 else 8
rbd = RBDReference(floating_quadruped_like(), build=False)
nv, nb = rbd.nv, rbd.n
rng = np.random.default_rng(5)
q, qd, qdd = (torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-1, 1, (B, nv)), rng.uniform(-1, 1, (B, nv)), rng.uniform(-1, 1, (B, nv))))
c = torch.empty((B, nv), dtype=dt, device="cuda")
v = torch.empty((B, 6, nb), dtype=dt, device="cuda"); a = torch.empty_like(v); f = torch.empty_like(v)
dc = torch.empty((B, nv, 2 * nv), dtype=dt, device="cuda")
M = torch.empty((B, nv, nv), dtype=dt, device="cuda")
st = torch.cuda.current_stream().cuda_stream
frnea = rbd._fn("rbd_rnea", dt); fgrad = rbd._fn("rbd_rnea_grad", dt); fminv = rbd._fn("rbd_minv", dt)


def t(fn, iters=100):
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


def line(what, us, nbytes):
    gbs = B * nbytes / us / 1e3
    print(f"fb_quadruped {what:34s} B={B} {os.environ.get('DTYPE', 'f32')}: {us:8.2f} us  {gbs:8.1f} GB/s  {gbs / 80:5.1f} % of 8 TB/s")


us = t(lambda: frnea(q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), -9.81, B, c.data_ptr(), v.data_ptr(), a.data_ptr(), f.data_ptr(), st))
line("rnea (c,v,a,f)", us, (4 * nv + 18 * nb) * esz)
us = t(lambda: frnea(q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), -9.81, B, c.data_ptr(), None, None, None, st))
line("rnea (c only)", us, 4 * nv * esz)
for nm, opt in (("auto", RBD_GRAD_KERNEL_AUTO), ("cols", RBD_GRAD_KERNEL_COLS)):
    rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, opt)
    us = t(lambda: fgrad(q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), -9.81, 0, B, c.data_ptr(), dc.data_ptr(), st), 50)
    line(f"rnea_grad (c, dc_du) [{nm}] {rbd._lib.kernel_name(1, esz, B)[:24]}", us, (4 * nv + 2 * nv * nv) * esz)
rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, RBD_GRAD_KERNEL_AUTO)
us = t(lambda: fminv(q.data_ptr(), B, 1, M.data_ptr(), None, 0, st), 50)
line("minv dense", us, (nv + nv * nv) * esz)
u = torch.tensor(rng.uniform(-1, 1, (B, nv)), dtype=dt, device="cuda")
for nm, fn in (("forward_dynamics", lambda: rbd.forward_dynamics(q, qd, u)),):
    try:
        us = t(fn, 20)
        line(nm + " (API)", us, 4 * nv * esz)
    except Exception as e:   # noqa
        print(nm, "failed:", e)
try:
    us = t(lambda: rbd.forward_dynamics_grad(q, qd, u), 20)
    line("forward_dynamics_grad (API)", us, (3 * nv + 2 * nv * nv) * esz)
except Exception as e:   # noqa
    print("forward_dynamics_grad:", str(e)[:100])
