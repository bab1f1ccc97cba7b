#!/usr/bin/env python3
"""rbd_rnea_grad under every kernel option of a robot (C-ABI, pre-allocated buffers, HIP events).

    ROBOT=quadruped_like B=65536 DTYPE=f64 python tools/time_grad_options.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_robot
from rbdreference_amd import RBDReference
from rbdreference_amd._lib import (RBD_GRAD_KERNEL_AUTO, RBD_GRAD_KERNEL_BATCH, RBD_GRAD_KERNEL_COLS, RBD_GRAD_KERNEL_TREE,
                                   RBD_OP_RNEA_GRAD, RBD_OPT_GRAD_KERNEL)
from tools.time_generic import t

for spec in (os.environ.get("CASES") or "quadruped_like:65536:f64,quadruped_like:65536:f32,iiwa_like:1048576:f64,random_chain_n7:1048576:f64,random_tree_n9:65536:f64").split(","):
    name, B, dn = spec.split(":"); B = int(B)
    dt = torch.float64 if dn == "f64" else torch.float32; esz = 8 if dn == "f64" else 4
    rbd = RBDReference(make_robot(name), build=False, generic="never"); n = rbd.n
    rng = np.random.default_rng(2)
    q, qd, qdd = (torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-np.pi, np.pi, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n))))
    c = torch.empty((B, n), dtype=dt, device="cuda"); dc = torch.empty((B, n, 2 * n), dtype=dt, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    f = rbd._fn("rbd_rnea_grad", dt)
    ref = None
    for nm, opt in (("auto", RBD_GRAD_KERNEL_AUTO), ("batch", RBD_GRAD_KERNEL_BATCH), ("tree", RBD_GRAD_KERNEL_TREE), ("cols", RBD_GRAD_KERNEL_COLS)):
        rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, opt)
        kn = rbd._lib.kernel_name(RBD_OP_RNEA_GRAD, esz, B)
        us = t(lambda: f(q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), -9.81, 0, B, c.data_ptr(), dc.data_ptr(), st), 20)
        if ref is None: ref = dc.clone()
        d = ((dc - ref).abs().amax() / ref.abs().amax()).item()
        gb = B * (4 * n + 2 * n * n) * esz / us / 1e3
        print(f"{name:18s} B={B:8d} {dn} {nm:6s} {kn:44s} {us:9.1f} us  {gb:7.1f} GB/s ({gb / 80:4.1f} %)  diff vs auto {d:.1e}")
    rbd._lib.set_option(RBD_OPT_GRAD_KERNEL, RBD_GRAD_KERNEL_AUTO)
