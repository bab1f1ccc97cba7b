#!/usr/bin/env python3
"""One-off: build-on-first-use of NEW robots on the GPU box (hipcc there) + parity vs the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import rbd_oracle as orc
from rbdreference_amd import RBDReference, random_tree
for parent, seed, pris in (([-1, 0, 1, 2, 3], 31, 0), ([-1, 0, 0, 1, -1, 4], 32, 0), ([-1, 0, 1], 33, 2)):
    robot = random_tree(parent, seed=seed, prismatic_every=pris)
    t = time.time(); rbd = RBDReference(robot); dt_build = time.time() - t
    om = orc.model_from_robot(robot); n = om.n
    rng = np.random.default_rng(seed)
    q = rng.uniform(-3, 3, (100, n)); qd = rng.uniform(-1, 1, (100, n)); qdd = rng.uniform(-1, 1, (100, n))
    tq, tqd, tqdd = (torch.tensor(x, device="cuda") for x in (q, qd, qdd))
    def err(x, r):
        x = x.cpu().numpy().reshape(100, -1); r = r.reshape(100, -1)
        return float(np.max(np.max(np.abs(x - r), 1) / np.maximum(np.max(np.abs(r), 1), 1e-300)))
    c, dc = rbd.rnea_grad(tq, tqd, tqdd, return_c=True)
    cr, dcr = orc.rnea_grad(om, q, qd, qdd, return_c=True)
    e = dict(dc=err(dc, dcr), c=err(c, cr), minv=err(rbd.minv(tq), orc.minv(om, q)), H=err(rbd.crba(tq), orc.crba(om, q)),
             fd=err(rbd.forward_dynamics(tq, tqd, tqdd), orc.forward_dynamics(om, q, qd, qdd)))
    print(f"{robot.name}: built in {dt_build:.0f} s; fp64 errors {e}")
    assert all(v < 1e-9 for v in e.values()), e
print("JIT check OK")
