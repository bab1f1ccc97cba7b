#!/usr/bin/env python3
"""Run ONE op of ONE robot a few times (for rocprofv3 --pmc / --kernel-trace):  python tools/run_op.py <robot> <f32|f64> <rnea|rnea_grad|minv|fd_grad> <B> [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, builtin_robot
name, prec, op, B = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
dt = torch.float64 if prec == "f64" else torch.float32
r = RBDReference(builtin_robot(name), build=False, generic="never")
rng = np.random.default_rng(3)
q, qd, qdd = [torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-np.pi, np.pi, (B, r.nv)), rng.uniform(-1, 1, (B, r.nv)), rng.uniform(-1, 1, (B, r.nv)))]
for _ in range(reps):
    if op == "rnea": r.rnea(q, qd, qdd)
    elif op == "rnea_grad": r.rnea_grad(q, qd, qdd, return_c=True)
    elif op == "minv": r.minv(q)
    else: r.forward_dynamics_grad(q, qd, qdd)
torch.cuda.synchronize()
