#!/usr/bin/env python3
"""Run the non-headline BASELINE configurations a few times (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import RBDReference, atlas_like, floating_quadruped_like, iiwa_like, quadruped_like
REPS = int(os.environ.get("REPS", "20"))
def inputs(B, n, seed, dt):
    rng = np.random.default_rng(seed)
    return [torch.tensor(x, dtype=dt, device="cuda") for x in (rng.uniform(-np.pi, np.pi, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n)))]
which = sys.argv[1:] or ["atlas", "quad", "iiwa", "fb", "fdg"]
if "atlas" in which:
    r = RBDReference(atlas_like(), build=False); q, qd, qdd = inputs(16384, 30, 2, torch.float32)
    for _ in range(REPS): r.minv(q); r.rnea(q, qd, qdd); r.rnea_grad(q, qd, qdd, return_c=True); r.aba(q, qd, qdd)
if "atlas64" in which or not sys.argv[1:]:   # the 30-body robot in the reference's own precision: workspace tree kernel (DESIGN 3.1 c')
    r = RBDReference(atlas_like(), build=False); q, qd, qdd = inputs(16384, 30, 2, torch.float64)
    for _ in range(REPS): r.rnea_grad(q, qd, qdd, return_c=True)
if "quad" in which:
    r = RBDReference(quadruped_like(), build=False); q, qd, qdd = inputs(65536, 12, 4, torch.float64)
    for _ in range(REPS): r.rnea_grad(q, qd, qdd, return_c=True); r.minv(q)
if "fb" in which:          # floating base: 13 bodies, nv = 18, fp32, B = 65 536
    r = RBDReference(floating_quadruped_like(), build=False); q, qd, qdd = inputs(65536, r.nv, 5, torch.float32)
    for _ in range(REPS): r.rnea(q, qd, qdd); r.minv(q); r.rnea_grad(q, qd, qdd, return_c=True)
    for _ in range(max(2, REPS // 4)): r.forward_dynamics_grad(q, qd, qdd)
if "iiwa4k" in which:     # BASELINE configs[1] alone: rnea + rnea_grad in one call at B = 4096
    r = RBDReference(iiwa_like(), build=False); q, qd, qdd = inputs(4096, 7, 1, torch.float32)
    for _ in range(REPS): r.rnea_and_grad(q, qd, qdd)
if "iiwa" in which:
    r = RBDReference(iiwa_like(), build=False); q, qd, qdd = inputs(4096, 7, 1, torch.float32)
    for _ in range(REPS): r.rnea(q, qd, qdd); r.rnea_grad(q, qd, qdd, return_c=True); r.minv(q); r.rnea_and_grad(q, qd, qdd)
    q, qd, qdd = inputs(1 << 20, 7, 3, torch.float32)
    for _ in range(REPS): r.rnea(q, qd, qdd); r.minv(q); r.aba(q, qd, qdd)
    for _ in range(max(2, REPS // 4)): r.forward_dynamics_grad(q, qd, qdd)
if "fdg" in which:        # forward_dynamics_grad of the three fixed-base BASELINE robots at their batch sizes
    r = RBDReference(iiwa_like(), build=False); q, qd, qdd = inputs(1 << 20, 7, 3, torch.float32)
    for _ in range(max(3, REPS // 2)): r.forward_dynamics_grad(q, qd, qdd)
    r = RBDReference(atlas_like(), build=False); q, qd, qdd = inputs(16384, 30, 2, torch.float32)
    for _ in range(max(3, REPS // 2)): r.forward_dynamics_grad(q, qd, qdd)
    r = RBDReference(quadruped_like(), build=False); q, qd, qdd = inputs(65536, 12, 4, torch.float64)
    for _ in range(max(3, REPS // 2)): r.forward_dynamics_grad(q, qd, qdd)
torch.cuda.synchronize()
