#!/usr/bin/env python3
"""minv of the small-tree test robots under every phase-A option (rbd_set_option), C-ABI launches with pre-allocated
buffers: which path a dual-arm / hexapod-shaped robot should take ('topology lottery')."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_robot
from rbdreference_amd import RBDReference
from rbdreference_amd import _lib as L
from tools.time_generic import t

for name in sys.argv[1:] or ("random_forest_n8", "random_limbs_n14", "random_twochains_n18", "random_tree_n9"):
    rbd = RBDReference(make_robot(name), build=False, generic="never"); n = rbd.n
    for dt in (torch.float32, torch.float64):
      for B in [int(x) for x in os.environ.get("BS", "65536").split(",")]:
          esz = 4 if dt == torch.float32 else 8
          q = torch.tensor(np.random.default_rng(1).uniform(-3, 3, (B, n)), dtype=dt, device="cuda")
          M = torch.empty((B, n, n), dtype=dt, device="cuda")
          for mode, mv in (("AUTO", L.RBD_MINV_PHASE_A_AUTO), ("LANE", L.RBD_MINV_PHASE_A_LANE), ("IA8", L.RBD_MINV_PHASE_A_IA8), ("FUSED", L.RBD_MINV_PHASE_A_FUSED)):
              rbd._lib.set_option(L.RBD_OPT_MINV_PHASE_A, mv)
              wsb = rbd.minv_workspace_bytes(B, dt)
              ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device="cuda")
              us = t(lambda: rbd.minv(q, out=M, workspace=ws), 20)
              print(f"{name:22s} {str(dt)[6:]:8s} {mode:6s} minv {us:9.1f} us  {B * (n + n * n) * esz / us / 1e3:8.1f} GB/s  {rbd._lib.kernel_name(L.RBD_OP_MINV, esz, B)}")
          rbd._lib.set_option(L.RBD_OPT_MINV_PHASE_A, L.RBD_MINV_PHASE_A_AUTO)
