#!/usr/bin/env python3
"""minv of the 30-body robot over batch sizes and precisions (API with a pre-allocated output), plus two small trees.

    python tools/time_minv_sizes.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_robot
from rbdreference_amd import RBDReference
from rbdreference_amd._lib import RBD_OP_MINV
from tools.time_generic import t

for name, sizes in (("atlas_like", (4096, 16384, 131072, 524288)), ("random_tree_n9", (65536,)), ("random_forest_n8", (65536,)), ("random_limbs_n14", (65536,))):
    rbd = RBDReference(make_robot(name), build=False, generic="never"); n = rbd.n
    for dt in (torch.float32, torch.float64):
        for B in sizes:
            q = torch.tensor(np.random.default_rng(1).uniform(-3, 3, (B, n)), dtype=dt, device="cuda")
            M = torch.empty((B, n, n), dtype=dt, device="cuda")
            us = t(lambda: rbd.minv(q, out=M), 20)
            esz = 4 if dt == torch.float32 else 8
            print(f"{name:18s} {str(dt)[6:]:8s} B={B:7d} minv {us:9.1f} us  {B * (n + n * n) * esz / us / 1e3:8.1f} GB/s  {rbd._lib.kernel_name(RBD_OP_MINV, esz, B)}")
            del q, M
