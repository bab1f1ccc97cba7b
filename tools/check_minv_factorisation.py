#!/usr/bin/env python3
"""Minv = Psi^T D^-1 Psi from the BACKWARD pass alone, checked against the reference's own outputs in tests/golden/*.npz:

    Minv[i, j] = sum over k in anc(i) & anc(j) of  D_k m[k, i] m[k, j],     m = minv_bpass's Minv (m[k, k] = 1 / D_k), D = its Dinv

(the operator factorisation the reference's minv_bpass + minv_fpass evaluate recursively, RBDReference.py:630-783).  The
one-lane kernels (csrc/rbd_fd_chain.h, csrc/rbd_minv_lane.h) use it instead of the forward pass: ~n^3 / 6 scalar FMAs instead
of n (n + 1) / 2 six-vector transforms."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import all_golden_names, load_golden

worst = 0.0
for name in all_golden_names():
    g = load_golden(name)
    parent = g["parent"]; n = len(parent)
    anc = []
    for i in range(n):
        s, j = set(), i
        while j != -1:
            s.add(j); j = parent[j]
        anc.append(s)
    err = 0.0
    for s in range(g["q"].shape[0]):
        m, D, Md = g["mb_Minv"][s], g["mb_Dinv"][s], g["Minv_dense"][s]
        R = np.array([[sum(D[k] * m[k, i] * m[k, j] for k in anc[i] & anc[j]) for j in range(n)] for i in range(n)])
        err = max(err, np.abs(R - Md).max() / np.abs(Md).max())
    worst = max(worst, err)
    print(f"{name:24s} n={n:3d}  max |factorisation - reference Minv| / max |Minv| = {err:.2e}")
assert worst < 1e-13
print("OK")
