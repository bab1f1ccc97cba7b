#!/usr/bin/env python3
"""Measured per-tensor error of minv / minv_bpass / minv_fpass.

For every robot (default: all golden robots) and dtype: worst-row normwise relative error
max|x - ref| / max|ref| against (i) the golden vectors of the real reference and (ii) the fp64 numpy
oracle on `ROWS` random rows.  PHASE_A=lane|ia8 pins phase A of the two-phase robots
(rbd_set_option).  Prints one JSON object.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from conftest import all_golden_names, load_golden, make_robot, rel_err_rows   # noqa: E402
from oracle import rbd_oracle as orc                                           # noqa: E402  (checker only)
from rbdreference_amd import RBDReference                                      # noqa: E402

ROWS = int(os.environ.get("ROWS", "1000"))


def err(x, ref):
    return float(rel_err_rows(x.double().cpu().numpy(), ref))


def main():
    names = sys.argv[1:] or all_golden_names()
    out = {}
    for name in names:
        robot = make_robot(name)
        rbd = RBDReference(robot, build=False)
        rbd._lib.set_option(1, {"": 0, "lane": 1, "ia8": 2}[os.environ.get("PHASE_A", "")])
        om = orc.model_from_robot(robot)
        g = load_golden(name)
        rng = np.random.default_rng(77)
        qr = rng.uniform(-np.pi, np.pi, (ROWS, rbd.n))
        Mi_ref = orc.minv(om, qr)
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            q = torch.tensor(g["q"], device="cuda", dtype=dt)
            r = {}
            r["golden_minv_dense"] = err(rbd.minv(q), g["Minv_dense"])
            r["golden_minv_upper"] = err(rbd.minv(q, output_dense=False), np.triu(g["Minv_upper"]))
            Mb, F, U, D = rbd.minv_bpass(q)
            r["golden_bpass_Minv"] = err(Mb, g["mb_Minv"])
            r["golden_bpass_F"] = err(F, g["mb_F"])
            r["golden_bpass_U"] = err(U, g["mb_U"])
            r["golden_bpass_D"] = err(D, g["mb_Dinv"])
            Mf = rbd.minv_fpass(q, Mb.clone(), F.clone(), U, D)
            r["golden_fpass_Minv_upper"] = err(torch.triu(Mf), np.triu(g["Minv_upper"]))
            r["random_minv_dense"] = err(rbd.minv(torch.tensor(qr, device="cuda", dtype=dt)), Mi_ref)
            out[f"{name}_{tag}"] = r
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
