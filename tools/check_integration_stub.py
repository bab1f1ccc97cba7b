#!/usr/bin/env python3
"""Execute the `_HipGeneric` ctypes stub of INTEGRATION.md §2b as written there (GPU box) and compare its
rbd_g_rnea_grad_f64 with the oracle: the documentation is what a maintainer would paste, so it has to run."""
import ctypes, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle import rbd_oracle as orc          # checker
from rbdreference_amd import iiwa_like
from rbdreference_amd.build import build_generic
txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
block = next(b for b in re.findall(r"```python\n(.*?)```", txt, flags=re.S) if "class _HipGeneric" in b)
ns = {"ctypes": ctypes, "np": np}
exec(block, ns)
robot = iiwa_like()
h = ns["_HipGeneric"](robot, lib_path=build_generic(), device=0)
B, n = 50, h.n
rng = np.random.default_rng(0)
q, qd, qdd = rng.uniform(-3, 3, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n))
tq, tqd, tqdd = (torch.tensor(x, device="cuda:0") for x in (q, qd, qdd))
dc = torch.empty((B, n, 2 * n), dtype=torch.float64, device="cuda:0")
rc = h.L.rbd_g_rnea_grad_f64(h.h, tq.data_ptr(), tqd.data_ptr(), tqdd.data_ptr(), -9.81, 0, B, None, dc.data_ptr(), None)
torch.cuda.synchronize()
ref = orc.rnea_grad(orc.model_from_robot(robot), q, qd, qdd)
err = np.abs(dc.cpu().numpy() - ref).max() / np.abs(ref).max()
print("rc", rc, "max rel err vs oracle", err)
assert rc == 0 and err < 1e-11
