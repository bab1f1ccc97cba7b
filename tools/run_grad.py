#!/usr/bin/env python3
"""Launch rbd_rnea_grad_f32 of ONE library a few times (the command profiled with rocprofv3 --pmc).

    LIB=<path to .so or tag> ROBOT=iiwa_like B=1048576 REPS=6 python3 tools/run_grad.py
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import builtin_robot, pack_robot
from rbdreference_amd.build import lib_path
m = pack_robot(builtin_robot(os.environ.get("ROBOT", "iiwa_like")))
B = int(os.environ.get("B", str(1 << 20))); REPS = int(os.environ.get("REPS", "6"))
lib = os.environ.get("LIB", "")
path = lib if lib.endswith(".so") else (lib_path(m)[:-3] + (f".{lib}.so" if lib else ".so"))
L = ctypes.CDLL(path)
f = L.rbd_rnea_grad_f32
f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 3
rng = np.random.default_rng(0); n = m.n
q, qd, qdd = (torch.tensor(x, dtype=torch.float32, device="cuda") for x in
              (rng.uniform(-np.pi, np.pi, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n))))
c = torch.empty((B, n), dtype=torch.float32, device="cuda"); dc = torch.empty((B, n, 2 * n), dtype=torch.float32, device="cuda")
for _ in range(REPS):
    rc = f(q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), -9.81, 0, B, c.data_ptr(), dc.data_ptr(), None)
    assert rc == 0, rc
torch.cuda.synchronize()
print(path, float(dc.double().abs().sum()))
