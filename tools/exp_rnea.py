#!/usr/bin/env python3
"""Time rbd_rnea_f32 (c, v, a, f) of every tagged library variant of ROBOT (see tools/exp_grad.py build)."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import builtin_robot, pack_robot
from rbdreference_amd.build import lib_path
m = pack_robot(builtin_robot(os.environ.get("ROBOT", "atlas_like")))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
base = lib_path(m)
libs = [("base", base)] + sorted((os.path.basename(p).split(".")[-2], p) for p in glob.glob(base[:-3] + ".*.so"))
n = m.n
rng = np.random.default_rng(0)
q, qd, qdd = (torch.tensor(x, dtype=torch.float32, device="cuda") for x in (rng.uniform(-np.pi, np.pi, (B, n)), rng.uniform(-1, 1, (B, n)), rng.uniform(-1, 1, (B, n))))
c = torch.empty((B, n), dtype=torch.float32, device="cuda"); v = torch.empty((B, 6, n), dtype=torch.float32, device="cuda"); a = torch.empty_like(v); f = torch.empty_like(v)
for tag, p in libs:
    L = ctypes.CDLL(p)
    fn = L.rbd_rnea_f32
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_int64] + [ctypes.c_void_p] * 5
    call = lambda: fn(q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), -9.81, B, c.data_ptr(), v.data_ptr(), a.data_ptr(), f.data_ptr(), None)
    ts = []
    for rnd in range(5):
        for _ in range(10): call()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): call()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 100 * 1e3)
    print(f"{tag:20s} rnea min {min(ts):7.2f} us  med {sorted(ts)[2]:7.2f} us   checksum {float(f.double().abs().sum()):.6e}")
