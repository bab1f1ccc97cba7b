#!/usr/bin/env python3
"""Time rbd_minv_f32 of every tagged Atlas library variant (tools/exp_grad.py build ROBOT=atlas_like tag=-DFLAG)."""
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
q = torch.tensor(np.random.default_rng(0).uniform(-np.pi, np.pi, (B, n)), dtype=torch.float32, device="cuda")
M = torch.empty((B, n, n), dtype=torch.float32, device="cuda")
for tag, p in libs:
    L = ctypes.CDLL(p)
    L.rbd_minv_workspace_bytes.restype = ctypes.c_size_t; L.rbd_minv_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
    wsb = L.rbd_minv_workspace_bytes(B, 4); ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device="cuda")
    f = L.rbd_minv_f32
    f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    ts = []
    for rnd in range(5):
        for _ in range(5): f(q.data_ptr(), B, 1, M.data_ptr(), ws.data_ptr(), wsb, None)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f(q.data_ptr(), B, 1, M.data_ptr(), ws.data_ptr(), wsb, None)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(f"{tag:20s} minv min {min(ts):7.2f} us  med {sorted(ts)[2]:7.2f} us")
