#!/usr/bin/env python3
"""Time rbd_minv_{f32,f64} (DTYPE=f32 | f64) of every tagged library variant of ROBOT (UNIT=MINV tools/exp_tws.py f32 tag=-DFLAG)."""
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
F64 = os.environ.get("DTYPE", "f32") == "f64"
dt = torch.float64 if F64 else torch.float32
q = torch.tensor(np.random.default_rng(0).uniform(-np.pi, np.pi, (B, n)), dtype=dt, device="cuda")
M = torch.empty((B, n, n), dtype=dt, device="cuda")
ref = None
for tag, p in libs:
    L = ctypes.CDLL(p)
    L.rbd_minv_workspace_bytes.restype = ctypes.c_size_t; L.rbd_minv_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
    wsb = L.rbd_minv_workspace_bytes(B, 8 if F64 else 4); ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device="cuda")
    f = L.rbd_minv_f64 if F64 else L.rbd_minv_f32
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
    if ref is None: ref = M.clone()
    d = ((M - ref).abs().amax() / ref.abs().amax()).item()
    print(f"{tag:20s} diff vs base {d:.1e}  minv min {min(ts):7.2f} us  med {sorted(ts)[2]:7.2f} us")
