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
if os.environ.get("VARIANT"):        # one variant, ten launches: for rocprofv3 --pmc
    libs = [x for x in libs if x[0] == os.environ["VARIANT"]]
PMC = bool(os.environ.get("VARIANT"))
fs = []
for tag, p in libs:
    L = ctypes.CDLL(p)
    L.rbd_minv_workspace_bytes.restype = ctypes.c_size_t; L.rbd_minv_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int]
    wsb = L.rbd_minv_workspace_bytes(B, 8 if F64 else 4); ws = torch.empty((max(wsb, 16),), dtype=torch.uint8, device="cuda")
    f = L.rbd_minv_f64 if F64 else L.rbd_minv_f32
    f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    fs.append((tag, f, ws, wsb))
if PMC:
    for tag, f, ws, wsb in fs:
        for _ in range(10): f(q.data_ptr(), B, 1, M.data_ptr(), ws.data_ptr(), wsb, None)
    torch.cuda.synchronize(); sys.exit(0)
# interleaved rounds (the clock of a fresh box settles over the first seconds: timing the variants one after the other
# favours whichever comes last)
ts = {tag: [] for tag, *_ in fs}
diff = {}
for rnd in range(int(os.environ.get("ROUNDS", "12"))):
    for tag, f, ws, wsb in fs:
        if rnd == 0: M.zero_()
        for _ in range(5): f(q.data_ptr(), B, 1, M.data_ptr(), ws.data_ptr(), wsb, None)
        torch.cuda.synchronize()
        if rnd == 0:
            if ref is None: ref = M.clone()
            diff[tag] = ((M - ref).abs().amax() / ref.abs().amax()).item()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f(q.data_ptr(), B, 1, M.data_ptr(), ws.data_ptr(), wsb, None)
        e1.record(); torch.cuda.synchronize()
        if rnd >= 2: ts[tag].append(e0.elapsed_time(e1) / 50 * 1e3)
for tag, *_ in fs:
    v = sorted(ts[tag])
    print(f"{tag:20s} diff vs base {diff[tag]:.1e}  minv min {v[0]:7.2f} us  med {v[len(v) // 2]:7.2f} us")
