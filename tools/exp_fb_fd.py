#!/usr/bin/env python3
"""Interleaved A/B of rbd_forward_dynamics(_grad)_f32 of the floating quadruped over tagged library variants
(python tools/exp_fb_fd.py build tag=-DFLAG ... ; python tools/exp_fb_fd.py run [B] on the GPU box)."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from rbdreference_amd import floating_quadruped_like, pack_robot
from rbdreference_amd.build import build_model, lib_path
m = pack_robot(floating_quadruped_like())
if sys.argv[1] == "build":
    for spec in sys.argv[2:]:
        tag, fl = spec.split("=", 1)
        print(build_model(m, extra_flags=[f for f in fl.split(",") if f], tag=tag))
    sys.exit(0)
import numpy as np, torch
from rbdreference_amd._lib import _declare
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
nv = m.n + 5
libs = {"base": lib_path(m)}
for p in sorted(glob.glob(lib_path(m)[:-3] + ".*.so")):
    tag = p.split(".")[-2]
    if tag != "asan" and "_f" not in tag: libs[tag] = p
L = {}
for t, p in libs.items():
    L[t] = ctypes.CDLL(p); _declare(L[t])
rng = np.random.default_rng(0)
q, qd, u = (torch.tensor(rng.uniform(-1, 1, (B, nv)), dtype=torch.float32, device="cuda") for _ in range(3))
out = torch.empty((B, nv, 2 * nv), device="cuda"); qdd = torch.empty((B, nv), device="cuda")
wsb = max(x.rbd_fd_workspace_bytes(B, 4) for x in L.values()); ws = torch.empty((wsb,), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
calls = {"forward_dynamics_grad": lambda x: x.rbd_forward_dynamics_grad_f32(q.data_ptr(), qd.data_ptr(), u.data_ptr(), ctypes.c_float(-9.81), B, qdd.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, st),
         "forward_dynamics": lambda x: x.rbd_forward_dynamics_f32(q.data_ptr(), qd.data_ptr(), u.data_ptr(), ctypes.c_float(-9.81), B, qdd.data_ptr(), ws.data_ptr(), wsb, st)}
for name, call in calls.items():
    ref = None; res = {t: [] for t in L}
    for rnd in range(10):
        for t, x in L.items():
            for _ in range(3): assert call(x) == 0, x.rbd_last_error()
            torch.cuda.synchronize()
            if rnd == 0:
                o = (out if "grad" in name else qdd).clone()
                if ref is None: ref = o
                print(f"  {name} {t}: max |diff| vs base {(o - ref).abs().max().item():.2e} (max |ref| {ref.abs().max().item():.2e})")
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call(x)
            e1.record(); torch.cuda.synchronize()
            if rnd >= 2: res[t].append(e0.elapsed_time(e1) / 20 * 1e3)
    for t in L: print(f"{name:24s} {t:12s} min {min(res[t]):8.2f} us  med {sorted(res[t])[len(res[t]) // 2]:8.2f} us")
