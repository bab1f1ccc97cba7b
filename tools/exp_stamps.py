#!/usr/bin/env python3
"""Diagnostic: phase durations (s_memtime ticks) of rnea_grad_kernel from a -DRBD_EXP_STAMPS build."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import iiwa_like, pack_robot
from rbdreference_amd.build import lib_path
m = pack_robot(iiwa_like()); p = lib_path(m)[:-3] + ".stamps.so"
L = ctypes.CDLL(p); f = L.rbd_rnea_grad_f32
f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_float, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 3
B = 1 << 20
rng = np.random.default_rng(0)
q = torch.tensor(rng.uniform(-np.pi, np.pi, (B, 7)), dtype=torch.float32, device="cuda")
qd = torch.tensor(rng.uniform(-1, 1, (B, 7)), dtype=torch.float32, device="cuda"); qdd = qd.clone()
c = torch.empty((B, 7), dtype=torch.float32, device="cuda"); dc = torch.empty((B, 7, 14), dtype=torch.float32, device="cuda")
for _ in range(3):
    f(q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), -9.81, 0, B, c.data_ptr(), dc.data_ptr(), None)
torch.cuda.synchronize()
s = c[::64].cpu().numpy()
names = ["loads", "sincos", "forward", "backward", "flush (+store drain)"]
tot = s[:, :5].sum(1)
print("per-wave ticks (s_memtime, 100 MHz realtime? or shader clock): median / mean")
for k, nm in enumerate(names):
    print(f"  {nm:22s} {np.median(s[:,k]):10.0f} {s[:,k].mean():10.0f}   {100*s[:,k].mean()/tot.mean():5.1f} %")
print(f"  total                  {np.median(tot):10.0f} {tot.mean():10.0f}")
