#!/usr/bin/env python3
"""A/B timing of rbd_forward_dynamics_grad_f32 over tagged variant libraries of a robot (built by tools/exp_tws.py with
UNIT=FD), interleaved in one process, HIP events around each call:

    ROBOT=iiwa_like UNIT=FD KFILTER=fd_pre python tools/exp_tws.py f32 base= nobias=-DRBD_FDP_EXP_NOBIAS ...
    ROBOT=iiwa_like python tools/exp_fd.py [B]            # on the GPU box
"""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from rbdreference_amd import builtin_robot, pack_robot
from rbdreference_amd._lib import _declare
from rbdreference_amd.build import lib_path


def main():
    m = pack_robot(builtin_robot(os.environ.get("ROBOT", "iiwa_like")))
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    n = m.n
    libs = {}
    for p in sorted(glob.glob(lib_path(m)[:-3] + ".*.so")):
        tag = p.split(".")[-2]
        if tag in ("asan",) or "_f32" in tag or "_f64" in tag:
            continue
        L = ctypes.CDLL(p); _declare(L); libs[tag] = L
    rng = np.random.default_rng(0)
    q, qd, u = (torch.tensor(rng.uniform(-1, 1, (B, n)), dtype=torch.float32, device="cuda") for _ in range(3))
    out = torch.empty((B, n, 2 * n), device="cuda"); qdd = torch.empty((B, n), device="cuda")
    wsb = max(L.rbd_fd_workspace_bytes(B, 4) for L in libs.values())
    ws = torch.empty((wsb,), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    call = lambda L: L.rbd_forward_dynamics_grad_f32(q.data_ptr(), qd.data_ptr(), u.data_ptr(), ctypes.c_float(-9.81), B, qdd.data_ptr(), out.data_ptr(), ws.data_ptr(), wsb, st)   # noqa: E731
    for L in libs.values():
        for _ in range(5):
            assert call(L) == 0, L.rbd_last_error()
    torch.cuda.synchronize()
    res = {t: [] for t in libs}
    for rnd in range(int(os.environ.get("ROUNDS", 12))):
        for t, L in libs.items():
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call(L)
            e1.record(); torch.cuda.synchronize()
            res[t].append(e0.elapsed_time(e1) / 10 * 1e3)
    for t, v in res.items():
        v = sorted(v)
        print(f"{t:16s} median {v[len(v)//2]:8.1f} us   min {v[0]:8.1f} us")


if __name__ == "__main__":
    main()
