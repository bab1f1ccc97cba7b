#!/usr/bin/env python3
"""Experiment harness: build tagged variants of the iiwa library (extra hipcc flags) here, then
time rbd_rnea_grad_f32 of every variant found on the GPU box in ONE process (interleaved rounds).

    python tools/exp_grad.py build  tag1=-DFOO,-DBAR tag2=-fno-slp-vectorize ...
    python tools/exp_grad.py run    [B]
"""
import ctypes, glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from rbdreference_amd import builtin_robot, pack_robot
    from rbdreference_amd.build import build_model, lib_path
    m = pack_robot(builtin_robot(os.environ.get("ROBOT", "iiwa_like")))
    F64 = os.environ.get("DTYPE", "f32") == "f64"
    if sys.argv[1] == "build":
        from concurrent.futures import ThreadPoolExecutor
        specs = [a.split("=", 1) for a in sys.argv[2:]]
        with ThreadPoolExecutor(4) as ex:
            for p in ex.map(lambda s: build_model(m, force=True, extra_flags=[f for f in s[1].split(",") if f], tag=s[0]), specs):
                print(p)
        return
    import numpy as np, torch
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
    base = lib_path(m)
    libs = [("base", base)] + sorted((os.path.basename(p).split(".")[-2], p) for p in glob.glob(base[:-3] + ".*.so"))
    rng = np.random.default_rng(0)
    n = m.n
    tdt = torch.float64 if F64 else torch.float32
    q = torch.tensor(rng.uniform(-np.pi, np.pi, (B, n)), dtype=tdt, device="cuda")
    qd = torch.tensor(rng.uniform(-1, 1, (B, n)), dtype=tdt, device="cuda")
    qdd = torch.tensor(rng.uniform(-1, 1, (B, n)), dtype=tdt, device="cuda")
    c = torch.empty((B, n), dtype=tdt, device="cuda"); dc = torch.empty((B, n, 2 * n), dtype=tdt, device="cuda")
    fns = []
    for tag, p in libs:
        L = ctypes.CDLL(p)
        f = L.rbd_rnea_grad_f64 if F64 else L.rbd_rnea_grad_f32
        f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_double if F64 else ctypes.c_float, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 3
        fns.append((tag, f))
    st = torch.cuda.current_stream().cuda_stream
    res = {t: [] for t, _ in fns}
    ref = None
    ITERS = int(os.environ.get("EXP_ITERS", "10"))
    for rnd in range(int(os.environ.get("EXP_ROUNDS", "6"))):
        for tag, f in fns:
            for _ in range(3):
                f(q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), -9.81, 0, B, c.data_ptr(), dc.data_ptr(), st)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITERS):
                f(q.data_ptr(), qd.data_ptr(), qdd.data_ptr(), -9.81, 0, B, c.data_ptr(), dc.data_ptr(), st)
            e1.record(); torch.cuda.synchronize()
            res[tag].append(e0.elapsed_time(e1) / ITERS)
            if rnd == 0:
                if tag == "base":
                    ref = dc.clone()
                else:
                    d = ((dc - ref).abs().amax() / ref.abs().amax()).item()
                    print(f"  {tag}: max diff vs base {d:.2e}")
    for tag, v in res.items():
        v = sorted(v)
        q1 = sum(v[:max(1, len(v) // 4)]) / max(1, len(v) // 4)       # mean of the fastest quarter of the rounds
        print(f"{tag:24s} min {v[0]*1e3:8.1f} us  fastest-quarter mean {q1*1e3:8.1f} us  med {v[len(v)//2]*1e3:8.1f} us   {B/(v[0]*1e-3)/1e9:6.3f} G evals/s")


if __name__ == "__main__":
    main()
