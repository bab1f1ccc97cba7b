#!/usr/bin/env python3
"""No-GPU checks of a per-robot library's HOST code, meant to run against the sanitizer build
(`rbdreference_amd.build.build_sanitized`, ASan + UBSan on the host side) with the ASan runtime preloaded:

    LD_PRELOAD=$(python -c 'from rbdreference_amd.build import sanitizer_runtime as s; print(s())') \
    ASAN_OPTIONS=detect_leaks=0 python tools/asan_host_checks.py <lib.asan.so>

Everything here returns before a kernel could launch (or fails at the first HIP call on a machine without a GPU):
argument, alignment and workspace checks of every entry point, the option atomics from several threads, kernel-name
and model-info queries with tight buffers, the error buffer, the workspace pool's release path.  A sanitizer report
aborts the process (non-zero exit); tests/test_host_logic.py runs this as a child process."""
import ctypes
import sys
import threading

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
from rbdreference_amd._lib import EXPORTED_SYMBOLS, RbdModelInfo, _declare  # noqa: E402


def main(path):
    L = ctypes.CDLL(path)
    _declare(L)
    for s in EXPORTED_SYMBOLS:
        getattr(L, s)
    info = RbdModelInfo()
    assert L.rbd_model_info(ctypes.byref(info)) == 0
    assert L.rbd_model_info(None) != 0
    n = info.n
    A = 1 << 20                      # a 16-byte aligned fake device address (never dereferenced on the host)
    M = A + 4                        # misaligned
    checks = 0
    for sfx, g in (("f32", ctypes.c_float(-9.81)), ("f64", ctypes.c_double(-9.81))):
        f = lambda nm: getattr(L, f"{nm}_{sfx}")          # noqa: E731
        # null / inconsistent arguments, negative B, B = 0, misaligned outputs: every one returns before any HIP call
        assert f("rbd_rnea")(None, None, None, g, 4, None, None, None, None, None) == -1
        assert f("rbd_rnea")(A, A, A, g, -1, A, None, None, None, None) == -1
        assert f("rbd_rnea")(A, A, A, g, 0, A, None, None, None, None) == 0
        assert f("rbd_rnea")(A, A, A, g, 4, A, A, None, None, None) == -1          # v without a, f
        assert f("rbd_rnea")(A, A, A, g, 4, M, None, None, None, None) == -1
        assert f("rbd_rnea_fpass")(None, None, None, g, 4, None, None, None, None) == -1
        assert f("rbd_rnea_bpass")(None, None, 4, None, None) == -1
        assert f("rbd_rnea_grad")(A, A, None, g, 0, -5, None, A, None) == -1
        assert f("rbd_rnea_grad")(A, A, A, g, 0, 4, None, None, None) == -1
        assert f("rbd_rnea_grad")(A, A, A, g, 0, 4, None, M, None) == -1
        assert f("rbd_rnea_grad")(A, A, A, g, 1, 0, None, A, None) == 0
        assert f("rbd_rnea_with_grad")(A, A, A, g, 0, 4, A, A, A, None, A, None) == -1
        assert f("rbd_rnea_with_grad")(A, A, A, g, 0, 4, A, A, A, A, M, None) == -1
        assert f("rbd_minv")(A, 4, 1, None, None, 0, None) == -1
        assert f("rbd_minv")(A, -2, 1, A, None, 0, None) == -1
        assert f("rbd_minv")(A, 4, 1, M, None, 0, None) == -1
        for B in (1, 7, 4096, 1 << 20):
            for esz in (2, 4, 8):
                L.rbd_minv_workspace_bytes(B, esz); L.rbd_fd_workspace_bytes(B, esz)
        wsb = L.rbd_minv_workspace_bytes(10, 4 if sfx == "f32" else 8)
        if wsb:
            assert f("rbd_minv")(A, 10, 1, A, None, 0, None) == -3
            assert f("rbd_minv")(A, 10, 1, A, A, wsb - 1, None) == -3
            assert f("rbd_minv")(A, 10, 1, A, M, wsb, None) in (-1, -3)
        assert f("rbd_crba")(None, 4, None, None) < 0
        assert f("rbd_aba")(None, None, None, g, 4, None, None) < 0
        assert f("rbd_forward_dynamics")(None, None, None, g, 4, None, None, 0, None) < 0
        assert f("rbd_forward_dynamics_grad")(A, A, A, g, 4, A, A, None, 0, None) < 0          # workspace missing
        assert f("rbd_forward_dynamics_grad")(A, A, A, g, 4, A, M, A, 1 << 30, None) < 0
        assert f("rbd_rnea_grad_fpass_dq")(None, None, None, None, g, 4, None, None, None, None) < 0
        assert f("rbd_rnea_grad_fpass_dqd")(None, None, None, 4, None, None, None, None) < 0
        assert f("rbd_rnea_grad_bpass_dq")(None, None, None, 4, None, None) < 0
        assert f("rbd_rnea_grad_bpass_dqd")(None, None, 0, 4, None, None) < 0
        assert f("rbd_minv_bpass")(None, 4, None, None, None, None, None) < 0
        assert f("rbd_minv_fpass")(None, 4, None, None, None, None, None) < 0
        assert len(L.rbd_last_error()) > 0
        checks += 30
    # kernel names into buffers of every size (snprintf truncation), bad ops / sizes
    for op in (0, 1, 2, 3, -1):
        for esz in (4, 8, 3):
            for B in (1, 4096, 1 << 20):
                for ln in (1, 2, 8, 33, 128):
                    buf = ctypes.create_string_buffer(ln)
                    L.rbd_kernel_name(op, esz, B, buf, ln)
    L.rbd_kernel_name(1, 4, 16, None, 0)
    # options: out-of-range ids / values are refused or ignored, concurrent set / get on the atomics
    assert L.rbd_set_option(-1, 0) != 0 and L.rbd_set_option(99, 0) != 0
    L.rbd_get_option(-1); L.rbd_get_option(99)

    def hammer(k):
        for i in range(2000):
            L.rbd_set_option(i % 4, (i + k) % 4 if i % 4 != 3 else (i * 1000) % (1 << 30))
            L.rbd_get_option(i % 4)
            buf = ctypes.create_string_buffer(64)
            L.rbd_kernel_name(i % 3, 4 if i % 2 else 8, 1 << (i % 21), buf, 64)
            L.rbd_rnea_f32(None, None, None, ctypes.c_float(0), 4, None, None, None, None, None)    # thread-local error buffer
    ts = [threading.Thread(target=hammer, args=(k,)) for k in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
    for o in range(4):
        L.rbd_set_option(o, 0)
    # the workspace pool with nothing in it (no GPU: nothing was ever allocated), twice
    assert L.rbd_release_workspaces() == 0 and L.rbd_release_workspaces() == 0
    print(f"asan host checks OK ({checks} argument checks, n = {n})")


if __name__ == "__main__":
    main(sys.argv[1])
