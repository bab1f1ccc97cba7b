"""ctypes binding of the model-handle library (include/rbd_generic.h, csrc_generic/rbd_generic.hip).

``librbd_generic.so`` is compiled once (``build.build_generic``, part of ``__graft_entry__.build()``), not per robot:
the packed model becomes an ``rbd_model_desc`` and ``rbd_model_create`` returns a handle.  ``GenericModel`` wraps the
handle(s) of one robot and exposes the entry points under the names and argument lists of the per-robot library
(``rbd_rnea_f32(q, qd, qdd, g, B, c, v, a, f, stream)`` ...), so ``_lib.RbdLibrary`` can hand either to ``api.py``.
HIP kernels only -- like everything else in the package there is no CPU path, and the oracle is never imported.
"""
from __future__ import annotations

import ctypes
import threading
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import numpy as np

from .packer import PackedModel

RBD_G_ABI_VERSION = 1
RBD_G_MAX_BODIES = 64
RBD_G_GRAD_KERNEL_AUTO, RBD_G_GRAD_KERNEL_COLUMNS, RBD_G_GRAD_KERNEL_WORLD = 0, 1, 2

# every symbol include/rbd_generic.h declares (tests check the built library exports all of them)
GENERIC_EXPORTED_SYMBOLS = [
    "rbd_g_abi_version", "rbd_g_last_error", "rbd_model_create", "rbd_model_destroy", "rbd_model_n", "rbd_model_nv",
    "rbd_g_set_grad_kernel", "rbd_g_grad_kernel_of", "rbd_g_set_output_staging",
    "rbd_g_rnea_f32", "rbd_g_rnea_f64", "rbd_g_rnea_grad_f32", "rbd_g_rnea_grad_f64",
    "rbd_g_minv_f32", "rbd_g_minv_f64", "rbd_g_fd_workspace_bytes",
    "rbd_g_forward_dynamics_f32", "rbd_g_forward_dynamics_f64",
    "rbd_g_forward_dynamics_grad_f32", "rbd_g_forward_dynamics_grad_f64",
] + [f"rbd_g_{nm}_{sfx}" for sfx in ("f32", "f64") for nm in (
    "rnea_fpass", "rnea_bpass", "rnea_grad_fpass_dq", "rnea_grad_fpass_dqd", "rnea_grad_bpass_dq", "rnea_grad_bpass_dqd",
    "minv_bpass", "minv_fpass", "crba")]
_PASS_NAMES = ("rnea_fpass", "rnea_bpass", "rnea_grad_fpass_dq", "rnea_grad_fpass_dqd", "rnea_grad_bpass_dq", "rnea_grad_bpass_dqd",
               "minv_bpass", "minv_fpass", "crba")


class RbdModelDesc(Structure):
    _fields_ = [("abi_version", c_int32), ("n", c_int32),
                ("parent", POINTER(c_int32)), ("joint_type", POINTER(c_int32)),
                ("S", POINTER(c_double)), ("X0", POINTER(c_double)), ("Xs", POINTER(c_double)),
                ("Xc", POINTER(c_double)), ("I", POINTER(c_double)), ("damping", POINTER(c_double)),
                ("floating_base", c_int32)]


def _declare(lib):
    lib.rbd_g_abi_version.restype = c_int
    lib.rbd_g_abi_version.argtypes = []
    lib.rbd_g_last_error.restype = c_char_p
    lib.rbd_g_last_error.argtypes = []
    lib.rbd_model_create.restype = c_int
    lib.rbd_model_create.argtypes = [POINTER(RbdModelDesc), c_int, POINTER(c_void_p)]
    lib.rbd_model_destroy.restype = None
    lib.rbd_model_destroy.argtypes = [c_void_p]
    lib.rbd_model_n.restype = c_int
    lib.rbd_model_n.argtypes = [c_void_p]
    lib.rbd_model_nv.restype = c_int
    lib.rbd_model_nv.argtypes = [c_void_p]
    lib.rbd_g_set_grad_kernel.restype = c_int
    lib.rbd_g_set_grad_kernel.argtypes = [c_int]
    lib.rbd_g_set_output_staging.restype = c_int
    lib.rbd_g_set_output_staging.argtypes = [c_int]
    lib.rbd_g_grad_kernel_of.restype = c_int
    lib.rbd_g_grad_kernel_of.argtypes = [c_void_p]
    lib.rbd_g_fd_workspace_bytes.restype = c_size_t
    lib.rbd_g_fd_workspace_bytes.argtypes = [c_void_p, c_int64, c_int, c_int]
    for sfx, ft in (("f32", c_float), ("f64", c_double)):
        f = getattr(lib, f"rbd_g_rnea_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, ft, c_int64] + [c_void_p] * 5
        f = getattr(lib, f"rbd_g_rnea_grad_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, ft, c_int, c_int64] + [c_void_p] * 3
        f = getattr(lib, f"rbd_g_minv_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]
        f = getattr(lib, f"rbd_g_forward_dynamics_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, ft, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]
        f = getattr(lib, f"rbd_g_forward_dynamics_grad_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, ft, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
        # per-pass surface + crba: the argument lists of include/rbd_hip.h behind the model handle
        for nm, at in (("rnea_fpass", [c_void_p] * 3 + [ft, c_int64] + [c_void_p] * 4),
                       ("rnea_bpass", [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
                       ("rnea_grad_fpass_dq", [c_void_p] * 4 + [ft, c_int64] + [c_void_p] * 4),
                       ("rnea_grad_fpass_dqd", [c_void_p] * 3 + [c_int64] + [c_void_p] * 4),
                       ("rnea_grad_bpass_dq", [c_void_p] * 3 + [c_int64] + [c_void_p] * 2),
                       ("rnea_grad_bpass_dqd", [c_void_p] * 2 + [c_int, c_int64] + [c_void_p] * 2),
                       ("minv_bpass", [c_void_p, c_int64] + [c_void_p] * 5),
                       ("minv_fpass", [c_void_p, c_int64] + [c_void_p] * 5),
                       ("crba", [c_void_p, c_int64, c_void_p, c_void_p])):
            f = getattr(lib, f"rbd_g_{nm}_{sfx}")
            f.restype = c_int
            f.argtypes = [c_void_p] + at


_LIB = None
_LIB_LOCK = threading.Lock()


def load_generic_library(build: bool = True):
    """The process-wide ``librbd_generic.so`` (built first if it is missing / stale and ``build``)."""
    global _LIB
    with _LIB_LOCK:
        if _LIB is None:
            import os
            import torch  # noqa: F401  -- FIRST: torch ships its own libamdhip64; loading this library before it would pull
            #               a second HIP runtime into the process (hipErrorNoDevice from whichever initialises later)
            from .build import build_generic, generic_lib_path
            path = build_generic() if build else generic_lib_path()
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} not found; build it with rbdreference_amd.build.build_generic()")
            lib = ctypes.CDLL(path)
            _declare(lib)
            if lib.rbd_g_abi_version() != RBD_G_ABI_VERSION:
                raise RuntimeError(f"{path}: ABI {lib.rbd_g_abi_version()} != {RBD_G_ABI_VERSION}")
            _LIB = lib
        return _LIB


def model_desc_arrays(m: PackedModel):
    """Packed model -> the host arrays of ``rbd_model_desc``: ``X_i(q) = X_J(q) X_i(0)`` split into its constant, sine
    (or linear, for a prismatic joint) and cosine parts (exact: no sampling of transcendental functions)."""
    n = m.n
    S = np.zeros((n, 6)); X0 = np.zeros((n, 6, 6)); Xs = np.zeros((n, 6, 6)); Xc = np.zeros((n, 6, 6))
    for i in range(n):
        k = m.axis[i]; a, b = (k + 1) % 3, (k + 2) % 3
        C0 = np.zeros((6, 6)); Cs = np.zeros((6, 6)); Cc = np.zeros((6, 6))
        if m.floating and i == 0:
            continue                     # the 6-DoF base joint: rbd_model_desc.floating_base, nothing of body 0's joint is read
        if m.jtype[i] == 0:
            S[i, k] = 1.0
            for o in (0, 3):
                C0[o + k, o + k] = 1.0
                Cc[o + a, o + a] = 1.0; Cc[o + b, o + b] = 1.0
                Cs[o + a, o + b] = 1.0; Cs[o + b, o + a] = -1.0
        elif m.jtype[i] == 1:
            S[i, 3 + k] = 1.0
            C0 = np.eye(6)
            Cs[3 + a, b] = 1.0; Cs[3 + b, a] = -1.0
        else:
            raise ValueError(f"body {i}: joint type {m.jtype[i]} is not served by the model-handle library")
        X0[i] = C0 @ m.Xtree[i]; Xs[i] = Cs @ m.Xtree[i]; Xc[i] = Cc @ m.Xtree[i]
    jt = [0 if (m.floating and i == 0) else t for i, t in enumerate(m.jtype)]
    return dict(parent=np.ascontiguousarray(m.parent, dtype=np.int32), joint_type=np.ascontiguousarray(jt, dtype=np.int32),
                S=np.ascontiguousarray(S), X0=np.ascontiguousarray(X0), Xs=np.ascontiguousarray(Xs), Xc=np.ascontiguousarray(Xc),
                I=np.ascontiguousarray(m.I, dtype=np.float64), damping=np.ascontiguousarray(m.damping, dtype=np.float64))


class GenericModel:
    """One robot on the model-handle library: a handle per device (created on first use of that device), and the
    per-robot library's entry-point names bound to it."""
    is_generic = True
    SERVES = ("rbd_rnea", "rbd_rnea_grad", "rbd_rnea_with_grad", "rbd_minv", "rbd_forward_dynamics",
              "rbd_forward_dynamics_grad", "rbd_minv_workspace_bytes", "rbd_fd_workspace_bytes")

    def __init__(self, model: PackedModel, build: bool = True):
        self.model = model
        self.lib = load_generic_library(build)
        self._arr = model_desc_arrays(model)
        self._handles = {}
        self._lock = threading.Lock()
        for sfx in ("f32", "f64"):
            for base in ("rnea", "rnea_grad", "forward_dynamics", "forward_dynamics_grad"):
                setattr(self, f"rbd_{base}_{sfx}", self._bind(f"rbd_g_{base}_{sfx}"))
            for base in _PASS_NAMES:                 # fixed base only (the library refuses a floating-base model)
                setattr(self, f"rbd_{base}_{sfx}", self._bind(f"rbd_g_{base}_{sfx}"))
            setattr(self, f"rbd_minv_{sfx}", self._bind_minv(sfx))
            setattr(self, f"rbd_rnea_with_grad_{sfx}", self._bind_with_grad(sfx))
            setattr(self, f"rbd_aba_{sfx}", self._bind_aba(sfx))

    # ---- handles ----------------------------------------------------------------------------------------------------
    def handle(self, device: int | None = None):
        import torch
        dev = torch.cuda.current_device() if device is None else int(device)
        h = self._handles.get(dev)
        if h is None:
            with self._lock:
                h = self._handles.get(dev)
                if h is None:
                    a = self._arr
                    d = RbdModelDesc(RBD_G_ABI_VERSION, self.model.n,
                                     a["parent"].ctypes.data_as(POINTER(c_int32)), a["joint_type"].ctypes.data_as(POINTER(c_int32)),
                                     *[a[k].ctypes.data_as(POINTER(c_double)) for k in ("S", "X0", "Xs", "Xc", "I", "damping")],
                                     1 if self.model.floating else 0)
                    out = c_void_p()
                    rc = self.lib.rbd_model_create(ctypes.byref(d), dev, ctypes.byref(out))
                    if rc != 0:
                        from ._lib import RbdError
                        raise RbdError(rc, (self.lib.rbd_g_last_error() or b"").decode())
                    h = out.value
                    self._handles[dev] = h
        return h

    def close(self):
        with self._lock:
            for h in self._handles.values():
                self.lib.rbd_model_destroy(h)
            self._handles.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass

    # ---- the per-robot library's names --------------------------------------------------------------------------------
    def _bind(self, name):
        fn = getattr(self.lib, name)
        return lambda *args: fn(self.handle(), *args)

    def _bind_minv(self, sfx):
        fn = getattr(self.lib, f"rbd_g_minv_{sfx}")
        return lambda q, B, dense, M, ws, wsb, stream: fn(self.handle(), q, B, dense, M, stream)

    def _bind_with_grad(self, sfx):
        rnea = getattr(self.lib, f"rbd_g_rnea_{sfx}")
        grad = getattr(self.lib, f"rbd_g_rnea_grad_{sfx}")

        def call(q, qd, qdd, g, damp, B, c, v, a, f, dc, stream):
            h = self.handle()
            rc = rnea(h, q, qd, qdd, g, B, c, v, a, f, stream)
            return rc if rc != 0 else grad(h, q, qd, qdd, g, damp, B, None, dc, stream)
        return call

    def _bind_aba(self, sfx):
        """rbd_aba (RBDReference.py:940-1024: defined against forward_dynamics, as for the per-robot libraries): the
        model-handle library has no articulated-body kernel, it evaluates Minv (tau - c) and owns the scratch here."""
        fd = getattr(self.lib, f"rbd_g_forward_dynamics_{sfx}")
        esz = 4 if sfx == "f32" else 8

        def call(q, qd, tau, g, B, qdd, stream):
            import torch
            h = self.handle()
            wsb = int(self.lib.rbd_g_fd_workspace_bytes(h, B, esz, 0))
            ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
            rc = fd(h, q, qd, tau, g, B, qdd, ws.data_ptr(), wsb, stream)
            ws.record_stream(torch.cuda.current_stream())       # (the launches are asynchronous: the block outlives them)
            return rc
        return call

    def rbd_minv_workspace_bytes(self, B, esz):
        return 0

    def rbd_fd_workspace_bytes(self, B, esz):
        return int(self.lib.rbd_g_fd_workspace_bytes(self.handle(), B, esz, 1))

    def rbd_last_error(self):
        return self.lib.rbd_g_last_error()

    def serves(self, base: str) -> bool:
        if base == "rbd_aba":
            return not self.model.floating       # (the reference's own aba raises for a floating base, :900)
        if base.startswith("rbd_") and base[4:] in _PASS_NAMES:
            return not self.model.floating       # per-pass surface + crba: fixed-base models
        return base in self.SERVES

    def kernel_name(self, op: int, elem_size: int) -> str:
        n = self.model.n
        nm = 8 if n <= 8 else 16 if n <= 16 else 32 if n <= 32 else 64
        name = ('g_rnea_kernel', 'g_rnea_grad_kernel', 'g_minv_kernel')[op]
        if op == 1 and self._handles:        # (the choice between the two gradient kernels is the library's: ask it)
            if self.lib.rbd_g_grad_kernel_of(next(iter(self._handles.values()))) == RBD_G_GRAD_KERNEL_WORLD:
                name = "g_rnea_grad_world_kernel"
        return f"{name}<{'float' if elem_size == 4 else 'double'}, {nm}>"
