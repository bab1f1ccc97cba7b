"""ctypes front-end of oracle/rbd_oracle.c -- TEST INFRASTRUCTURE (see that file's header).

    from oracle import c_oracle
    co = c_oracle.COracle(robot)            # builds oracle/_build/librbd_oracle.so with gcc if needed
    c, dc_du = co.rnea_grad(q, qd, qdd)     # [B, n] float64 in -> numpy out, OpenMP over rows
"""
import ctypes
import os
import subprocess

import numpy as np

from . import rbd_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "librbd_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "rbd_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        cmd = ["gcc", "-O2", "-fPIC", "-fopenmp", "-std=c11", "-shared", "-o", LIB, src, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed for oracle/rbd_oracle.c:\n" + r.stderr)
    return LIB


class _Model(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int), ("parent", ctypes.c_void_p), ("prismatic", ctypes.c_void_p),
                ("S", ctypes.c_void_p), ("I", ctypes.c_void_p), ("X0", ctypes.c_void_p),
                ("Xs", ctypes.c_void_p), ("Xc", ctypes.c_void_p), ("damping", ctypes.c_void_p)]


class COracle:
    def __init__(self, robot):
        self.lib = ctypes.CDLL(build())
        m = orc.model_from_robot(robot)
        self.n = m.n
        self._keep = dict(parent=np.ascontiguousarray(m.parent, dtype=np.int32),
                          prismatic=np.ascontiguousarray(m.prismatic, dtype=np.int32),
                          S=np.ascontiguousarray(m.S), I=np.ascontiguousarray(m.I),
                          X0=np.ascontiguousarray(m.X0), Xs=np.ascontiguousarray(m.Xs),
                          Xc=np.ascontiguousarray(m.Xc), damping=np.ascontiguousarray(m.damping))
        self.model = _Model(m.n, *[self._keep[k].ctypes.data for k in
                                   ("parent", "prismatic", "S", "I", "X0", "Xs", "Xc", "damping")])
        self.lib.rbdo_num_threads.restype = ctypes.c_int
        dp = ctypes.c_void_p
        self.lib.rbdo_rnea_grad.argtypes = [ctypes.POINTER(_Model), dp, dp, dp, ctypes.c_double, ctypes.c_int,
                                            ctypes.c_int64, dp, dp, ctypes.c_int]
        self.lib.rbdo_rnea.argtypes = [ctypes.POINTER(_Model), dp, dp, dp, ctypes.c_double, ctypes.c_int64,
                                       dp, dp, dp, dp, ctypes.c_int]
        self.lib.rbdo_minv.argtypes = [ctypes.POINTER(_Model), dp, ctypes.c_int, ctypes.c_int64, dp, ctypes.c_int]

    @property
    def max_threads(self) -> int:
        return int(self.lib.rbdo_num_threads())

    @staticmethod
    def _a(x):
        return np.ascontiguousarray(np.atleast_2d(np.asarray(x, dtype=np.float64)))

    def rnea_grad(self, q, qd, qdd=None, GRAVITY=-9.81, USE_VELOCITY_DAMPING=False, threads=0, out=None):
        q, qd = self._a(q), self._a(qd)
        qdd = None if qdd is None else self._a(qdd)
        B, n = q.shape
        if out is None:
            c = np.empty((B, n)); dc = np.empty((B, n, 2 * n))
        else:                       # reuse caller's buffers (timing loops: no page faults per call)
            c, dc = out
            assert c.shape == (B, n) and dc.shape == (B, n, 2 * n) and c.flags.c_contiguous and dc.flags.c_contiguous
        self.lib.rbdo_rnea_grad(ctypes.byref(self.model), q.ctypes.data, qd.ctypes.data,
                                None if qdd is None else qdd.ctypes.data, float(GRAVITY),
                                int(USE_VELOCITY_DAMPING), B, c.ctypes.data, dc.ctypes.data, int(threads))
        return c, dc

    def rnea(self, q, qd, qdd=None, GRAVITY=-9.81, threads=0):
        q, qd = self._a(q), self._a(qd)
        qdd = None if qdd is None else self._a(qdd)
        B, n = q.shape
        c = np.empty((B, n)); v = np.empty((B, 6, n)); a = np.empty((B, 6, n)); f = np.empty((B, 6, n))
        self.lib.rbdo_rnea(ctypes.byref(self.model), q.ctypes.data, qd.ctypes.data,
                           None if qdd is None else qdd.ctypes.data, float(GRAVITY), B,
                           c.ctypes.data, v.ctypes.data, a.ctypes.data, f.ctypes.data, int(threads))
        return c, v, a, f

    def minv(self, q, output_dense=True, threads=0):
        q = self._a(q)
        B, n = q.shape
        M = np.empty((B, n, n))
        self.lib.rbdo_minv(ctypes.byref(self.model), q.ctypes.data, int(bool(output_dense)), B, M.ctypes.data,
                           int(threads))
        return M
