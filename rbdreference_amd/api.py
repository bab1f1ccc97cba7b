"""Drop-in ``RBDReference`` whose rnea / rnea_grad / minv run as HIP kernels on MI355X.

Mirrors the reference's class (``/root/reference/RBDReference.py:5-7``) and the three methods its
README lists (``README.md:15-17``): same names, keyword arguments, defaults and return arity.
Extension: ``q, qd, qdd`` may be ``[n]`` (one configuration, reference shapes come back) or
``[B, n]`` (a batch; every output gains a leading B).  PyTorch-ROCm tensors in -> tensors out on the
same device / dtype (float32 or float64); numpy arrays or lists in -> uploaded to ``cuda:0`` in
float64, numpy out.  There is no CPU implementation here: without a GPU and the per-robot HIP
library every call raises.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ._lib import RbdLibrary
from .packer import PackedModel, pack_robot

__all__ = ["RBDReference"]


class RBDReference:
    def __init__(self, robotObj, build: bool = True):
        self.robot = robotObj                     # RBDReference.py:7
        self.model: PackedModel = pack_robot(robotObj)
        self._lib = RbdLibrary(self.model, build=build)
        self.n = self.model.n

    # ------------------------------------------------------------------------------------
    def _prep(self, *arrs):
        """Normalise inputs -> (list of contiguous [B, n] device tensors, unbatched?, numpy?)."""
        first = arrs[0]
        is_np = not isinstance(first, torch.Tensor)
        if is_np:
            if not torch.cuda.is_available():
                raise RuntimeError("rbdreference_amd needs a ROCm GPU: numpy inputs are uploaded to "
                                   "cuda:0 and run through the HIP kernels (no CPU fallback)")
            dev, dt = torch.device("cuda", 0), torch.float64
        else:
            dev, dt = first.device, first.dtype
            if dev.type != "cuda":
                raise RuntimeError("inputs must live on a ROCm GPU (cuda device); no CPU fallback")
            if dt not in (torch.float32, torch.float64):
                raise TypeError(f"unsupported dtype {dt}; use float32 or float64")
        out = []
        unb = None
        for x in arrs:
            if x is None:
                out.append(None)
                continue
            if isinstance(x, torch.Tensor):
                if is_np:
                    raise TypeError("mixing numpy and torch inputs is not supported")
                if x.device != dev or x.dtype != dt:
                    raise TypeError("all inputs must share device and dtype")
                t = x
            else:
                t = torch.as_tensor(np.asarray(x, dtype=np.float64), device=dev)
            u = t.dim() == 1
            if u:
                t = t[None, :]
            if t.dim() != 2 or t.shape[1] != self.n:
                raise ValueError(f"expected shape [{self.n}] or [B, {self.n}], got {tuple(x.shape) if hasattr(x, 'shape') else len(x)}")
            if unb is None:
                unb, B = u, t.shape[0]
            elif u != unb or t.shape[0] != B:
                raise ValueError("q, qd, qdd must have identical shapes")
            out.append(t.contiguous())
        return out, unb, is_np, dev, dt

    @staticmethod
    def _ptr(t: Optional[torch.Tensor]):
        return None if t is None else t.data_ptr()

    @staticmethod
    def _ret(t: torch.Tensor, unb: bool, is_np: bool):
        if unb:
            t = t[0]
        return t.cpu().numpy() if is_np else t

    def _fn(self, base: str, dt):
        return getattr(self._lib.lib, f"{base}_{'f32' if dt == torch.float32 else 'f64'}")

    # ------------------------------------------------------------------------------------
    def rnea(self, q, qd, qdd=None, GRAVITY=-9.81, f_ext=None, outputs: str = "cvaf"):
        """RBDReference.rnea (``RBDReference.py:623-628``) -> ``(c, v, a, f)``; ``f`` is the
        accumulated force, ``f_ext`` is accepted and ignored exactly as there.
        ``outputs="c"`` skips v, a, f (returns ``(c, None, None, None)``)."""
        (q, qd, qdd), unb, is_np, dev, dt = self._prep(q, qd, qdd)
        B = q.shape[0]
        with torch.cuda.device(dev):
            c = torch.empty((B, self.n), device=dev, dtype=dt)
            if outputs == "cvaf":
                v = torch.empty((B, 6, self.n), device=dev, dtype=dt)
                a = torch.empty_like(v)
                f = torch.empty_like(v)
            elif outputs == "c":
                v = a = f = None
            else:
                raise ValueError("outputs must be 'cvaf' or 'c'")
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_rnea", dt)(
                self._ptr(q), self._ptr(qd), self._ptr(qdd), float(GRAVITY), B,
                self._ptr(c), self._ptr(v), self._ptr(a), self._ptr(f), st))
        if v is None:
            return self._ret(c, unb, is_np), None, None, None
        return tuple(self._ret(t, unb, is_np) for t in (c, v, a, f))

    def rnea_fpass(self, q, qd, qdd=None, GRAVITY=-9.81):
        """RBDReference.rnea_fpass (``RBDReference.py:559-598``) -> ``(v, a, f)`` with the LOCAL body
        forces (no backward accumulation) -- the per-pass surface of ``README.md:19``."""
        (q, qd, qdd), unb, is_np, dev, dt = self._prep(q, qd, qdd)
        B = q.shape[0]
        with torch.cuda.device(dev):
            v = torch.empty((B, 6, self.n), device=dev, dtype=dt)
            a = torch.empty_like(v)
            f = torch.empty_like(v)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_rnea_fpass", dt)(
                self._ptr(q), self._ptr(qd), self._ptr(qdd), float(GRAVITY), B,
                self._ptr(v), self._ptr(a), self._ptr(f), st))
        return tuple(self._ret(t, unb, is_np) for t in (v, a, f))

    def rnea_bpass(self, q, f):
        """RBDReference.rnea_bpass (``RBDReference.py:600-621``) -> ``(c, f)``.  Like the reference it
        accumulates into ``f`` IN PLACE when ``f`` is a contiguous device tensor (``:619``) and
        returns that same tensor; numpy input gets a new array back."""
        (q,), unb, is_np, dev, dt = self._prep(q)
        B = q.shape[0]
        if isinstance(f, torch.Tensor):
            if is_np or f.device != dev or f.dtype != dt:
                raise TypeError("f must share device and dtype with q")
            ft = f[None] if unb else f
            if ft.shape != (B, 6, self.n):
                raise ValueError(f"f must have shape [B, 6, {self.n}]")
            fc = ft if ft.is_contiguous() else ft.contiguous()
        else:
            fa = np.asarray(f, dtype=np.float64)
            fa = fa[None] if unb else fa
            if fa.shape != (B, 6, self.n):
                raise ValueError(f"f must have shape [B, 6, {self.n}]")
            fc = torch.as_tensor(fa, device=dev).contiguous()
            ft = None
        with torch.cuda.device(dev):
            c = torch.empty((B, self.n), device=dev, dtype=dt)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_rnea_bpass", dt)(self._ptr(q), self._ptr(fc), B, self._ptr(c), st))
        if ft is not None and fc is not ft:
            ft.copy_(fc)
        if ft is not None:
            return self._ret(c, unb, is_np), f
        return self._ret(c, unb, is_np), self._ret(fc, unb, is_np)

    def rnea_grad(self, q, qd, qdd=None, GRAVITY=-9.81, USE_VELOCITY_DAMPING=False,
                  return_c: bool = False):
        """RBDReference.rnea_grad (``RBDReference.py:1345-1368``) -> ``dc_du = [dc_dq | dc_dqd]``,
        ``(n, 2n)`` per configuration.  ``return_c=True`` also returns the bias force ``c`` the
        reference computes internally (``:1353``) -> ``(c, dc_du)``."""
        (q, qd, qdd), unb, is_np, dev, dt = self._prep(q, qd, qdd)
        B = q.shape[0]
        with torch.cuda.device(dev):
            dc = torch.empty((B, self.n, 2 * self.n), device=dev, dtype=dt)
            c = torch.empty((B, self.n), device=dev, dtype=dt) if return_c else None
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_rnea_grad", dt)(
                self._ptr(q), self._ptr(qd), self._ptr(qdd), float(GRAVITY),
                1 if USE_VELOCITY_DAMPING else 0, B, self._ptr(c), self._ptr(dc), st))
        if return_c:
            return self._ret(c, unb, is_np), self._ret(dc, unb, is_np)
        return self._ret(dc, unb, is_np)

    def minv(self, q, output_dense=True):
        """RBDReference.minv (``RBDReference.py:785-806``) -> ``(n, n)`` per configuration.
        ``output_dense=False`` returns the upper triangle with a ZERO strict lower triangle (the
        reference leaves forward-pass by-products there, ``:771``)."""
        (q,), unb, is_np, dev, dt = self._prep(q)
        B = q.shape[0]
        esz = 4 if dt == torch.float32 else 8
        with torch.cuda.device(dev):
            M = torch.empty((B, self.n, self.n), device=dev, dtype=dt)
            wsb = int(self._lib.lib.rbd_minv_workspace_bytes(B, esz))
            ws = torch.empty((max(wsb, 1),), device=dev, dtype=torch.uint8)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_minv", dt)(
                self._ptr(q), B, 1 if output_dense else 0, self._ptr(M), ws.data_ptr(), wsb, st))
        return self._ret(M, unb, is_np)

    def crba(self, q):
        """RBDReference.crba (fixed-base branch, ``RBDReference.py:1091-1124``) -> joint-space inertia
        ``H``, ``(n, n)`` per configuration."""
        (q,), unb, is_np, dev, dt = self._prep(q)
        B = q.shape[0]
        with torch.cuda.device(dev):
            H = torch.empty((B, self.n, self.n), device=dev, dtype=dt)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_crba", dt)(self._ptr(q), B, self._ptr(H), st))
        return self._ret(H, unb, is_np)

    # ---- next row of SURVEY.md §8f: forward dynamics on top of the three kernels --------------
    def _fd(self, q, qd, u, GRAVITY, want_grad):
        (q, qd, u), unb, is_np, dev, dt = self._prep(q, qd, u)
        B = q.shape[0]
        esz = 4 if dt == torch.float32 else 8
        with torch.cuda.device(dev):
            qdd = torch.empty((B, self.n), device=dev, dtype=dt)
            wsb = int(self._lib.lib.rbd_fd_workspace_bytes(B, esz))
            ws = torch.empty((max(wsb, 1),), device=dev, dtype=torch.uint8)
            st = torch.cuda.current_stream(dev).cuda_stream
            if want_grad:
                d = torch.empty((B, self.n, 2 * self.n), device=dev, dtype=dt)
                self._lib.check(self._fn("rbd_forward_dynamics_grad", dt)(
                    self._ptr(q), self._ptr(qd), self._ptr(u), float(GRAVITY), B, self._ptr(qdd),
                    self._ptr(d), ws.data_ptr(), wsb, st))
                return qdd, d, unb, is_np
            self._lib.check(self._fn("rbd_forward_dynamics", dt)(
                self._ptr(q), self._ptr(qd), self._ptr(u), float(GRAVITY), B, self._ptr(qdd),
                ws.data_ptr(), wsb, st))
            return qdd, None, unb, is_np

    def forward_dynamics(self, q, qd, u, GRAVITY=-9.81):
        """RBDReference.forward_dynamics (``RBDReference.py:1371-1374``): ``minv(q) @ (u - c(q, qd))``."""
        qdd, _, unb, is_np = self._fd(q, qd, u, GRAVITY, False)
        return self._ret(qdd, unb, is_np)

    def forward_dynamics_grad(self, q, qd, u, GRAVITY=-9.81):
        """RBDReference.forward_dynamics_grad (``RBDReference.py:1376-1384``) -> ``(qdd_dq, qdd_dqd)``,
        each ``(n, n)`` per configuration (views of one ``[B, n, 2n]`` buffer for tensor inputs)."""
        _, d, unb, is_np = self._fd(q, qd, u, GRAVITY, True)
        n = self.n
        d = self._ret(d, unb, is_np)
        return d[..., :n], d[..., n:]
