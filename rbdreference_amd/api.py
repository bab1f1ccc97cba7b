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

import contextlib
import os
import threading
from sys import getrefcount as _refs
from typing import Optional

import numpy as np
import torch

from ._lib import RbdLibrary
from .packer import PackedModel, pack_robot

__all__ = ["RBDReference", "BoundLaunch"]

_T = torch.Tensor
_F32, _F64 = torch.float32, torch.float64
_storage_uses = getattr(torch._C, "_storage_Use_Count", None)          # holders of a StorageImpl (views, detached aliases)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)     # hipStream_t of torch's current stream, as an int
_cur_dev = torch.cuda.current_device
_tid = threading.get_ident

# ---- one-shot calls without per-call allocations --------------------------------------------
# A method call used to cost 7-14 us of Python around an 11-25 us kernel: shape checks, a device guard, a stream
# look-up and one torch.empty per output.  The second call with the same signature (op, B, dtype, device, stream)
# is now: validate three tensors by attribute, take a FREE output set from a small pool, ONE ctypes call.
# "Free" keeps the reference's contract that every call returns fresh outputs (RBDReference.py:623, :785, :1345):
# a set is handed out again only when the caller holds neither its tensors (Python reference counts) nor any view or
# alias of their storage (StorageImpl use count) any more; while anything is still held, another set is used (up to
# POOL_SETS per signature, then a new one replaces the oldest -- which stays valid for whoever holds it).  Big outputs
# (> POOL_MAX_BYTES per set) are never pooled: there the kernel dwarfs the Python.  RBD_OUTPUT_POOL=0 turns it off.
POOL_SETS = 3
POOL_MAX_BYTES = 160 << 20
POOL_SIGNATURES = 16


class _OutSet:
    __slots__ = ("flat", "outs", "ptrs", "cdata", "base")

    def __init__(self, shapes, dt, dev):
        esz = 4 if dt is _F32 else 8
        sizes = [(int(np.prod(sh)) * esz + 15) & ~15 for sh in shapes]
        self.flat = torch.empty((sum(sizes),), device=dev, dtype=torch.uint8)
        outs, off = [], 0
        for sh, nb in zip(shapes, sizes):
            outs.append(self.flat[off:off + int(np.prod(sh)) * esz].view(dt).view(sh))
            off += nb
        self.outs = tuple(outs)
        del outs
        self.ptrs = tuple(t.data_ptr() for t in self.outs)
        self.cdata = self.flat.untyped_storage()._cdata
        self.base = None
        self.base = self._probe()

    def _probe(self):
        return (_storage_uses(self.cdata), _refs(self.outs), *[_refs(t) for t in self.outs])

    def free(self):
        return self._probe() == self.base


class _Plan:
    """Everything of a call signature that does not change between calls."""
    __slots__ = ("fn", "sets", "epoch", "ws", "wsp", "wsb", "shapes", "dt", "dev")

    def __init__(self, fn, shapes, dt, dev, epoch, ws=None, wsb=0):
        self.fn, self.shapes, self.dt, self.dev, self.epoch = fn, shapes, dt, dev, epoch
        self.ws, self.wsb, self.wsp = ws, wsb, (ws.data_ptr() if ws is not None else None)
        self.sets = []

    def take(self):
        for s in self.sets:
            if s.free():
                return s
        s = _OutSet(self.shapes, self.dt, self.dev)
        if len(self.sets) >= POOL_SETS:
            self.sets.pop(0)
        self.sets.append(s)
        return s


class BoundLaunch:
    """What `RBDReference.bind` returns: ``launch()`` enqueues the bound entry point on torch's current stream of the
    inputs' device and returns ``launch.outputs`` (the same tensors every time).  The library that answers is the one
    that was serving the robot when `bind` ran (bind again after `RbdLibrary.wait_specialized()` to move from the
    model-handle library to the robot's own)."""

    def __init__(self, rbd, fn, args, inputs, outputs, keep, dev):
        self._rbd, self._fn, self._args, self.inputs, self.outputs, self._keep, self._dev = rbd, fn, args, inputs, outputs, keep, dev
        self._idx = dev.index if dev.index is not None else torch.cuda.current_device()

    def __call__(self):
        if torch.cuda.current_device() != self._idx:
            with torch.cuda.device(self._dev):
                return self.__call__()
        rc = self._fn(*self._args(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            self._rbd._lib.check(rc)
        return self.outputs


class RBDReference:
    def __init__(self, robotObj, build: bool = True, generic=None):
        """``generic`` ('auto' | 'only' | 'never', default ``RBD_GENERIC`` or 'auto'): whether the model-handle library
        (include/rbd_generic.h) may answer rnea / rnea_grad / minv / forward_dynamics(_grad) while -- or instead of --
        the robot's own compiled library (``_lib.RbdLibrary``)."""
        self.robot = robotObj                     # RBDReference.py:7
        self.model: PackedModel = pack_robot(robotObj)
        self._lib = RbdLibrary(self.model, build=build, generic=generic)
        self.n = self.model.n            # bodies
        self.nv = self.model.nv          # columns of q, qd, qdd, c: n, or n + 5 with a floating base
        self._plans = {}
        self._pool_on = (os.environ.get("RBD_OUTPUT_POOL", "1") != "0" and _storage_uses is not None and _raw_stream is not None)

    # ---- cached launch plans (see the note at the top of the module) ---------------------------------------------
    def _sig(self, q, qd, qdd):
        """``(B, dtype, device index, raw stream)`` when q, qd (qdd unless None) are plain contiguous ``[B, nv]`` tensors of one
        dtype on one GPU and the robot's own library is loaded; None sends the call down the general path."""
        if not self._pool_on or type(q) is not _T or self._lib._full is None:
            return None
        dt = q.dtype
        if dt is not _F32 and dt is not _F64:
            return None
        idx = q.get_device()
        shp = q.shape
        if idx < 0 or len(shp) != 2 or shp[1] != self.nv or shp[0] == 0 or not q.is_contiguous():
            return None
        if qd is not None and (type(qd) is not _T or qd.dtype is not dt or qd.get_device() != idx or qd.shape != shp or not qd.is_contiguous()):
            return None
        if qdd is not None and (type(qdd) is not _T or qdd.dtype is not dt or qdd.get_device() != idx or qdd.shape != shp or not qdd.is_contiguous()):
            return None
        st = _raw_stream(idx)
        if st == 0 and _cur_dev() != idx:       # the null stream belongs to the CURRENT device (include/rbd_hip.h)
            return None
        return shp[0], dt, idx, st

    def _plan(self, key, make):
        key = (_tid(), key)                     # one pool per host thread: a set is "free" by reference counts, which another
        p = self._plans.get(key)                # thread could not tell from "handed out a moment ago, not yet bound"
        if p is None or p.epoch != self._lib.epoch:
            p = make()
            if p is None:
                return None
            if len(self._plans) >= POOL_SIGNATURES:
                self._plans.pop(next(iter(self._plans)))
            self._plans[key] = p
        return p

    def _mk_plan(self, base, dt, idx, shapes, has_qdd=True, ws_query=None):
        """A `_Plan` for entry point ``base``: function object, output shapes, and -- resolved from the SAME library object
        -- the scratch the entry point needs (``ws_query(lib, esz) -> bytes``)."""
        esz = 4 if dt is _F32 else 8
        if sum(int(np.prod(sh)) for sh in shapes) * esz > POOL_MAX_BYTES:
            return None
        dev = torch.device("cuda", idx)
        sfx = "f32" if esz == 4 else "f64"
        lib = self._lib.resolve(base, sfx, has_qdd)
        ws, wsb = None, 0
        if ws_query is not None:
            wsb = int(ws_query(lib, esz))
            ws = torch.empty((max(wsb, 1),), device=dev, dtype=torch.uint8)
        return _Plan(getattr(lib, f"{base}_{sfx}"), shapes, dt, dev, self._lib.epoch, ws, wsb)

    def release_pools(self):
        """Drop the cached launch plans and their pooled output sets (tensors the caller still holds stay valid)."""
        self._plans.clear()

    @contextlib.contextmanager
    def shard_of(self, global_rows: int):
        """Declare that the calls inside the block evaluate SHARDS of a global batch of ``global_rows`` rows: wherever
        the library picks a kernel from the batch size it then decides from the global size (``RBD_OPT_SELECT_BATCH``,
        include/rbd_hip.h), so a shard is computed by the same kernel -- and is bit-identical, row by row -- as the
        unsharded call.  Process-wide like every option; restored on exit."""
        from ._lib import RBD_OPT_SELECT_BATCH
        g = int(global_rows)
        if not (0 <= g < 2 ** 31):
            raise ValueError("global_rows must be in [0, 2^31)")
        old = self._lib.get_option(RBD_OPT_SELECT_BATCH)
        self._lib.set_option(RBD_OPT_SELECT_BATCH, g)
        try:
            yield self
        finally:
            self._lib.set_option(RBD_OPT_SELECT_BATCH, old)

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
            if t.dim() != 2 or t.shape[1] != self.nv:
                raise ValueError(f"expected shape [{self.nv}] or [B, {self.nv}], got {tuple(x.shape) if hasattr(x, 'shape') else len(x)}")
            if unb is None:
                unb, B = u, t.shape[0]
            elif u != unb or t.shape[0] != B:
                raise ValueError("q, qd, qdd must have identical shapes")
            out.append(t.contiguous())
        return out, unb, is_np, dev, dt

    def _minv_ws(self, B: int, esz: int) -> int:
        """rbd_minv_workspace_bytes from the library that serves rbd_minv of that precision right now (a QUERY for
        callers who bring their own scratch; the methods themselves resolve the library once per call)."""
        lib = self._lib.serving("rbd_minv", "f32" if esz == 4 else "f64")
        return int(lib.rbd_minv_workspace_bytes(B, esz))

    @staticmethod
    def _ptr(t: Optional[torch.Tensor]):
        return None if t is None else t.data_ptr()

    @staticmethod
    def _ret(t: torch.Tensor, unb: bool, is_np: bool):
        if unb:
            t = t[0]
        return t.cpu().numpy() if is_np else t

    @staticmethod
    def _check_out(t, shape, dev, dt, name):
        """A caller-provided output tensor (`out=`): right shape, dtype, device, contiguous."""
        if not isinstance(t, torch.Tensor) or tuple(t.shape) != tuple(shape) or t.dtype != dt or t.device != dev or not t.is_contiguous():
            raise ValueError(f"out: `{name}` must be a contiguous {dt} tensor of shape {tuple(shape)} on {dev}")
        return t

    @staticmethod
    def _out_ok(t, shape, dt, idx):
        """`out=` tensor usable by the cached launch plan as it is (else the general path checks it and says what is wrong)."""
        return type(t) is _T and t.dtype is dt and t.get_device() == idx and tuple(t.shape) == shape and t.is_contiguous()

    def _fn(self, base: str, dt, has_qdd: bool = True):
        return self._lib.fn(base, "f32" if dt == torch.float32 else "f64", has_qdd)

    # ------------------------------------------------------------------------------------
    def rnea(self, q, qd, qdd=None, GRAVITY=-9.81, f_ext=None, outputs: str = "cvaf", out=None):
        """RBDReference.rnea (``RBDReference.py:623-628``) -> ``(c, v, a, f)``; ``f`` is the
        accumulated force, ``f_ext`` is accepted and ignored exactly as there.
        ``outputs="c"`` skips v, a, f (returns ``(c, None, None, None)``).  ``out=(c, v, a, f)``: write into these
        pre-allocated device tensors (batched tensor inputs only) -- a loop that owns its buffers then pays the
        launch alone, not four allocations per call."""
        if out is None and outputs == "cvaf":
            sg = self._sig(q, qd, qdd)
            if sg is not None:
                B, dt, idx, st = sg
                p = self._plan(("rnea", B, dt, idx, st), lambda: self._mk_plan("rbd_rnea", dt, idx, ((B, self.nv), (B, 6, self.n), (B, 6, self.n), (B, 6, self.n))))
                if p is not None:
                    o = p.take()
                    rc = p.fn(q.data_ptr(), qd.data_ptr(), None if qdd is None else qdd.data_ptr(), float(GRAVITY), B, *o.ptrs, st)
                    if rc != 0:
                        self._lib.check(rc)
                    return o.outs
        if out is not None and outputs == "cvaf" and len(out) == 4:
            # pre-allocated outputs through the cached plan too: the call is the same single ctypes launch, without the pool
            sg = self._sig(q, qd, qdd)
            if sg is not None:
                B, dt, idx, st = sg
                vs = (B, 6, self.n)
                if self._out_ok(out[0], (B, self.nv), dt, idx) and all(self._out_ok(t, vs, dt, idx) for t in out[1:]):
                    p = self._plan(("rnea", B, dt, idx, st), lambda: self._mk_plan("rbd_rnea", dt, idx, ((B, self.nv), vs, vs, vs)))
                    if p is not None:
                        rc = p.fn(q.data_ptr(), qd.data_ptr(), None if qdd is None else qdd.data_ptr(), float(GRAVITY), B,
                                  out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), st)
                        if rc != 0:
                            self._lib.check(rc)
                        return out[0], out[1], out[2], out[3]
        (q, qd, qdd), unb, is_np, dev, dt = self._prep(q, qd, qdd)
        B = q.shape[0]
        if out is not None:
            if is_np or unb or outputs != "cvaf" or len(out) != 4:
                raise ValueError("out= needs batched tensor inputs, outputs='cvaf' and four tensors (c, v, a, f)")
            c = self._check_out(out[0], (B, self.nv), dev, dt, "c")
            v, a, f = (self._check_out(t, (B, 6, self.n), dev, dt, nm) for t, nm in zip(out[1:], "vaf"))
            with torch.cuda.device(dev):
                self._lib.check(self._fn("rbd_rnea", dt)(
                    self._ptr(q), self._ptr(qd), self._ptr(qdd), float(GRAVITY), B,
                    self._ptr(c), self._ptr(v), self._ptr(a), self._ptr(f), torch.cuda.current_stream(dev).cuda_stream))
            return c, v, a, f
        with torch.cuda.device(dev):
            c = torch.empty((B, self.nv), device=dev, dtype=dt)
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
            c = torch.empty((B, self.nv), device=dev, dtype=dt)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_rnea_bpass", dt)(self._ptr(q), self._ptr(fc), B, self._ptr(c), st))
        if ft is not None and fc is not ft:
            ft.copy_(fc)
        if ft is not None:
            return self._ret(c, unb, is_np), f
        return self._ret(c, unb, is_np), self._ret(fc, unb, is_np)

    # ---- auxiliary (non-[B,n]) operands of the per-pass methods ---------------------------------
    def _aux(self, x, shape, unb, is_np, dev, dt, name):
        """Device tensor [B, *shape] for a per-pass operand.  Returns ``(work, origin)``: ``work`` is
        the contiguous device tensor handed to the kernel, ``origin`` the caller's object (tensor or
        writable ndarray) that an in-place pass must see updated, or None."""
        B = shape[0]
        if isinstance(x, torch.Tensor):
            if is_np or x.device != dev or x.dtype != dt:
                raise TypeError(f"{name} must share device and dtype with q")
            t = x[None] if unb else x
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{name} must have shape {list(shape[1:]) if unb else list(shape)}, got {list(x.shape)}")
            return (t if t.is_contiguous() else t.contiguous()), x
        if not is_np:
            raise TypeError("mixing numpy and torch inputs is not supported")
        xa = np.asarray(x, dtype=np.float64)
        xb = xa[None] if unb else xa
        if xb.shape != tuple(shape):
            raise ValueError(f"{name} must have shape {list(shape[1:]) if unb else list(shape)}, got {list(xa.shape)}")
        origin = x if isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.writeable else None
        return torch.as_tensor(np.ascontiguousarray(xb), device=dev), origin

    @staticmethod
    def _writeback(work, origin, unb):
        """Mirror the reference's in-place mutation of an argument (e.g. ``RBDReference.py:1291``)."""
        if origin is None:
            return
        src = work[0] if unb else work
        if isinstance(origin, torch.Tensor):
            if origin.data_ptr() != src.data_ptr():
                origin.copy_(src)
        else:
            np.copyto(origin, src.cpu().numpy())

    def rnea_grad_fpass_dq(self, q, qd, v, a, GRAVITY=-9.81):
        """RBDReference.rnea_grad_fpass_dq (``RBDReference.py:1127-1187``) -> ``(dv_dq, da_dq,
        df_dq)``, each ``(6, n, NB)`` per configuration; ``v, a`` are rnea's outputs."""
        (q, qd), unb, is_np, dev, dt = self._prep(q, qd)
        B, n = q.shape
        nb = self.n                        # bodies (== n unless the base floats: n = NB + 5)
        v, _ = self._aux(v, (B, 6, nb), unb, is_np, dev, dt, "v")
        a, _ = self._aux(a, (B, 6, nb), unb, is_np, dev, dt, "a")
        with torch.cuda.device(dev):
            dv = torch.empty((B, 6, n, nb), device=dev, dtype=dt)
            da = torch.empty_like(dv)
            df = torch.empty_like(dv)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_rnea_grad_fpass_dq", dt)(
                self._ptr(q), self._ptr(qd), self._ptr(v), self._ptr(a), float(GRAVITY), B,
                self._ptr(dv), self._ptr(da), self._ptr(df), st))
        return tuple(self._ret(t, unb, is_np) for t in (dv, da, df))

    def rnea_grad_fpass_dqd(self, q, qd, v):
        """RBDReference.rnea_grad_fpass_dqd (``RBDReference.py:1189-1255``) -> ``(dv_dqd, da_dqd,
        df_dqd)``, each ``(6, n, NB)`` per configuration."""
        (q, qd), unb, is_np, dev, dt = self._prep(q, qd)
        B, n = q.shape
        nb = self.n
        v, _ = self._aux(v, (B, 6, nb), unb, is_np, dev, dt, "v")
        with torch.cuda.device(dev):
            dv = torch.empty((B, 6, n, nb), device=dev, dtype=dt)
            da = torch.empty_like(dv)
            df = torch.empty_like(dv)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_rnea_grad_fpass_dqd", dt)(
                self._ptr(q), self._ptr(qd), self._ptr(v), B, self._ptr(dv), self._ptr(da), self._ptr(df), st))
        return tuple(self._ret(t, unb, is_np) for t in (dv, da, df))

    def rnea_grad_bpass_dq(self, q, f, df_dq):
        """RBDReference.rnea_grad_bpass_dq (``RBDReference.py:1257-1297``) -> ``dc_dq (n, n)``.  ``f`` is
        rnea's ACCUMULATED force; ``df_dq`` is accumulated child -> parent IN PLACE like the
        reference's argument (``:1291-1294``)."""
        (q,), unb, is_np, dev, dt = self._prep(q)
        B, n = q.shape
        f, _ = self._aux(f, (B, 6, self.n), unb, is_np, dev, dt, "f")
        df, origin = self._aux(df_dq, (B, 6, n, self.n), unb, is_np, dev, dt, "df_dq")
        with torch.cuda.device(dev):
            dc = torch.empty((B, n, n), device=dev, dtype=dt)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_rnea_grad_bpass_dq", dt)(
                self._ptr(q), self._ptr(f), self._ptr(df), B, self._ptr(dc), st))
        self._writeback(df, origin, unb)
        return self._ret(dc, unb, is_np)

    def rnea_grad_bpass_dqd(self, q, df_dqd, USE_VELOCITY_DAMPING=False):
        """RBDReference.rnea_grad_bpass_dqd (``RBDReference.py:1299-1343``) -> ``dc_dqd (n, n)``;
        ``df_dqd`` accumulated in place (``:1331``)."""
        (q,), unb, is_np, dev, dt = self._prep(q)
        B, n = q.shape
        df, origin = self._aux(df_dqd, (B, 6, n, self.n), unb, is_np, dev, dt, "df_dqd")
        with torch.cuda.device(dev):
            dc = torch.empty((B, n, n), device=dev, dtype=dt)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_rnea_grad_bpass_dqd", dt)(
                self._ptr(q), self._ptr(df), int(bool(USE_VELOCITY_DAMPING)), B, self._ptr(dc), st))
        self._writeback(df, origin, unb)
        return self._ret(dc, unb, is_np)

    def minv_bpass(self, q):
        """RBDReference.minv_bpass (``RBDReference.py:630-735``) -> ``(Minv, F, U, Dinv)`` with shapes
        ``(n, n), (n, 6, n), (n, 6), (n,)`` per configuration; ``Dinv`` holds ``D`` as there (``:698``)."""
        (q,), unb, is_np, dev, dt = self._prep(q)
        B, n = q.shape
        with torch.cuda.device(dev):
            Minv = torch.empty((B, n, n), device=dev, dtype=dt)
            F = torch.empty((B, n, 6, n), device=dev, dtype=dt)
            U = torch.empty((B, n, 6), device=dev, dtype=dt)
            D = torch.empty((B, n), device=dev, dtype=dt)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_minv_bpass", dt)(
                self._ptr(q), B, self._ptr(Minv), self._ptr(F), self._ptr(U), self._ptr(D), st))
        return tuple(self._ret(t, unb, is_np) for t in (Minv, F, U, D))

    def minv_fpass(self, q, Minv, F, U, Dinv):
        """RBDReference.minv_fpass (``RBDReference.py:737-783``) -> ``Minv`` (updated in place, whole
        rows as at ``:771``: valid upper triangle, by-products below it).  ``F`` is rebuilt in place
        (``:774-781``); its incoming values are not read."""
        (q,), unb, is_np, dev, dt = self._prep(q)
        B, n = q.shape
        Mw, Mo = self._aux(Minv, (B, n, n), unb, is_np, dev, dt, "Minv")
        Fw, Fo = self._aux(F, (B, n, 6, n), unb, is_np, dev, dt, "F")
        Uw, _ = self._aux(U, (B, n, 6), unb, is_np, dev, dt, "U")
        Dw, _ = self._aux(Dinv, (B, n), unb, is_np, dev, dt, "Dinv")
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_minv_fpass", dt)(
                self._ptr(q), B, self._ptr(Mw), self._ptr(Fw), self._ptr(Uw), self._ptr(Dw), st))
        self._writeback(Mw, Mo, unb)
        self._writeback(Fw, Fo, unb)
        if isinstance(Mo, torch.Tensor):
            return Minv
        return self._ret(Mw, unb, is_np)

    def rnea_grad(self, q, qd, qdd=None, GRAVITY=-9.81, USE_VELOCITY_DAMPING=False,
                  return_c: bool = False, out=None):
        """RBDReference.rnea_grad (``RBDReference.py:1345-1368``) -> ``dc_du = [dc_dq | dc_dqd]``,
        ``(n, 2n)`` per configuration.  ``return_c=True`` also returns the bias force ``c`` the
        reference computes internally (``:1353``) -> ``(c, dc_du)``.  Floating base: ``n = NB + 5`` and the
        base's six position columns are derivatives along a base-frame twist, as in the reference; robots
        with fewer than six bodies are refused (the reference raises IndexError for them, ``:1168``)."""
        if out is None:
            sg = self._sig(q, qd, qdd)
            if sg is not None:
                B, dt, idx, st = sg
                hq = qdd is not None
                p = self._plan(("rnea_grad", B, dt, idx, st, hq, return_c),
                               lambda: self._mk_plan("rbd_rnea_grad", dt, idx, ((B, self.nv), (B, self.nv, 2 * self.nv)) if return_c else ((B, self.nv, 2 * self.nv),), hq))
                if p is not None:
                    o = p.take()
                    rc = p.fn(q.data_ptr(), qd.data_ptr(), qdd.data_ptr() if hq else None, float(GRAVITY), 1 if USE_VELOCITY_DAMPING else 0, B,
                              o.ptrs[0] if return_c else None, o.ptrs[-1], st)
                    if rc != 0:
                        self._lib.check(rc)
                    return o.outs if return_c else o.outs[0]
        if out is not None:
            sg = self._sig(q, qd, qdd)
            if sg is not None:
                B, dt, idx, st = sg
                hq = qdd is not None
                oc, odc = (out if (return_c and type(out) in (tuple, list) and len(out) == 2) else (None, out))
                if self._out_ok(odc, (B, self.nv, 2 * self.nv), dt, idx) and (not return_c or self._out_ok(oc, (B, self.nv), dt, idx)):
                    p = self._plan(("rnea_grad", B, dt, idx, st, hq, return_c),
                                   lambda: self._mk_plan("rbd_rnea_grad", dt, idx, ((B, self.nv), (B, self.nv, 2 * self.nv)) if return_c else ((B, self.nv, 2 * self.nv),), hq))
                    if p is not None:
                        rc = p.fn(q.data_ptr(), qd.data_ptr(), qdd.data_ptr() if hq else None, float(GRAVITY), 1 if USE_VELOCITY_DAMPING else 0, B,
                                  oc.data_ptr() if return_c else None, odc.data_ptr(), st)
                        if rc != 0:
                            self._lib.check(rc)
                        return (oc, odc) if return_c else odc
        (q, qd, qdd), unb, is_np, dev, dt = self._prep(q, qd, qdd)
        B = q.shape[0]
        if out is not None:      # out=dc_du, or out=(c, dc_du) with return_c=True: pre-allocated device tensors
            if is_np or unb:
                raise ValueError("out= needs batched tensor inputs")
            oc, odc = (out if return_c else (None, out))
            dc = self._check_out(odc, (B, self.nv, 2 * self.nv), dev, dt, "dc_du")
            c = self._check_out(oc, (B, self.nv), dev, dt, "c") if return_c else None
            with torch.cuda.device(dev):
                self._lib.check(self._fn("rbd_rnea_grad", dt, qdd is not None)(
                    self._ptr(q), self._ptr(qd), self._ptr(qdd), float(GRAVITY), 1 if USE_VELOCITY_DAMPING else 0, B,
                    self._ptr(c), self._ptr(dc), torch.cuda.current_stream(dev).cuda_stream))
            return (c, dc) if return_c else dc
        with torch.cuda.device(dev):
            dc = torch.empty((B, self.nv, 2 * self.nv), device=dev, dtype=dt)
            c = torch.empty((B, self.nv), device=dev, dtype=dt) if return_c else None
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_rnea_grad", dt, qdd is not None)(
                self._ptr(q), self._ptr(qd), self._ptr(qdd), float(GRAVITY),
                1 if USE_VELOCITY_DAMPING else 0, B, self._ptr(c), self._ptr(dc), st))
        if return_c:
            return self._ret(c, unb, is_np), self._ret(dc, unb, is_np)
        return self._ret(dc, unb, is_np)

    def rnea_and_grad(self, q, qd, qdd=None, GRAVITY=-9.81, USE_VELOCITY_DAMPING=False):
        """``rnea`` and ``rnea_grad`` of the same inputs in one call -> ``(c, v, a, f, dc_du)``: what the
        reference computes inside ``rnea_grad`` (``RBDReference.py:1353`` runs ``rnea`` and drops its
        outputs).  For small batches this is one launch of the column kernel."""
        sg = self._sig(q, qd, qdd)
        if sg is not None:
            B, dt, idx, st = sg
            p = self._plan(("rnea_and_grad", B, dt, idx, st), lambda: self._mk_plan(
                "rbd_rnea_with_grad", dt, idx, ((B, self.nv), (B, 6, self.n), (B, 6, self.n), (B, 6, self.n), (B, self.nv, 2 * self.nv))))
            if p is not None:
                o = p.take()
                rc = p.fn(q.data_ptr(), qd.data_ptr(), None if qdd is None else qdd.data_ptr(), float(GRAVITY), 1 if USE_VELOCITY_DAMPING else 0, B, *o.ptrs, st)
                if rc != 0:
                    self._lib.check(rc)
                return o.outs
        (q, qd, qdd), unb, is_np, dev, dt = self._prep(q, qd, qdd)
        B = q.shape[0]
        with torch.cuda.device(dev):
            c = torch.empty((B, self.nv), device=dev, dtype=dt)
            v = torch.empty((B, 6, self.n), device=dev, dtype=dt)
            a = torch.empty_like(v)
            f = torch.empty_like(v)
            dc = torch.empty((B, self.nv, 2 * self.nv), device=dev, dtype=dt)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_rnea_with_grad", dt)(
                self._ptr(q), self._ptr(qd), self._ptr(qdd), float(GRAVITY), 1 if USE_VELOCITY_DAMPING else 0, B,
                self._ptr(c), self._ptr(v), self._ptr(a), self._ptr(f), self._ptr(dc), st))
        return tuple(self._ret(t, unb, is_np) for t in (c, v, a, f, dc))

    def minv(self, q, output_dense=True, out=None, workspace=None):
        """RBDReference.minv (``RBDReference.py:785-806``) -> ``(n, n)`` per configuration.
        ``output_dense=False`` returns the upper triangle with a ZERO strict lower triangle (the
        reference leaves forward-pass by-products there, ``:771``).  ``out=`` / ``workspace=``: a pre-allocated
        ``[B, n, n]`` result tensor and a uint8 scratch tensor of ``minv_workspace_bytes(B, dtype)`` bytes
        (batched tensor input only), for loops that own their buffers."""
        if out is None and workspace is None:
            sg = self._sig(q, None, None)
            if sg is not None:
                B, dt, idx, st = sg
                p = self._plan(("minv", B, dt, idx, st), lambda: self._mk_plan(
                    "rbd_minv", dt, idx, ((B, self.nv, self.nv),), ws_query=lambda lib, esz: lib.rbd_minv_workspace_bytes(B, esz)))
                if p is not None:
                    o = p.take()
                    rc = p.fn(q.data_ptr(), B, 1 if output_dense else 0, o.ptrs[0], p.wsp if p.wsb else None, p.wsb, st)
                    if rc != 0:
                        self._lib.check(rc)
                    return o.outs[0]
        if out is not None and workspace is None:
            sg = self._sig(q, None, None)
            if sg is not None:
                B, dt, idx, st = sg
                if self._out_ok(out, (B, self.nv, self.nv), dt, idx):
                    p = self._plan(("minv", B, dt, idx, st), lambda: self._mk_plan(
                        "rbd_minv", dt, idx, ((B, self.nv, self.nv),), ws_query=lambda lib, esz: lib.rbd_minv_workspace_bytes(B, esz)))
                    if p is not None:
                        rc = p.fn(q.data_ptr(), B, 1 if output_dense else 0, out.data_ptr(), p.wsp if p.wsb else None, p.wsb, st)
                        if rc != 0:
                            self._lib.check(rc)
                        return out
        (q,), unb, is_np, dev, dt = self._prep(q)
        B = q.shape[0]
        esz = 4 if dt == torch.float32 else 8
        if out is not None and (is_np or unb):
            raise ValueError("out= needs a batched tensor input")
        with torch.cuda.device(dev):
            M = torch.empty((B, self.nv, self.nv), device=dev, dtype=dt) if out is None else \
                self._check_out(out, (B, self.nv, self.nv), dev, dt, "Minv")
            sfx = "f32" if esz == 4 else "f64"
            lib = self._lib.resolve("rbd_minv", sfx)            # ONE resolution: size and entry point from the same library
            wsb = int(lib.rbd_minv_workspace_bytes(B, esz))
            if workspace is not None:
                if workspace.dtype != torch.uint8 or workspace.device != dev or workspace.numel() < wsb or not workspace.is_contiguous():
                    raise ValueError(f"workspace: a contiguous uint8 tensor of >= {wsb} bytes on {dev}")
                ws = workspace
            else:
                ws = torch.empty((wsb,), device=dev, dtype=torch.uint8) if wsb > 0 else None   # most robots: no scratch at all
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(getattr(lib, f"rbd_minv_{sfx}")(
                self._ptr(q), B, 1 if output_dense else 0, self._ptr(M), ws.data_ptr() if ws is not None else None, wsb, st))
        return self._ret(M, unb, is_np)

    # ---- pre-resolved launches for loops that own their buffers --------------------------------------------------
    def bind(self, op: str, q, qd=None, qdd=None, GRAVITY=-9.81, USE_VELOCITY_DAMPING=False, output_dense=True):
        """A control loop calls the same entry point on the same buffers thousands of times; a call through the
        methods above costs 15-25 us of Python (shape checks, allocations, device guard) next to a 7 us kernel at
        B = 4096.  ``bind`` validates once, allocates the outputs once, and returns a `BoundLaunch`: calling it is ONE
        ctypes call into the C-ABI on torch's current stream (the caller rewrites ``q, qd, qdd`` in place between calls;
        capturable in a ``torch.cuda.graph``).  ``op``: 'rnea' -> outputs (c, v, a, f); 'rnea_grad' -> (c, dc_du);
        'rnea_and_grad' -> (c, v, a, f, dc_du); 'minv' -> (Minv,).  Batched tensor inputs only."""
        ins = (q,) if op == "minv" else (q, qd, qdd)
        if op != "minv" and (qd is None):
            raise ValueError("bind: qd is required")
        tens, unb, is_np, dev, dt = self._prep(*ins)
        if is_np or unb:
            raise ValueError("bind needs batched torch tensors (the bound launch reads them in place)")
        for given, prepared in zip(ins, tens):
            if given is not None and prepared.data_ptr() != given.data_ptr():
                raise ValueError("bind: inputs must be contiguous (the bound launch reads the caller's own buffers)")
        B = tens[0].shape[0]
        esz = 4 if dt == torch.float32 else 8
        E = lambda *shape: torch.empty(shape, device=dev, dtype=dt)   # noqa: E731
        with torch.cuda.device(dev):
            g, damp = float(GRAVITY), 1 if USE_VELOCITY_DAMPING else 0
            p = [self._ptr(t) for t in tens]
            if op == "rnea":
                outs = (E(B, self.nv), E(B, 6, self.n), E(B, 6, self.n), E(B, 6, self.n))
                fn = self._fn("rbd_rnea", dt)
                args = lambda st: (p[0], p[1], p[2], g, B, *[o.data_ptr() for o in outs], st)   # noqa: E731
            elif op == "rnea_grad":
                outs = (E(B, self.nv), E(B, self.nv, 2 * self.nv))
                fn = self._fn("rbd_rnea_grad", dt, tens[2] is not None)
                args = lambda st: (p[0], p[1], p[2], g, damp, B, outs[0].data_ptr(), outs[1].data_ptr(), st)   # noqa: E731
            elif op == "rnea_and_grad":
                outs = (E(B, self.nv), E(B, 6, self.n), E(B, 6, self.n), E(B, 6, self.n), E(B, self.nv, 2 * self.nv))
                fn = self._fn("rbd_rnea_with_grad", dt)
                args = lambda st: (p[0], p[1], p[2], g, damp, B, *[o.data_ptr() for o in outs], st)   # noqa: E731
            elif op == "minv":
                outs = (E(B, self.nv, self.nv),)
                lib = self._lib.resolve("rbd_minv", "f32" if esz == 4 else "f64")
                wsb = int(lib.rbd_minv_workspace_bytes(B, esz))
                ws = torch.empty((wsb,), device=dev, dtype=torch.uint8) if wsb > 0 else None
                fn = getattr(lib, "rbd_minv_f32" if esz == 4 else "rbd_minv_f64")
                wp = ws.data_ptr() if ws is not None else None
                args = lambda st: (p[0], B, 1 if output_dense else 0, outs[0].data_ptr(), wp, wsb, st)   # noqa: E731
                outs = outs + ((ws,) if ws is not None else ())      # (kept alive with the launch; not a result)
            else:
                raise ValueError("bind: op must be 'rnea', 'rnea_grad', 'rnea_and_grad' or 'minv'")
        n_res = len(outs) - (1 if (op == "minv" and len(outs) == 2) else 0)
        return BoundLaunch(self, fn, args, tens, outs[:n_res], outs, dev)

    def minv_workspace_bytes(self, B: int, dtype=torch.float32) -> int:
        """Scratch bytes ``minv`` needs for B rows (0 for robots served by a kernel without workspace)."""
        return int(self._minv_ws(int(B), 4 if dtype == torch.float32 else 8))

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
    def aba(self, q, qd, tau, f_ext=[], GRAVITY=-9.81):
        """RBDReference.aba (fixed-base branch, ``RBDReference.py:940-1024``) -> ``qdd``; ``f_ext`` is
        accepted and ignored exactly as that branch does."""
        (q, qd, tau), unb, is_np, dev, dt = self._prep(q, qd, tau)
        B = q.shape[0]
        with torch.cuda.device(dev):
            qdd = torch.empty((B, self.n), device=dev, dtype=dt)
            st = torch.cuda.current_stream(dev).cuda_stream
            self._lib.check(self._fn("rbd_aba", dt)(
                self._ptr(q), self._ptr(qd), self._ptr(tau), float(GRAVITY), B, self._ptr(qdd), st))
        return self._ret(qdd, unb, is_np)

    def _fd(self, q, qd, u, GRAVITY, want_grad):
        sg = self._sig(q, qd, u) if u is not None else None
        if sg is not None:
            B, dt, idx, st = sg
            base = "rbd_forward_dynamics_grad" if want_grad else "rbd_forward_dynamics"
            shapes = ((B, self.nv), (B, self.nv, 2 * self.nv)) if want_grad else ((B, self.nv),)
            p = self._plan((base, B, dt, idx, st), lambda: self._mk_plan(base, dt, idx, shapes, ws_query=lambda lib, esz: lib.rbd_fd_workspace_bytes(B, esz)))
            if p is not None:
                o = p.take()
                if want_grad:
                    rc = p.fn(q.data_ptr(), qd.data_ptr(), u.data_ptr(), float(GRAVITY), B, o.ptrs[0], o.ptrs[1], p.wsp, p.wsb, st)
                else:
                    rc = p.fn(q.data_ptr(), qd.data_ptr(), u.data_ptr(), float(GRAVITY), B, o.ptrs[0], p.wsp, p.wsb, st)
                if rc != 0:
                    self._lib.check(rc)
                return o.outs[0], (o.outs[1] if want_grad else None), False, False
        (q, qd, u), unb, is_np, dev, dt = self._prep(q, qd, u)
        B = q.shape[0]
        esz = 4 if dt == torch.float32 else 8
        sfx = "f32" if esz == 4 else "f64"
        with torch.cuda.device(dev):
            qdd = torch.empty((B, self.nv), device=dev, dtype=dt)
            st = torch.cuda.current_stream(dev).cuda_stream
            base = "rbd_forward_dynamics_grad" if want_grad else "rbd_forward_dynamics"
            lib = self._lib.resolve(base, sfx)              # ONE resolution: workspace size, branch and entry point from the same library
            generic = getattr(lib, "is_generic", False)
            fn = getattr(lib, f"{base}_{sfx}")
            if not want_grad and (self.model.floating or generic):   # rnea (bias force) + minv + one product: scratch for c and Minv
                wsb = int(lib.rbd_fd_workspace_bytes(B, esz))
                ws = torch.empty((max(wsb, 1),), device=dev, dtype=torch.uint8)
                self._lib.check(fn(self._ptr(q), self._ptr(qd), self._ptr(u), float(GRAVITY), B, self._ptr(qdd), ws.data_ptr(), wsb, st))
                return qdd, None, unb, is_np
            if not want_grad:        # one articulated-body launch, no scratch
                self._lib.check(fn(self._ptr(q), self._ptr(qd), self._ptr(u), float(GRAVITY), B, self._ptr(qdd), None, 0, st))
                return qdd, None, unb, is_np
            wsb = int(lib.rbd_fd_workspace_bytes(B, esz))
            ws = torch.empty((max(wsb, 1),), device=dev, dtype=torch.uint8)
            d = torch.empty((B, self.nv, 2 * self.nv), device=dev, dtype=dt)
            self._lib.check(fn(self._ptr(q), self._ptr(qd), self._ptr(u), float(GRAVITY), B, self._ptr(qdd),
                               self._ptr(d), ws.data_ptr(), wsb, st))
            return qdd, d, unb, is_np

    def forward_dynamics(self, q, qd, u, GRAVITY=-9.81):
        """RBDReference.forward_dynamics (``RBDReference.py:1371-1374``): ``minv(q) @ (u - c(q, qd))``,
        evaluated by the articulated-body sweeps (same value to rounding, neither factor is formed)."""
        qdd, _, unb, is_np = self._fd(q, qd, u, GRAVITY, False)
        return self._ret(qdd, unb, is_np)

    def forward_dynamics_grad(self, q, qd, u, GRAVITY=-9.81):
        """RBDReference.forward_dynamics_grad (``RBDReference.py:1376-1384``) -> ``(qdd_dq, qdd_dqd)``,
        each ``(n, n)`` per configuration (views of one ``[B, n, 2n]`` buffer for tensor inputs)."""
        _, d, unb, is_np = self._fd(q, qd, u, GRAVITY, True)
        n = self.nv
        d = self._ret(d, unb, is_np)
        return d[..., :n], d[..., n:]
