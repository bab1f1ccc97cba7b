"""ctypes binding of the per-robot C-ABI library (include/rbd_hip.h).

There is NO fallback: if the library for a robot cannot be found or built, or exports the wrong
model, this module raises.  The oracle under /oracle is never imported from here.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, c_char, c_char_p, c_double, c_float, c_int, c_int32,
                    c_int64, c_size_t, c_uint64, c_void_p)

from .build import build_model, lib_path
from .packer import ABI_VERSION, PackedModel

RBD_MAX_BODIES = 64
RBD_ERR_ARG, RBD_ERR_UNSUPPORTED, RBD_ERR_WORKSPACE = -1, -2, -3
# rbd_set_option / rbd_kernel_name constants (include/rbd_hip.h)
RBD_OPT_GRAD_KERNEL, RBD_OPT_MINV_PHASE_A, RBD_OPT_RNEA_KERNEL, RBD_OPT_SELECT_BATCH = 0, 1, 2, 3
RBD_RNEA_KERNEL_AUTO, RBD_RNEA_KERNEL_BATCH, RBD_RNEA_KERNEL_GROUPS = 0, 1, 2
RBD_GRAD_KERNEL_AUTO, RBD_GRAD_KERNEL_TREE, RBD_GRAD_KERNEL_COLS, RBD_GRAD_KERNEL_BATCH = 0, 1, 2, 3
RBD_MINV_PHASE_A_AUTO, RBD_MINV_PHASE_A_LANE, RBD_MINV_PHASE_A_IA8, RBD_MINV_PHASE_A_FUSED = 0, 1, 2, 3
RBD_OP_RNEA, RBD_OP_RNEA_GRAD, RBD_OP_MINV = 0, 1, 2

# every symbol include/rbd_hip.h declares (tests check the built library exports all of them)
EXPORTED_SYMBOLS = [
    "rbd_abi_version", "rbd_last_error", "rbd_model_info", "rbd_release_workspaces",
    "rbd_set_option", "rbd_get_option", "rbd_kernel_name",
    "rbd_rnea_f32", "rbd_rnea_f64", "rbd_rnea_grad_f32", "rbd_rnea_grad_f64",
    "rbd_rnea_with_grad_f32", "rbd_rnea_with_grad_f64",
    "rbd_rnea_fpass_f32", "rbd_rnea_fpass_f64", "rbd_rnea_bpass_f32", "rbd_rnea_bpass_f64",
    "rbd_minv_workspace_bytes", "rbd_minv_f32", "rbd_minv_f64",
    "rbd_crba_f32", "rbd_crba_f64",
    "rbd_fd_workspace_bytes", "rbd_forward_dynamics_f32", "rbd_forward_dynamics_f64",
    "rbd_forward_dynamics_grad_f32", "rbd_forward_dynamics_grad_f64",
    "rbd_rnea_grad_fpass_dq_f32", "rbd_rnea_grad_fpass_dq_f64",
    "rbd_rnea_grad_fpass_dqd_f32", "rbd_rnea_grad_fpass_dqd_f64",
    "rbd_rnea_grad_bpass_dq_f32", "rbd_rnea_grad_bpass_dq_f64",
    "rbd_rnea_grad_bpass_dqd_f32", "rbd_rnea_grad_bpass_dqd_f64",
    "rbd_aba_f32", "rbd_aba_f64",
    "rbd_minv_bpass_f32", "rbd_minv_bpass_f64", "rbd_minv_fpass_f32", "rbd_minv_fpass_f64",
]


class RbdModelInfo(Structure):
    _fields_ = [("abi_version", c_int32), ("n", c_int32), ("max_depth", c_int32),
                ("hash", c_uint64), ("name", c_char * 64),
                ("parent", c_int32 * RBD_MAX_BODIES), ("joint_type", c_int32 * RBD_MAX_BODIES),
                ("joint_axis", c_int32 * RBD_MAX_BODIES), ("floating_base", c_int32), ("nv", c_int32)]


class RbdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"rbd C-ABI error {code}: {msg}")
        self.code = code


def _declare(lib):
    lib.rbd_abi_version.restype = c_int
    lib.rbd_abi_version.argtypes = []
    lib.rbd_last_error.restype = c_char_p
    lib.rbd_last_error.argtypes = []
    lib.rbd_model_info.restype = c_int
    lib.rbd_model_info.argtypes = [POINTER(RbdModelInfo)]
    lib.rbd_release_workspaces.restype = c_int
    lib.rbd_release_workspaces.argtypes = []
    lib.rbd_set_option.restype = c_int
    lib.rbd_set_option.argtypes = [c_int, c_int]
    lib.rbd_get_option.restype = c_int
    lib.rbd_get_option.argtypes = [c_int]
    lib.rbd_kernel_name.restype = c_int
    lib.rbd_kernel_name.argtypes = [c_int, c_int, c_int64, c_char_p, c_size_t]
    for sfx, ft in (("f32", c_float), ("f64", c_double)):
        f = getattr(lib, f"rbd_rnea_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, ft, c_int64, c_void_p, c_void_p, c_void_p,
                      c_void_p, c_void_p]
        f = getattr(lib, f"rbd_rnea_fpass_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, ft, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]
        f = getattr(lib, f"rbd_rnea_bpass_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]
        f = getattr(lib, f"rbd_rnea_grad_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, ft, c_int, c_int64, c_void_p, c_void_p, c_void_p]
        f = getattr(lib, f"rbd_rnea_with_grad_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, ft, c_int, c_int64] + [c_void_p] * 6
        f = getattr(lib, f"rbd_minv_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_size_t, c_void_p]
        f = getattr(lib, f"rbd_crba_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_int64, c_void_p, c_void_p]
        f = getattr(lib, f"rbd_forward_dynamics_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, ft, c_int64, c_void_p, c_void_p, c_size_t, c_void_p]
        f = getattr(lib, f"rbd_forward_dynamics_grad_{sfx}")
        f.restype = c_int
        f.argtypes = [c_void_p, c_void_p, c_void_p, ft, c_int64, c_void_p, c_void_p, c_void_p, c_size_t,
                      c_void_p]
        for nm, at in (
                ("rbd_rnea_grad_fpass_dq", [c_void_p] * 4 + [ft, c_int64] + [c_void_p] * 4),
                ("rbd_rnea_grad_fpass_dqd", [c_void_p] * 3 + [c_int64] + [c_void_p] * 4),
                ("rbd_rnea_grad_bpass_dq", [c_void_p] * 3 + [c_int64] + [c_void_p] * 2),
                ("rbd_rnea_grad_bpass_dqd", [c_void_p] * 2 + [c_int, c_int64] + [c_void_p] * 2),
                ("rbd_aba", [c_void_p] * 3 + [ft, c_int64, c_void_p, c_void_p]),
                ("rbd_minv_bpass", [c_void_p, c_int64] + [c_void_p] * 5),
                ("rbd_minv_fpass", [c_void_p, c_int64] + [c_void_p] * 5)):
            f = getattr(lib, f"{nm}_{sfx}")
            f.restype = c_int
            f.argtypes = at
    lib.rbd_minv_workspace_bytes.restype = c_size_t
    lib.rbd_minv_workspace_bytes.argtypes = [c_int64, c_int]
    lib.rbd_fd_workspace_bytes.restype = c_size_t
    lib.rbd_fd_workspace_bytes.argtypes = [c_int64, c_int]


class RbdLibrary:
    """The loaded per-robot library, checked against the packed model it is meant to serve.

    First use of a robot (no up-to-date library on disk, ``build=True``): the full library is built in a background
    thread while calls are served from small FAMILY libraries built on demand (``build.build_family``: COMMON + the
    units of one family of entry points in one precision + stubs, 4-15 s each for a 7-DoF robot instead of the
    whole library's 40 s; minutes for a 30-body robot); as soon as the full library is ready every call goes to it.
    ``RBD_LAZY_BUILD=0`` (or ``lazy=False``) restores the blocking build.  ``.lib`` is always the FULL library (it
    waits for the background build if there is one).  Options are process-wide per library file, so they are
    remembered here and applied to every library of the robot as it is loaded.

    ``generic`` ('auto' | 'only' | 'never'; default from ``RBD_GENERIC``, else 'auto'): the MODEL-HANDLE library
    (include/rbd_generic.h, ``generic.GenericModel``) serves rnea / rnea_grad / minv / forward_dynamics(_grad) of a
    fixed- or floating-base robot with no per-robot compilation at all.  'auto': it answers while the robot's own library is being
    built in the background (first call after ``RBDReference(robot)`` returns in milliseconds instead of after a
    family build) and keeps answering if that build cannot happen (no hipcc on the machine); the other entry points
    still wait for their family library.  'only': nothing is built or loaded per robot.  'never': as before.
    The generic and the specialised kernels agree to rounding, not bit for bit; ``ShardedRBD`` therefore waits for the
    full library before it computes (dist.py)."""

    def __init__(self, model: PackedModel, build: bool = True, lazy=None, generic=None):
        import threading
        from .build import full_library_ready
        self.model = model
        self.path = lib_path(model)
        self._full = None
        self._fams = {}
        self._bg = None
        self._bg_err = None
        self._opts = {}
        self.epoch = 0                          # bumped by set_option: cached launch plans (api.py) are per epoch
        self._lock = threading.RLock()
        self._tls = threading.local()
        if lazy is None:
            lazy = os.environ.get("RBD_LAZY_BUILD", "1") != "0"
        if generic is None:
            generic = {"0": "never", "1": "auto", "": "auto"}.get(os.environ.get("RBD_GENERIC", "auto"), os.environ.get("RBD_GENERIC", "auto"))
        if generic not in ("auto", "only", "never"):
            raise ValueError("generic must be 'auto', 'only' or 'never'")
        self._generic_mode = generic
        self._generic = None
        if self._generic_mode == "only":
            from .generic import GenericModel
            self._generic = GenericModel(model, build=build)       # raises if the library is missing and cannot be built
            return
        if build and lazy and (not model.floating or generic != "never") and not full_library_ready(model):
            def work():
                try:
                    build_model(model)
                except BaseException as e:      # noqa: BLE001  (re-raised in the caller's thread by .lib)
                    self._bg_err = e
            self._bg = threading.Thread(target=work, name=f"rbd-build-{model.name}", daemon=True)
            self._bg.start()
            return
        if build:
            self.path = build_model(model)            # no-op when up to date; raises if hipcc is missing
        if not os.path.exists(self.path):
            raise FileNotFoundError(
                f"HIP library for robot {model.name!r} (hash {model.hash}) not found at {self.path}; "
                "build it with rbdreference_amd.build.build_model() -- there is no CPU fallback")
        self._full = self._load(self.path)

    # ---- loading ------------------------------------------------------------------------------------
    def _load(self, path: str):
        model = self.model
        lib = ctypes.CDLL(path)
        _declare(lib)
        if lib.rbd_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{path}: ABI {lib.rbd_abi_version()} != {ABI_VERSION}")
        info = RbdModelInfo()
        rc = lib.rbd_model_info(ctypes.byref(info))
        if rc != 0:
            raise RbdError(rc, (lib.rbd_last_error() or b"").decode())
        if info.n != model.n or f"{info.hash:016x}" != model.hash or \
                list(info.parent[:model.n]) != list(model.parent) or info.nv != model.nv or \
                bool(info.floating_base) != bool(model.floating):
            raise RuntimeError(f"{path}: compiled-in model does not match robot {model.name!r}")
        for o, v in self._opts.items():
            lib.rbd_set_option(o, v)
        self._info = info
        return lib

    def _full_if_ready(self):
        if self._full is not None:
            return self._full
        if self._generic_mode == "only":
            return None
        if self._bg is not None and not self._bg.is_alive():
            if self._bg_err is not None and self._generic_serving() is not None:
                return None                     # no per-robot library on this machine: the generic one keeps serving
            return self.lib
        return None

    def _generic_serving(self):
        """The GenericModel of this robot, or None (mode 'never', floating base, or its library cannot be had)."""
        if self._generic_mode == "never":
            return None
        if self._generic is None:
            with self._lock:
                if self._generic is None:
                    try:
                        from .generic import GenericModel
                        self._generic = GenericModel(self.model, build=True)
                    except Exception:           # noqa: BLE001  (no library and no compiler: the family path decides)
                        self._generic_mode = "never"
                        return None
        return self._generic

    def wait_specialized(self):
        """Block until the robot's own full library is loaded; returns self.  No-op in mode 'only', and where the
        background build failed (no compiler on this machine) while the model-handle library serves the robot: every
        call then stays on that one library, which is all `ShardedRBD` needs (no mixing of kernels between ranks)."""
        if self._generic_mode == "only":
            return self
        if self._bg is not None:
            self._bg.join()
        if self._full is None and self._bg_err is not None and self._generic_serving() is not None:
            return self
        self.lib
        return self

    def serving(self, base: str, sfx: str, has_qdd: bool = True):
        """The library object (full, family or generic) that answers entry point `base` right now."""
        from .build import family_of
        lib = self._full_if_ready()
        if lib is not None:
            return lib
        fam = (family_of(base, has_qdd), sfx or "f32")
        if fam in self._fams:
            return self._fams[fam]
        g = self._generic_serving()
        if g is not None and g.serves(base):
            return g
        if self._generic_mode == "only":
            raise RbdError(RBD_ERR_UNSUPPORTED, f"{base}: not served by the model-handle library (generic='only')")
        if self._bg_err is not None:
            raise self._bg_err
        return self._family(*fam)

    @property
    def lib(self):
        """The FULL library (blocks until a background build has finished)."""
        if self._generic_mode == "only":
            raise RuntimeError("generic='only': no per-robot library is loaded")
        with self._lock:
            if self._full is None:
                if self._bg is not None:
                    self._bg.join()
                if self._bg_err is not None:
                    raise self._bg_err
                self._full = self._load(self.path)
            return self._full

    @property
    def info(self):
        if getattr(self, "_info", None) is None:
            self.lib
        return self._info

    def _family(self, family: str, sfx: str):
        from .build import build_family, family_lib_path
        key = (family, sfx)
        with self._lock:
            lib = self._fams.get(key)
            if lib is None:
                lib = self._load(build_family(self.model, family, sfx))
                self._fams[key] = lib
            return lib

    def fn(self, base: str, sfx: str, has_qdd: bool = True):
        """C entry point ``<base>_<sfx>`` (sfx 'f32' | 'f64'; '' for the suffix-less ones): from the full library when
        it is ready, else from the family library that serves it (built now if need be)."""
        lib = self.serving(base, sfx, has_qdd)
        self._tls.lib = lib
        return getattr(lib, f"{base}_{sfx}" if sfx else base)

    def resolve(self, base: str, sfx: str, has_qdd: bool = True):
        """The library object that answers ``base`` NOW, resolved ONCE: a call that needs several things of it (the
        workspace size AND the entry point) must take them from this one object -- if the background build finishes
        between two separate look-ups, the size would come from one library and the launch from another."""
        lib = self.serving(base, sfx, has_qdd)
        self._tls.lib = lib
        return lib

    def served_by_generic(self) -> bool:
        """True if the calling thread's last ``fn()`` / ``serving()`` answer was the model-handle library."""
        return bool(getattr(getattr(self._tls, "lib", None), "is_generic", False))

    def check(self, rc: int):
        if rc != 0:
            lib = getattr(self._tls, "lib", None) or self._full_if_ready() or next(iter(self._fams.values()), None) or self._generic or self.lib
            raise RbdError(rc, (lib.rbd_last_error() or b"").decode())

    def _loaded(self):
        return ([self._full] if self._full is not None else []) + list(self._fams.values())

    def set_option(self, option: int, value: int) -> None:
        with self._lock:
            libs = self._loaded()
            if not libs and self._generic_serving() is None:
                libs = [self.lib]
            for lib in libs:                     # (the model-handle library has one kernel per entry point: nothing to select)
                self._tls.lib = lib
                self.check(lib.rbd_set_option(option, value))
            self._opts[option] = value
            self.epoch += 1

    @property
    def stable(self) -> bool:
        """True once the robot's own FULL library is loaded: from then on every entry point resolves to the same
        function object for good (api.py caches its launch plans only in this state)."""
        return self._full is not None

    def get_option(self, option: int) -> int:
        with self._lock:
            libs = self._loaded()
            if not libs and self._generic_serving() is not None:
                return int(self._opts.get(option, 0))
            return int((libs or [self.lib])[0].rbd_get_option(option))

    def kernel_name(self, op: int, elem_size: int, B: int) -> str:
        """Name of the (dominant) kernel entry point `op` launches for B rows (host-side, no GPU)."""
        base = {RBD_OP_RNEA: "rbd_rnea", RBD_OP_RNEA_GRAD: "rbd_rnea_grad", RBD_OP_MINV: "rbd_minv"}[op]
        lib = self.serving(base, "f32" if elem_size == 4 else "f64")
        if getattr(lib, "is_generic", False):
            return lib.kernel_name(op, elem_size)
        self._tls.lib = lib
        buf = ctypes.create_string_buffer(128)
        self.check(lib.rbd_kernel_name(op, elem_size, B, buf, len(buf)))
        return buf.value.decode()
