"""Per-robot build of the HIP library: packed model -> generated header -> hipcc (gfx950) -> .so.

Libraries land IN-TREE under ``rbdreference_amd/_build/`` (git-ignored, but they travel with the
gpurun snapshot), named ``librbd_<name>_<hash>.so``; the hash covers every model constant, so a
stale library can never be picked up for a changed robot.  ``python -m rbdreference_amd.build``
prebuilds the three built-in robots (what ``__graft_entry__.build()`` calls).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import threading
import time
from typing import Optional

from .packer import PackedModel, emit_header, pack_robot, safe_name

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD_DIR = os.path.join(HERE, "_build")
ARCH = "gfx950"

HIPCC_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-fno-slp-vectorize",         # packed-f32 pairs need aligned VGPR tuples: +70 VGPRs and scratch spills here
    "-fno-signed-zeros",          # lets structural zeros fold; the sign of a zero is never observable here
    "-munsafe-fp-atomics",        # LDS float adds lower to ds_add_f32 / ds_add_f64, never CAS loops
    "-Wno-unused-value",
]


def hipcc_path() -> str:
    p = os.environ.get("RBD_HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found (set RBD_HIPCC); the per-robot HIP library cannot be built")
    return p


def lib_path(m: PackedModel) -> str:
    return os.path.join(BUILD_DIR, f"librbd_{safe_name(m.name)}_{m.hash}.so")


def header_path(m: PackedModel) -> str:
    return os.path.join(BUILD_DIR, f"model_{m.hash}.h")


def _sources_digest(m: PackedModel, flags) -> str:
    """Content hash of everything a library is built from (kernel sources, C-ABI header, generated
    model header, compiler flags).  Staleness is decided by content, not mtimes: the libraries travel
    between machines (build container -> GPU box) where timestamps mean nothing."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(os.path.dirname(HERE), "include", "rbd_hip.h"), "rb") as fh:
        h.update(fh.read())
    h.update(emit_header(m).encode())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


TRANSLATION_UNITS = ["COMMON", "RNEA_F32", "RNEA_F64", "GRAD_F32", "GRAD_F64", "GRADN_F32", "GRADN_F64", "MINV_F32", "MINV_F64",
                     "FD_F32", "FD_F64", "PASS_F32", "PASS_F64"]
# floating-base robots: COMMON from rbd_kernels.hip + the two units of rbd_fb_kernels.hip
FB_TRANSLATION_UNITS = ["COMMON", "FB_F32", "FB_F64"]
class _PrioritySlots:
    """At most `n` hipcc processes at a time; when one finishes, the waiting job with the highest
    cost estimate goes next (longest-first keeps the one 3-minute unit of a 30-body robot from
    starting last and stretching the whole build)."""

    def __init__(self, n: int):
        self.free = n
        self.cv = threading.Condition()
        self.waiting = []          # costs of the jobs that are waiting

    def acquire(self, cost: float):
        with self.cv:
            self.waiting.append(cost)
            while not (self.free > 0 and cost >= max(self.waiting)):
                self.cv.wait()
            self.waiting.remove(cost)
            self.free -= 1
            self.cv.notify_all()

    def release(self):
        with self.cv:
            self.free += 1
            self.cv.notify_all()


_HIPCC_SLOTS = _PrioritySlots(max(1, (os.cpu_count() or 2)))
# relative compile cost of the translation units (measured, 30-body robot), scaled by n^2 per robot
_TU_COST = {"GRAD_F64": 110, "GRADN_F64": 105, "PASS_F32": 140, "PASS_F64": 125, "GRAD_F32": 60, "GRADN_F32": 57, "MINV_F64": 79, "MINV_F32": 74,
            "RNEA_F64": 50, "RNEA_F32": 47, "FD_F32": 45, "FD_F64": 39, "COMMON": 3}


def _run(cmd, what, cost: float = 0.0):
    _HIPCC_SLOTS.acquire(cost)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=BUILD_DIR)
    finally:
        _HIPCC_SLOTS.release()
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed ({what}):\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return r


class _BuildLock:
    """Inter-process lock around check -> build -> publish of one library (``flock`` on
    ``<lib>.lock``).  One process per GPU means every rank may construct ``RBDReference(robot)`` at
    the same moment: the first one builds, the others block here and then find the finished library."""

    def __init__(self, path: str):
        self.path = path + ".lock"
        self.fh = None

    def __enter__(self):
        import fcntl
        self.fh = open(self.path, "w")
        fcntl.flock(self.fh, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.fh, fcntl.LOCK_UN)
        self.fh.close()
        return False


def _up_to_date(out: str, digest: str) -> bool:
    stamp = out + ".stamp"
    try:
        with open(stamp) as f:
            return os.path.exists(out) and f.read().strip() == digest
    except OSError:
        return False


def build_model(m: PackedModel, force: bool = False, verbose: bool = False,
                extra_flags: Optional[list] = None, tag: str = "", link_flags: Optional[list] = None) -> str:
    """Compile the library for packed model `m` (no-op when an up-to-date one exists).  The single
    source file is compiled as several translation units in parallel (-DRBD_TU_*) and linked.
    Safe across processes: the whole check/build/publish sequence holds an exclusive file lock, and
    every intermediate file (generated header, objects, link output) has a process-unique name and
    is published with an atomic rename."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(BUILD_DIR, exist_ok=True)
    out = lib_path(m)
    if tag:                                   # experiment builds live beside the real one
        out = out[:-3] + f".{tag}.so"
    flags = list(HIPCC_FLAGS) + list(extra_flags or [])
    digest = _sources_digest(m, flags)
    if not force and _up_to_date(out, digest):
        return out
    with _BuildLock(out):
        if not force and _up_to_date(out, digest):     # another process finished it while we waited
            return out
        return _build_locked(m, out, flags, digest, force, verbose, list(link_flags or []))


# Host-side sanitizer build (SURVEY.md section 5 "Race detection / sanitizers"): the C-ABI's host code
# -- argument / alignment / workspace checks, the option atomics, the workspace pool, the per-device caches -- compiled
# with AddressSanitizer + UndefinedBehaviorSanitizer (HOST side only: -Xarch_host; GPU sanitizers are not available on
# this pool).  tests/test_host_logic.py loads it in a child process with the sanitizer runtime preloaded and runs the
# no-GPU checks against it.
SANITIZE_COMPILE = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-sanitize=vptr",
                    "-Xarch_host", "-fno-omit-frame-pointer", "-Xarch_host", "-fno-sanitize-recover=undefined"]
SANITIZE_LINK = ["-fsanitize=address,undefined", "-fno-sanitize=vptr", "-shared-libsan"]


def sanitizer_runtime() -> str:
    """Path of the shared ASan runtime of hipcc's clang (LD_PRELOAD it into the process that dlopens the library)."""
    clang = os.path.join(os.path.dirname(os.path.realpath(hipcc_path())), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    r = subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    if r.returncode != 0 or not os.path.isabs(p) or not os.path.exists(p):
        raise RuntimeError("the host AddressSanitizer runtime of hipcc's clang was not found")
    return p


def build_sanitized(m: PackedModel, verbose: bool = False) -> str:
    """The library of packed model `m` with its HOST code under ASan + UBSan: ``librbd_<name>_<hash>.asan.so``."""
    return build_model(m, verbose=verbose, extra_flags=SANITIZE_COMPILE, tag="asan", link_flags=SANITIZE_LINK)


def _build_locked(m, out, flags, digest, force, verbose, link_flags=()) -> str:
    from concurrent.futures import ThreadPoolExecutor
    uniq = f"{os.getpid()}.{threading.get_ident()}"
    hdr = header_path(m)
    hdr_tmp = f"{hdr}.{uniq}.tmp"
    with open(hdr_tmp, "w") as f:
        f.write(emit_header(m))
    os.replace(hdr_tmp, hdr)                  # same content from every writer; rename is atomic
    units = FB_TRANSLATION_UNITS if m.floating else TRANSLATION_UNITS

    def src_of(tu):
        return os.path.join(CSRC, "rbd_fb_kernels.hip" if tu.startswith("FB_") else "rbd_kernels.hip")

    cache_dir = os.path.join(BUILD_DIR, "objcache")
    os.makedirs(cache_dir, exist_ok=True)

    def compile_tu(tu):
        # Objects are cached by the hash of the PREPROCESSED unit: an edit to one header only
        # recompiles the units that include it (the optimiser, not the front end, is the cost).
        import hashlib
        base = [hipcc_path(), *[f for f in flags if f != "-shared"], "-DRBD_TU_SPLIT=1", f"-DRBD_TU_{tu}=1", "-include", hdr]
        src = src_of(tu)
        cost = _TU_COST.get(tu, 50) * m.n * m.n
        pre = _run([*base, "-E", "-P", src, "-o", "-"], f"{m.name} {tu} (preprocess)", cost + 1e9)   # cheap: first
        key = hashlib.sha256((pre.stdout + "\0" + " ".join(flags)).encode()).hexdigest()[:32]
        obj = os.path.join(cache_dir, f"{key}.o")
        if os.path.exists(obj) and not force:
            return obj
        tmp = f"{obj}.{uniq}.{tu}.tmp"
        cmd = [*base, "-c", src, "-o", tmp]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        if os.environ.get("RBD_BUILD_TIMES"):     # CPU seconds of this unit (children of this thread's call)
            t0 = time.time()
            r = _run(["/bin/bash", "-c", 'TIMEFORMAT="%U %S"; time "$@"', "sh", *cmd], f"{m.name} {tu}", cost)
            cpu = r.stderr.strip().splitlines()[-1] if r.stderr.strip() else "?"
            print(f"[build] {m.name:24s} {tu:9s} wall {time.time() - t0:7.1f} s  cpu(user sys) {cpu}", file=sys.stderr)
        else:
            r = _run(cmd, f"{m.name} {tu}", cost)
        if verbose and r.stderr:
            print(r.stderr, file=sys.stderr)
        os.replace(tmp, obj)
        return obj

    with ThreadPoolExecutor(max_workers=len(units)) as ex:
        objs = list(ex.map(compile_tu, units))
    link_tmp = f"{out}.{uniq}.tmp"
    _run([hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *link_flags, *objs, "-o", link_tmp],
         f"{m.name} link", 2e9)
    os.replace(link_tmp, out)
    stamp_tmp = f"{out}.stamp.{uniq}.tmp"
    with open(stamp_tmp, "w") as f:
        f.write(digest + "\n")
    os.replace(stamp_tmp, out + ".stamp")
    return out


# ---- first-use builds: one small library per FAMILY of entry points ---------------------------------------------
# A robot that has never been built blocks its first call for as long as its slowest translation unit takes (the
# gradient unit of a 30-body robot: minutes), although that call needs one family of kernels in one precision.
# A family library = COMMON + the units of that family + stubs for every other entry point (they return
# RBD_ERR_NOT_BUILT), linked and loaded like a full library; rbdreference_amd/_lib.py serves calls from family
# libraries while the full library builds in the background and switches to it when it is ready.  The gradient
# family is compiled with -DRBD_FAST_STAGE=1: only the kernel AUTO picks for large batches.
FAMILIES = {           # family -> units (suffix _F32 / _F64 appended)
    "rnea": ["RNEA"],
    "grad": ["GRAD"],                           # rbd_rnea_grad with qdd
    "gradn": ["GRAD", "GRADN"],                 # ... with qdd = None (:589)
    "gradr": ["GRAD", "GRADN", "RNEA"],         # rbd_rnea_with_grad: runs the rnea kernel too
    "minv": ["MINV"],
    "fd": ["FD", "RNEA", "MINV"],
    "pass": ["PASS"],
}
_FAST_UNITS = {"GRAD", "GRADN", "RNEA"}
_ALL_FAMILY_UNITS = ["RNEA", "GRAD", "GRADN", "MINV", "FD", "PASS"]


def family_of(symbol: str, has_qdd: bool = True) -> str:
    """Family that serves entry point `symbol` (without the _f32 / _f64 suffix)."""
    if symbol in ("rbd_rnea", "rbd_rnea_fpass", "rbd_rnea_bpass"):
        return "rnea"
    if symbol == "rbd_rnea_grad":
        return "grad" if has_qdd else "gradn"
    if symbol == "rbd_rnea_with_grad":
        return "gradr"
    if symbol in ("rbd_minv", "rbd_crba", "rbd_minv_workspace_bytes"):
        return "minv"
    if symbol in ("rbd_aba", "rbd_forward_dynamics", "rbd_forward_dynamics_grad"):
        return "fd"
    return "pass"


def family_lib_path(m: PackedModel, family: str, prec: str) -> str:
    return lib_path(m)[:-3] + f".{family}_{prec}.so"


def build_family(m: PackedModel, family: str, prec: str, verbose: bool = False) -> str:
    """Build (if needed) the family library of packed model `m`: ``family`` in FAMILIES, ``prec`` 'f32' | 'f64'.
    Fixed-base robots only (a floating-base library is three units: nothing to stage)."""
    from concurrent.futures import ThreadPoolExecutor
    import hashlib
    if m.floating:
        return build_model(m, verbose=verbose)
    os.makedirs(BUILD_DIR, exist_ok=True)
    out = family_lib_path(m, family, prec)
    flags = list(HIPCC_FLAGS)
    digest = _sources_digest(m, flags + [family, prec])
    if _up_to_date(out, digest):
        return out
    P = prec.upper()
    units = [f"{u}_{P}" for u in FAMILIES[family]]
    missing = [f"{u}_{q}" for u in _ALL_FAMILY_UNITS for q in ("F32", "F64") if f"{u}_{q}" not in units]
    with _BuildLock(out):
        if _up_to_date(out, digest):
            return out
        uniq = f"{os.getpid()}.{threading.get_ident()}"
        hdr = header_path(m)
        hdr_tmp = f"{hdr}.{uniq}.tmp"
        with open(hdr_tmp, "w") as f:
            f.write(emit_header(m))
        os.replace(hdr_tmp, hdr)
        cache_dir = os.path.join(BUILD_DIR, "objcache")
        os.makedirs(cache_dir, exist_ok=True)
        src = os.path.join(CSRC, "rbd_kernels.hip")

        srcs = _sources_digest(m, flags)

        def compile_unit(defs, what, cost):
            # (keyed by the digest of ALL sources + the unit's defines, not by the preprocessed text as the full build
            # does: the extra preprocessor run costs 2 s, a fifth of a first-use build)
            base = [hipcc_path(), *[f for f in flags if f != "-shared"], "-DRBD_TU_SPLIT=1", *defs, "-include", hdr]
            key = hashlib.sha256((srcs + "\0" + " ".join(defs)).encode()).hexdigest()[:32]
            obj = os.path.join(cache_dir, f"fam_{key}.o")
            if os.path.exists(obj):
                return obj
            tmp = f"{obj}.{uniq}.{what}.tmp"
            r = _run([*base, "-c", src, "-o", tmp], f"{m.name} {what}", cost)
            if verbose and r.stderr:
                print(r.stderr, file=sys.stderr)
            os.replace(tmp, obj)
            return obj

        jobs = [(["-DRBD_TU_COMMON=1"], "COMMON", 3.0),
                (["-DRBD_TU_STUBS=1", *[f"-DRBD_STUB_{u}=1" for u in missing]], f"STUBS[{family}_{prec}]", 3.0)]
        for u in units:
            fast = ["-DRBD_FAST_STAGE=1"] if u.split("_")[0] in _FAST_UNITS else []
            jobs.append(([f"-DRBD_TU_{u}=1", *fast], u + ("(fast)" if fast else ""), _TU_COST.get(u, 50) * m.n * m.n))
        with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            objs = list(ex.map(lambda j: compile_unit(*j), jobs))
        link_tmp = f"{out}.{uniq}.tmp"
        _run([hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", link_tmp], f"{m.name} {family}_{prec} link", 2e9)
        os.replace(link_tmp, out)
        stamp_tmp = f"{out}.stamp.{uniq}.tmp"
        with open(stamp_tmp, "w") as f:
            f.write(digest + "\n")
        os.replace(stamp_tmp, out + ".stamp")
    return out


def full_library_ready(m: PackedModel) -> bool:
    """True when the full library of `m` exists and was built from the current sources."""
    return _up_to_date(lib_path(m), _sources_digest(m, list(HIPCC_FLAGS)))


# ---- the model-handle library (include/rbd_generic.h): one build for every robot ------------------------------------
GENERIC_SRC = os.path.join(HERE, "csrc_generic", "rbd_generic.hip")
GENERIC_FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared"]


def generic_lib_path() -> str:
    # (RBD_GENERIC_LIB: a variant build for timing experiments, tools/time_generic.py)
    return os.environ.get("RBD_GENERIC_LIB") or os.path.join(BUILD_DIR, "librbd_generic.so")


def _generic_digest() -> str:
    import hashlib
    h = hashlib.sha256()
    for f in (GENERIC_SRC, os.path.join(os.path.dirname(HERE), "include", "rbd_generic.h"), os.path.join(CSRC, "rbd_sincos.h")):
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(GENERIC_FLAGS).encode())
    return h.hexdigest()


def generic_library_ready() -> bool:
    return _up_to_date(generic_lib_path(), _generic_digest())


def build_generic(force: bool = False) -> str:
    """Compile librbd_generic.so (robot-independent: the model is a run-time table).  ~15 s, once per source change;
    `__graft_entry__.build()` / an install step does it, so a user's machine needs no compiler for it."""
    os.makedirs(BUILD_DIR, exist_ok=True)
    out = generic_lib_path()
    digest = _generic_digest()
    if not force and _up_to_date(out, digest):
        return out
    with _BuildLock(out):
        if not force and _up_to_date(out, digest):
            return out
        uniq = f"{os.getpid()}.{threading.get_ident()}"
        tmp = f"{out}.{uniq}.tmp"
        _run([hipcc_path(), *GENERIC_FLAGS, GENERIC_SRC, "-o", tmp], "generic library", 1e9)
        os.replace(tmp, out)
        stamp_tmp = f"{out}.stamp.{uniq}.tmp"
        with open(stamp_tmp, "w") as f:
            f.write(digest + "\n")
        os.replace(stamp_tmp, out + ".stamp")
    return out


def build_models_parallel(models, force: bool = False, jobs: Optional[int] = None) -> list:
    """Build several per-robot libraries concurrently (one hipcc process each)."""
    from concurrent.futures import ThreadPoolExecutor
    jobs = jobs or max(1, len(models))      # hipcc processes are throttled by _HIPCC_SLOTS
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        return list(ex.map(lambda m: build_model(m, force=force), models))


def build_robot(robot, name: Optional[str] = None, **kw) -> str:
    return build_model(pack_robot(robot, name), **kw)


def build_builtins(force: bool = False, verbose: bool = False) -> list:
    from .robot import BUILTIN_ROBOTS
    return [build_robot(mk(), force=force, verbose=verbose) for mk in BUILTIN_ROBOTS.values()]


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args()
    for p in build_builtins(a.force, a.verbose):
        print(p)
