#!/usr/bin/env python3
"""Variant builds of ONE unit (UNIT=GRAD | MINV | ..., default GRAD) for tools/exp_grad.py: COMMON + the GRAD unit of a precision (fast stage) with
extra flags + stubs for everything else, linked as  librbd_<robot>_<hash>.<tag>.so  (seconds to a minute per variant
instead of the whole library).

    ROBOT=atlas_like python tools/exp_tws.py f64 dreg4=-DRBD_TWS_DREG=4 dreg5=-DRBD_TWS_DREG=5
    ROBOT=atlas_like DTYPE=f64 python tools/exp_grad.py run 16384          # on the GPU box
"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from rbdreference_amd import builtin_robot, pack_robot
    from rbdreference_amd.build import (ARCH, BUILD_DIR, CSRC, HIPCC_FLAGS, _ALL_FAMILY_UNITS, header_path, hipcc_path, lib_path)
    from rbdreference_amd.packer import emit_header
    m = pack_robot(builtin_robot(os.environ.get("ROBOT", "atlas_like")))
    prec = sys.argv[1].upper()
    specs = [a.split("=", 1) for a in sys.argv[2:]]
    hdr = header_path(m)
    if not os.path.exists(hdr):
        os.makedirs(BUILD_DIR, exist_ok=True)
        open(hdr, "w").write(emit_header(m))
    src = os.path.join(CSRC, "rbd_kernels.hip")
    base = [hipcc_path(), *[f for f in HIPCC_FLAGS if f != "-shared"], "-DRBD_TU_SPLIT=1", "-include", hdr]
    tmp = os.path.join(BUILD_DIR, "exp_tws")
    os.makedirs(tmp, exist_ok=True)
    unit = f"{os.environ.get('UNIT', 'GRAD')}_{prec}"
    missing = [f"{u}_{q}" for u in _ALL_FAMILY_UNITS for q in ("F32", "F64") if f"{u}_{q}" != unit]

    def cc(defs, out):
        subprocess.run([*base, *defs, "-c", src, "-o", out], check=True)
        return out

    common = cc(["-DRBD_TU_COMMON=1"], os.path.join(tmp, f"common_{m.hash}.o"))
    stubs = cc(["-DRBD_TU_STUBS=1", *[f"-DRBD_STUB_{u}=1" for u in missing]], os.path.join(tmp, f"stubs_{m.hash}_{prec}.o"))

    def variant(spec):
        tag, fl = spec
        obj = cc([f"-DRBD_TU_{unit}=1", *(["-DRBD_FAST_STAGE=1"] if unit.startswith("GRAD") else []), *[f for f in fl.split(",") if f]], os.path.join(tmp, f"{tag}_{m.hash}.o"))
        out = lib_path(m)[:-3] + f".{tag}.so"
        subprocess.run([hipcc_path(), f"--offload-arch={ARCH}", "-shared", "-fPIC", common, stubs, obj, "-o", out], check=True)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), out, os.environ.get("KFILTER", "rnea_grad")], capture_output=True, text=True)
        return out + "\n" + r.stdout

    with ThreadPoolExecutor(4) as ex:
        for p in ex.map(variant, specs):
            print(p)


if __name__ == "__main__":
    main()
