#!/usr/bin/env python3
"""hipModuleOccupancyMaxActiveBlocksPerMultiprocessor of kernels of a per-robot library, from its embedded gfx950 code objects
(on the GPU box):   python tools/occupancy_probe.py <lib.so> <kernel-name-filter> <block threads> <dynamic LDS bytes> [...]"""
import ctypes, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: F401  (loads the HIP runtime the process should use)
from isa_exec_audit import code_objects, ROCM_LLVM

hip = ctypes.CDLL("libamdhip64.so")
lib, flt, threads = sys.argv[1], sys.argv[2], int(sys.argv[3])
ldss = [int(x) for x in sys.argv[4:]] or [0]
torch.cuda.init(); torch.zeros(1, device="cuda")
with tempfile.TemporaryDirectory() as td:
    for co in code_objects(lib, td):
        syms = subprocess.run([f"{ROCM_LLVM}/llvm-objdump", "--syms", co], capture_output=True, text=True).stdout
        names = [l.split()[-1] for l in syms.splitlines() if " F .text" in l]
        for nm in names:
            dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
            if flt not in dem or nm.endswith(".kd"):
                continue
            mod = ctypes.c_void_p(); fn = ctypes.c_void_p()
            data = open(co, "rb").read()
            assert hip.hipModuleLoadData(ctypes.byref(mod), data) == 0
            if hip.hipModuleGetFunction(ctypes.byref(fn), mod, nm.encode()) != 0:
                continue
            for lds in ldss:
                n = ctypes.c_int()
                rc = hip.hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(ctypes.byref(n), fn, threads, ctypes.c_size_t(lds))
                print(f"{re.sub(r'[(].*', '', dem)[:70]:70s} threads {threads} lds {lds:6d} B -> rc {rc} blocks/CU {n.value}")
