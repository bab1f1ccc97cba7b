#!/usr/bin/env python3
"""Print VGPR / scratch / LDS usage of every kernel in a per-robot library (reads the code-object
metadata of the embedded gfx950 image; runs without a GPU).

    python tools/kernel_resources.py rbdreference_amd/_build/librbd_iiwa_like_*.so [filter]
"""
import re
import subprocess
import sys
import tempfile
import os

ROCM_LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    lib = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    txt = ""
    with tempfile.TemporaryDirectory() as td:
        # every translation unit contributes one offload bundle to the .hip_fatbin section
        fb = os.path.join(td, "fat.bin")
        subprocess.run([f"{ROCM_LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True)
        blob = open(fb, "rb").read()
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(magic, blob)]
        for k, st in enumerate(starts):
            end = starts[k + 1] if k + 1 < len(starts) else len(blob)
            part = os.path.join(td, f"b{k}.bin"); co = os.path.join(td, f"b{k}.co")
            open(part, "wb").write(blob[st:end])
            subprocess.run([f"{ROCM_LLVM}/clang-offload-bundler", "--unbundle", "--type=o",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}", f"--output={co}"],
                           check=True, capture_output=True)
            txt += subprocess.run([f"{ROCM_LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    cur = {}
    rows = []
    for line in txt.splitlines():
        m = re.match(r"\s+\.(\w+):\s+(.*)", line.replace("- .", "  ."))
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and cur:
            rows.append(cur); cur = {}
        cur[k] = v
    if cur:
        rows.append(cur)
    for r in rows:
        nm = r.get("name", "?")
        dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("void rbdk::", "")
        if flt and flt not in dem:
            continue
        print(f"{dem:58s} vgpr={r.get('vgpr_count','?'):>4} agpr={r.get('agpr_count','?'):>3} sgpr={r.get('sgpr_count','?'):>3} "
              f"scratch={r.get('private_segment_fixed_size','?'):>5} lds={r.get('group_segment_fixed_size','?'):>6} "
              f"spill={r.get('vgpr_spill_count','?')} sgpr_spill={r.get('sgpr_spill_count','?')}")


if __name__ == "__main__":
    main()
