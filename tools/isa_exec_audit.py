#!/usr/bin/env python3
"""Audit the gfx950 ISA of a per-robot library for vector writes that execute in an EXEC-masked window.

Background (docs/NOTEBOOK.md 3.1 a' (xv), round 3): `rnea_grad_idsva_kernel<double>` faulted because the register
allocator had placed a live-range copy (`v_accvgpr_write_b32 a1, v73`) at the HEAD of the ELSE side of a lane-masked
if / else -- i.e. in the block that `s_cbranch_execz` of the IF side jumps to -- BEFORE the instruction that restores
EXEC there (`s_or_saveexec_b64` / `s_andn2_saveexec_b64`).  When no lane takes the IF side the branch arrives with
EXEC = 0 and the copy does nothing.  The only lane-masked if / else in the device code was the large-argument test
inside the library `sincos(double)`.

What this tool reports, per kernel (reads the embedded code objects; runs without a GPU):
  * `else_blocks`   -- number of lane-masked if / ELSE lowerings (`s_or_saveexec` / `s_andn2_saveexec`),
  * `masked_writes` -- vector instructions (v_*, incl. v_accvgpr_write, buffer / ds / global loads) that sit between a
                       label targeted by `s_cbranch_execz` and the first instruction that rewrites EXEC: these run
                       with EXEC = 0 whenever the branch is taken.  Any entry here is a candidate for the round-3 fault.
  * `calls`         -- `s_swappc_b64` (out-of-line calls; none are expected).

    python tools/isa_exec_audit.py rbdreference_amd/_build/librbd_iiwa_like_*.so [kernel-name-filter] [-v]
"""
import os
import re
import subprocess
import sys
import tempfile

ROCM_LLVM = "/opt/rocm/lib/llvm/bin"
EXEC_WRITE = re.compile(r"^\s*(s_\w+saveexec_b64|s_(?:or|and|andn2|xor|mov|orn2|xnor|nand|nor|cselect|wqm)_b64\s+exec\b)")


def code_objects(lib, td):
    fb = os.path.join(td, "fat.bin")
    subprocess.run([f"{ROCM_LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True)
    blob = open(fb, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(magic, blob)]
    out = []
    for k, st in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(blob)
        part = os.path.join(td, f"b{k}.bin")
        co = os.path.join(td, f"b{k}.co")
        open(part, "wb").write(blob[st:end])
        subprocess.run([f"{ROCM_LLVM}/clang-offload-bundler", "--unbundle", "--type=o",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}", f"--output={co}"],
                       check=True, capture_output=True)
        if os.path.getsize(co):
            out.append(co)
    return out


def audit(lib, flt="", verbose=False):
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for co in code_objects(lib, td):
            dis = subprocess.run([f"{ROCM_LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--symbolize-operands", co],
                                 capture_output=True, text=True).stdout
            cur, body = None, []
            funcs = []
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\w+)>:", line)
                if m and not m.group(1).startswith("L"):
                    if cur:
                        funcs.append((cur, body))
                    cur, body = m.group(1), []
                elif cur is not None:
                    body.append(line)
            if cur:
                funcs.append((cur, body))
            for name, body in funcs:
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r"\(.*", "", dem).replace("void rbdk::", "")
                if flt and flt not in dem:
                    continue
                # labels that some s_cbranch_execz jumps to
                execz_targets = set()
                for ln in body:
                    m = re.search(r"s_cbranch_execz\s+(\S+)", ln)
                    if m:
                        execz_targets.add(m.group(1).strip("<>").split()[0])
                else_blocks = sum(1 for ln in body if re.search(r"s_(or|andn2)_saveexec_b64", ln))
                calls = sum(1 for ln in body if "s_swappc_b64" in ln)
                masked = []
                in_window = None
                for ln in body:
                    m = re.match(r"^<?(L\d+)>?:", ln.strip())
                    if m:
                        in_window = m.group(1) if m.group(1) in execz_targets else None
                        continue
                    ins = re.sub(r"^\s*[0-9a-f]*:?\s*", "", ln.split("//")[0]).strip()
                    if not ins:
                        continue
                    if in_window:
                        if EXEC_WRITE.match(ins) or ins.startswith("s_cbranch") or ins.startswith("s_branch") or ins.startswith("s_endpgm"):
                            in_window = None
                        elif re.match(r"(v_|buffer_|global_|flat_|ds_|scratch_)", ins) and not ins.startswith("v_readfirstlane") and not ins.startswith("v_readlane"):
                            masked.append((in_window, ins))
                rows.append((dem, len(body), else_blocks, len(execz_targets), calls, masked))
    return rows


def main():
    args = [a for a in sys.argv[1:] if a != "-v"]
    verbose = "-v" in sys.argv
    lib = args[0]
    flt = args[1] if len(args) > 1 else ""
    bad = 0
    for dem, n, eb, et, calls, masked in audit(lib, flt, verbose):
        print(f"{dem:70s} insts={n:6d} else_blocks={eb:3d} execz_targets={et:3d} calls={calls} masked_writes={len(masked)}")
        bad += len(masked)
        if verbose or masked:
            for lab, ins in masked[:20]:
                print(f"      {lab}: {ins}")
    print(f"total masked-window vector instructions: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
