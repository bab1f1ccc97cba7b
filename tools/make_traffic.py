#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a tools/pmc_summary.py JSON (FETCH_SIZE / WRITE_SIZE passes, optionally
the SQ passes) of `tools/run_grad.py` on the headline workload.

    python tools/make_traffic.py <pmc_summary.json> <kernel substring> <batch> [out.json]

HBM bytes per launch = 2 x FETCH_SIZE (gfx950 tallies the 128-B requests of wide coalesced reads at
64 B: MI355X_MICROARCH.md, HBM) + WRITE_SIZE, both reported by rocprofv3 in KiB.  The file carries the
kernel name and the digest of the kernel sources it was measured on; bench.py attaches it to a bench
line only when both match what it is timing.
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def main():
    from bench import sources_digest
    summ = json.load(open(sys.argv[1])); sub = sys.argv[2]; B = int(sys.argv[3])
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "hbm_traffic.json")
    keys = [k for k in summ if sub in k.replace(" ", "")]
    assert len(keys) == 1, keys
    c = {k: v["mean"] for k, v in summ[keys[0]].items()}
    name = keys[0].split(" grid=")[0].replace("void rbdk::", "").replace(" ", "")
    fetch = 2.0 * c["FETCH_SIZE"] * 1024.0
    write = c["WRITE_SIZE"] * 1024.0
    alg = B * 504.0
    j = {"batch": B, "kernel": name, "sources_digest": sources_digest(),
         "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
         "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": (fetch + write) / alg,
         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/run_grad.py), KiB per dispatch; "
                 "FETCH x2 = gfx950 correction",
         "source": os.path.relpath(sys.argv[1], ROOT)}
    if "SQ_INSTS_VALU" in c and "SQ_WAVE_CYCLES" in c:
        waves = c.get("SQ_WAVES", 0)
        j["valu"] = {"valu_wave_instructions_per_launch": c["SQ_INSTS_VALU"],
                     "per_64_evaluations": c["SQ_INSTS_VALU"] / (B / 64.0),
                     "wave_cycles_quad": c["SQ_WAVE_CYCLES"], "waves": waves,
                     "active_frac": c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
                     "wait_any_frac": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"],
                     "wait_inst_any_frac": c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"],
                     # tile-walking kernel: all `waves` are resident for the whole launch, so the launch lasts
                     # SQ_WAVE_CYCLES x 4 / waves shader cycles; a SIMD issues one wave64 VALU instruction per 2 cycles
                     "launch_cycles": c["SQ_WAVE_CYCLES"] * 4.0 / max(waves, 1),
                     "simd_valu_busy_frac": (c["SQ_INSTS_VALU"] * 2.0 / 1024.0) / (c["SQ_WAVE_CYCLES"] * 4.0 / max(waves, 1))}
        # fp32 flops per launch from the per-opcode-class counters (wave-instructions x 64 lanes; an FMA is two flops)
        fl = {k: c.get(k) for k in ("SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32")}
        if fl["SQ_INSTS_VALU_FMA_F32"] is not None:
            j["valu"].update({k.lower(): v for k, v in fl.items() if v is not None})
            j["valu"]["flops_per_launch"] = 64.0 * (2.0 * fl["SQ_INSTS_VALU_FMA_F32"] + (fl["SQ_INSTS_VALU_MUL_F32"] or 0.0) +
                                                    (fl["SQ_INSTS_VALU_ADD_F32"] or 0.0) + (fl["SQ_INSTS_VALU_TRANS_F32"] or 0.0))
            j["valu"]["flops_per_evaluation"] = j["valu"]["flops_per_launch"] / B
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(j, indent=1))


if __name__ == "__main__":
    main()
