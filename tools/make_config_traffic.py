#!/usr/bin/env python3
"""profiles/config_traffic.json from a profile round's pmc_cfg_summary.json (tools/profile_round.sh): HBM bytes per launch
of the non-headline kernels (2 x FETCH_SIZE + WRITE_SIZE, KiB per dispatch, separate --pmc passes; gfx950 x2 fetch correction as
in tools/make_traffic.py), keyed by kernel name, with the digest of the kernel sources they were measured on.  bench.py attaches
them to its `extra` entries when the digest matches.

    python tools/make_config_traffic.py profiles/r03b_final_pmc_configs.json
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def main():
    from bench import sources_digest
    summ = json.load(open(sys.argv[1]))
    out = {"sources_digest": sources_digest(), "source": os.path.relpath(sys.argv[1], ROOT),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/run_configs.py; bytes per launch = 2 x FETCH (gfx950) + WRITE",
           "kernels": {}}
    for k, v in summ.items():
        if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
            continue
        name = k.split(" grid=")[0].replace("void rbdk::", "").replace(" ", "")
        grid = k.split(" grid=")[1] if " grid=" in k else ""
        f = 2.0 * v["FETCH_SIZE"]["mean"] * 1024.0
        w = v["WRITE_SIZE"]["mean"] * 1024.0
        out["kernels"].setdefault(name, []).append({"grid": grid, "hbm_bytes_per_launch": f + w, "fetch_bytes_corrected_x2": f, "write_bytes": w})
    json.dump(out, open(os.path.join(ROOT, "profiles", "config_traffic.json"), "w"), indent=1)
    for n, e in out["kernels"].items():
        print(n, [round(x["hbm_bytes_per_launch"] / 1e6, 1) for x in e], "MB")


if __name__ == "__main__":
    main()
