#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per (kernel, grid size) mean counter values.

    python tools/pmc_summary.py <dir with *_counter_collection.csv> [kernel substring] > summary.json
"""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if flt and flt not in k:
                continue
            key = f"{k.split('(')[0]} grid={r.get('Grid_Size')} wg={r.get('Workgroup_Size')} vgpr={r.get('VGPR_Count')} lds={r.get('LDS_Block_Size')}"
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for key, cs in acc.items():
        out[key] = {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in sorted(cs.items())}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
