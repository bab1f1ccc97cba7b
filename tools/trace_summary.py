#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_trace.csv per (kernel, grid): median/min duration, VGPR, LDS, scratch."""
import csv, glob, collections, sys
f = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob('gpurun_out/*/runc/*kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"].split("(")[0].replace("void rbdk::", "")
    if "at::native" in nm or "rocclr" in nm: continue
    agg[(nm, r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v = sorted(v)
    print(f"{k[0]:36s} grid={k[1]:>9s} wg={k[2]:>4s} vgpr={k[3]:>4s}+{k[4]:<3s} lds={k[5]:>6s} scr={k[6]:>4s} n={len(v):3d} med={v[len(v)//2]:9.1f} us min={v[0]:9.1f}")
