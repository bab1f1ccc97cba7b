#!/bin/bash
# variants:  ROBOT=atlas_like UNIT=MINV KFILTER=minv_fused python tools/exp_tws.py f32 nolegs=-DRBD_MF_EXP_NOLEGS notorso=-DRBD_MF_EXP_NOTORSO \
#     stop1=-DRBD_MF_EXP_STOP=1,-DRBD_MF_EXP_NOLEGS stop2=-DRBD_MF_EXP_STOP=2,-DRBD_MF_EXP_NOLEGS stop3=-DRBD_MF_EXP_STOP=3,-DRBD_MF_EXP_NOLEGS noepi=-DRBD_MF_EXP_NOEPI,-DRBD_MF_EXP_NOLEGS
# SQ counters of the Atlas minv_fused_kernel<float> switch-off variants (tools/exp_tws.py tags): which phase owns the instructions.
OUT=gpurun_out/pmc_atlas_var
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ROBOT=atlas_like python3 tools/exp_minv_abi.py 16384 > $OUT/times.txt 2>&1
for v in base nolegs notorso stop1 stop2 stop3 noepi; do
  for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    VARIANT=$v ROBOT=atlas_like timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$v -- python3 tools/exp_minv_abi.py 16384 > $OUT/$v.log 2>&1 || echo "pass $v failed"
  done
done
python3 - <<P
import csv, glob, os
from collections import defaultdict
for v in "base nolegs notorso stop1 stop2 stop3 noepi".split():
    acc = defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            if "minv_fused" in r.get("Kernel_Name", ""): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("%-8s " % v + "  ".join("%s=%.3gM" % (c.replace("SQ_", ""), sum(x) / len(x) / 1e6) for c, x in sorted(acc.items())))
P
cat $OUT/times.txt
