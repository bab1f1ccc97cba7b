#!/bin/bash
# Where a wave of the headline kernel spends its cycles: instruction cache, FIFO-full stalls, issue cycles per
# instruction class (separate --pmc passes of tools/run_grad.py).   bash tools/pmc_stalls.sh <out dir under gpurun_out/>
set -u
OUT=gpurun_out/${1:-stalls}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES" "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  REPS=6 timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -- python3 tools/run_grad.py > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<P
import csv, glob, os
from collections import defaultdict
acc = defaultdict(list)
names = set()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "idsva" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            names.add(r["Kernel_Name"].split("(")[0].replace("void rbdk::", ""))
with open("$OUT/pmc_stalls.txt", "w") as fh:
    # the header names the kernel(s) the rows actually belong to (it used to be a literal string)
    fh.write("# " + " | ".join(sorted(names)) + ", B = 1 048 576: mean counter value per dispatch (SQ_* cycle counters in quad-cycles)\n")
    for k, v in sorted(acc.items()):
        line = f"{k:34s} {sum(v)/len(v):16.0f}  n={len(v)}"
        print(line); fh.write(line + "\n")
P
