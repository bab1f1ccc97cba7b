#!/bin/bash
# Instruction-cache, LDS / VMEM FIFO and issue-class counters of ONE kernel of tools/run_configs.py (separate --pmc
# passes, --kernel-trace only).   bash tools/pmc_kernel.sh <out dir under gpurun_out/> <kernel substring> <run_configs.py robots...>
set -u
OUT=gpurun_out/${1:-pmck}; KSUB=${2:-minv_fused}; shift 2
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES" \
            "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" \
            "SQ_ACTIVE_INST_VALU2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM" \
            "SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  REPS=6 timeout -k 10 150 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -- python3 tools/run_configs.py "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<P
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/pmc_${KSUB}.txt", "w") as fh:
    fh.write("# $KSUB ($*): mean counter value per dispatch (SQ_* cycle counters in quad-cycles)\n")
    for k, v in sorted(acc.items()):
        line = f"{k:34s} {sum(v)/len(v):16.0f}  n={len(v)}"
        print(line); fh.write(line + "\n")
P
