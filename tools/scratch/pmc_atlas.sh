set -u
OUT=gpurun_out/r2w
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pass in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_LDS" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  REPS=6 timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p$i -- python3 tools/run_configs.py atlas > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<P
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "minv" in k or "segments" in k:
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()): print(f"   {c:28s} {sum(v)/len(v):14.0f}")
P
