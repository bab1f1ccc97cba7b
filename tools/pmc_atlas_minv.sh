OUT=gpurun_out/pmc_atlas_minv
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CU_CYCLES SQ_IFETCH"; do
  tag=$(echo $pass | cut -d' ' -f1)
  REPS=6 timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/$tag -- python3 tools/run_configs.py atlas > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<P
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "minv_fused" not in k: continue
        acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()): print("   %-28s %14.0f  n=%d" % (c, sum(v)/len(v), len(v)))
P
