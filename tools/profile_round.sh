#!/bin/bash
# Round profile (run on the GPU box from the repo root):  bash tools/profile_round.sh <out dir under gpurun_out/>
# Kernel stats of the bench command, PMC passes (separate, --kernel-trace only) of the headline kernel
# and of the BASELINE configs[1]/[2]/[4] kernels, the bench line itself, and the N = 1 torchrun rehearsal.
set -u
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err; echo bench_rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extra > $OUT/bench_prof.json 2> $OUT/bench_prof.err; echo benchprof_rc=$?
REPS=20 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/configs_stats -- python3 tools/run_configs.py > $OUT/configs.log 2>&1; echo configs_rc=$?
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  REPS=6 timeout -k 10 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_head_$tag -- python3 tools/run_grad.py > $OUT/pmc_head_$tag.log 2>&1 || echo "pmc head $tag failed"
done
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | cut -d' ' -f1)
  REPS=6 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_cfg_$tag -- python3 tools/run_configs.py atlas atlas64 iiwa4k quad fb fdg > $OUT/pmc_cfg_$tag.log 2>&1 || echo "pmc cfg $tag failed"
done
bash tools/pmc_stalls.sh ${1:-prof}/stalls > $OUT/pmc_stalls.log 2>&1 || echo "pmc stalls failed"
mkdir -p $OUT/head $OUT/cfg
python3 tools/pmc_summary.py $OUT rnea_grad_idsva > $OUT/pmc_head_summary.json
for d in $OUT/pmc_cfg_*; do :; done
python3 - <<P
import json, subprocess, sys, glob, os
out = "$OUT"
# configs: summarise only the pmc_cfg_* directories
import csv
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_cfg_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "rbdk" not in k: continue
        key = f"{k.split('(')[0]} grid={r.get('Grid_Size')} wg={r.get('Workgroup_Size')}"
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
json.dump({k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in sorted(cs.items())} for k, cs in acc.items()},
          open(os.path.join(out, "pmc_cfg_summary.json"), "w"), indent=1)
P
timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 1 --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err; echo torchrun_rc=$?
tail -c 300 $OUT/bench_torchrun1.json
