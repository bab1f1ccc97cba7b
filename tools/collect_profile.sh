#!/bin/bash
# Copy the summaries of `bash tools/profile_round.sh <dir>` (merged back under gpurun_out/<dir>) into profiles/ and
# regenerate profiles/hbm_traffic.json.   bash tools/collect_profile.sh <dir> <round tag, e.g. r02_final>
set -eu
O=gpurun_out/$1; T=$2
cp $O/bench_stats/runc/*_kernel_stats.csv profiles/${T}_bench_kernel_stats.csv
cp $O/configs_stats/runc/*_kernel_stats.csv profiles/${T}_configs_kernel_stats.csv
cp $O/pmc_head_summary.json profiles/${T}_pmc_rnea_grad_B1048576.json
cp $O/pmc_cfg_summary.json profiles/${T}_pmc_configs.json
cp $O/bench.json profiles/${T}_bench.json.log
cp $O/stalls/pmc_stalls.txt profiles/${T}_pmc_stalls.txt
cp $O/bench_torchrun1.json profiles/${T}_bench_torchrun1.json.log
python tools/make_traffic.py $O/pmc_head_summary.json "rnea_grad_idsva_pipe_kernel<float,true,false>" 1048576 > /dev/null
python tools/make_config_traffic.py profiles/${T}_pmc_configs.json > /dev/null
python - <<P
import bench, json
t = json.load(open("profiles/hbm_traffic.json"))
d = json.loads([l for l in open("profiles/${T}_bench.json.log") if l.startswith("{")][0])
print("digest now / traffic file / bench line:", bench.sources_digest(), t["sources_digest"], d["roofline"]["sources_digest"])
print("traffic / algorithmic:", t["traffic_over_algorithmic"], " frac:", d["roofline"]["frac"], " kernel_ms:", d["roofline"]["kernel_ms"])
P
