#!/bin/bash
# round 3: everything profiles/r03/ holds for the final state, one gpurun call:
#   default bench line (driver's command), rocprofv3 kernel stats + PMC passes of the headline workload AND of configs[3] (--dims C4)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; TAG="${1:-q}"
cd "$REPO"; mkdir -p gpurun_out
json_line() { grep '^{' "$1" | tail -1 > "$2"; }
timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2>&1; echo "bench exit: $?"
json_line gpurun_out/bench_$TAG.log gpurun_out/bench_$TAG.json
bash scripts/gpu_profile.sh $TAG > gpurun_out/prof_$TAG.txt 2>&1
bash scripts/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.txt 2>&1
bash scripts/gpu_profile.sh ${TAG}_c4 --dims C4 --steps 8 > gpurun_out/prof_${TAG}_c4.txt 2>&1
MOPS=SQ_INSTS_VALU_MFMA_MOPS_BF16 bash scripts/gpu_pmc.sh ${TAG}_c4 --dims C4 > gpurun_out/pmc_${TAG}_c4.txt 2>&1
python - <<PY
import json
d = json.load(open("gpurun_out/bench_$TAG.json"))
print("bench", d["value"], d["ms_per_step"], d.get("roofline"), d.get("mfma_frac_end_to_end"))
PY
head -14 gpurun_out/prof_$TAG/kernel_stats.csv
head -14 gpurun_out/prof_${TAG}_c4/kernel_stats.csv
tail -30 gpurun_out/pmc_${TAG}_c4/summary.json
