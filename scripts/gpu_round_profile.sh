#!/bin/bash
# Everything profiles/<round>/ holds for one state of the code, in one gpurun call:
#   bench line (default command, with the CPU baseline leg), rocprofv3 kernel stats, PMC passes, train-/score-only lines.
# usage: scripts/gpu_round_profile.sh <tag>      -> gpurun_out/{bench_<tag>.json, prof_<tag>/, pmc_<tag>/, ...}
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-x}"
cd "$REPO"; mkdir -p gpurun_out
json_line() { grep '^{' "$1" | tail -1 > "$2"; }
timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2>&1; echo "bench exit: $?"
json_line gpurun_out/bench_$TAG.log gpurun_out/bench_$TAG.json
timeout 300 python bench.py --no-cpu-baseline --train-only > gpurun_out/bench_${TAG}_train_only.log 2>&1
json_line gpurun_out/bench_${TAG}_train_only.log gpurun_out/bench_${TAG}_train_only.json
timeout 300 python bench.py --no-cpu-baseline --score-only > gpurun_out/bench_${TAG}_score_only.log 2>&1
json_line gpurun_out/bench_${TAG}_score_only.log gpurun_out/bench_${TAG}_score_only.json
for cd in 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --no-alt --compute-dtype $cd > gpurun_out/bench_${TAG}_dtype$cd.log 2>&1
  json_line gpurun_out/bench_${TAG}_dtype$cd.log gpurun_out/bench_${TAG}_dtype$cd.json
done
bash scripts/gpu_profile.sh $TAG > gpurun_out/prof_$TAG.txt 2>&1
bash scripts/gpu_pmc.sh $TAG > gpurun_out/pmc_$TAG.txt 2>&1
python - <<PY
import json
for f in ("bench_$TAG", "bench_${TAG}_train_only", "bench_${TAG}_score_only", "bench_${TAG}_dtype2", "bench_${TAG}_dtype3"):
    try:
        d = json.load(open("gpurun_out/%s.json" % f))
        print(f, d["value"], d["ms_per_step"], d.get("roofline"), d.get("cpu_baseline"))
    except Exception as e:
        print(f, "unreadable", e)
PY
head -12 gpurun_out/prof_$TAG/kernel_stats.csv
# the other configurations' lines (parity-tested configs, not the headline): dims B, shipped, configs[3] bf16 / fp32, configs[4], DP path
extra() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --batch-feed resident "$@" > gpurun_out/bench_${TAG}_$name.log 2>&1
  json_line gpurun_out/bench_${TAG}_$name.log gpurun_out/bench_${TAG}_$name.json
  python -c "
import json; d=json.load(open('gpurun_out/bench_${TAG}_$name.json')); print('$name', d['value'], d['ms_per_step'], d.get('roofline'))" || tail -5 gpurun_out/bench_${TAG}_$name.log; }
extra dimsB --dims B --steps 8 --warmup 3
extra shipped --dims shipped --steps 8 --warmup 3
extra c4_bf16 --dims C4 --steps 8 --warmup 3
extra c4_fp32 --dims C4 --c4-fp32 --steps 6 --warmup 2
extra force_dp --force-dp
timeout 300 python bench.py --workload c5 --steps 40 --warmup 5 > gpurun_out/bench_${TAG}_c5.log 2>&1; json_line gpurun_out/bench_${TAG}_c5.log gpurun_out/bench_${TAG}_c5.json
python -c "
import json; d=json.load(open('gpurun_out/bench_${TAG}_c5.json')); print('c5', d['value'], d['ms_per_step'], d.get('streaming'), d.get('roofline'), d.get('cpu_baseline'))"
timeout 300 python scripts/gpu_dp_sim.py > gpurun_out/dp_sim_$TAG.json 2> gpurun_out/dp_sim_$TAG.log; tail -c 1500 gpurun_out/dp_sim_$TAG.json
timeout 300 python scripts/gpu_gemm_bench.py > gpurun_out/gemm_bench_$TAG.txt 2>&1; tail -30 gpurun_out/gemm_bench_$TAG.txt
# the reference-shaped host loops end to end (files -> Batcher -> MyOptimizer / test_from_checkpoint -> engine)
python scripts/gpu_train_epoch.py 2>&1 | tail -1 > gpurun_out/train_epoch_$TAG.json; cat gpurun_out/train_epoch_$TAG.json
python scripts/gpu_score_files.py 400000 2>&1 | tail -1 > gpurun_out/score_files_$TAG.json; cat gpurun_out/score_files_$TAG.json
