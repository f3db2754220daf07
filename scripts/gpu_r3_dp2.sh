#!/bin/bash
# round 3: the data-parallel step with the union inside the row update (dp_fused_update) against the separate merge and the plain step
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-c}"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "union or gathered or replicas" > gpurun_out/dp_${TAG}_tests.log 2>&1; tail -3 gpurun_out/dp_${TAG}_tests.log
timeout 600 python -m pytest tests/test_gpu_host.py tests/test_gpu_persist.py -x -q -m gpu > gpurun_out/dp_${TAG}_tests2.log 2>&1; tail -3 gpurun_out/dp_${TAG}_tests2.log
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/dp_${TAG}_$name.log 2>&1
  grep '^{' gpurun_out/dp_${TAG}_$name.log | tail -1 > gpurun_out/dp_${TAG}_$name.json
  python -c "
import json; d=json.load(open('gpurun_out/dp_${TAG}_$name.json')); print('$name', d['value'], d['ms_per_step'], json.dumps(d.get('dp'))[:700])" || tail -5 gpurun_out/dp_${TAG}_$name.log; }
run plain --steps 60 --warmup 10
run force_dp --force-dp --steps 60 --warmup 10
run force_dp_unfused --force-dp --dp-unfused --steps 60 --warmup 10
run force_dp_score_first --force-dp --dp-score-first --steps 60 --warmup 10
run plain2 --steps 60 --warmup 10
for F in 1 0; do FUSED=$F timeout 300 python scripts/gpu_dp_sim.py > gpurun_out/dp_sim_${TAG}_fused$F.json 2> gpurun_out/dp_sim_${TAG}_fused$F.log; tail -c 900 gpurun_out/dp_sim_${TAG}_fused$F.json; done
bash scripts/gpu_timeline.sh --force-dp --no-other-configs --no-batch-sweep > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt gpurun_out/dp_${TAG}_timeline_force_dp.txt
tail -60 gpurun_out/dp_${TAG}_timeline_force_dp.txt
