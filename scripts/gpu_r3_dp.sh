#!/bin/bash
# round 3: the data-parallel step at world 1 over RCCL against the plain step, same call
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-a}"
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --batch-feed resident "$@" > gpurun_out/dp_${TAG}_$name.log 2>&1
  grep '^{' gpurun_out/dp_${TAG}_$name.log | tail -1 > gpurun_out/dp_${TAG}_$name.json
  python -c "
import json; d=json.load(open('gpurun_out/dp_${TAG}_$name.json')); print('$name', d['value'], d['ms_per_step'], json.dumps(d.get('dp'))[:700])" || tail -5 gpurun_out/dp_${TAG}_$name.log; }
run plain --steps 60 --warmup 10
run force_dp --force-dp --steps 60 --warmup 10
run force_dp_score_first --force-dp --dp-score-first --steps 60 --warmup 10
run plain2 --steps 60 --warmup 10
timeout 300 python scripts/gpu_dp_sim.py > gpurun_out/dp_sim_$TAG.json 2> gpurun_out/dp_sim_$TAG.log; tail -c 1200 gpurun_out/dp_sim_$TAG.json
