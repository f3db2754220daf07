#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-h}"
timeout 900 python -m pytest tests/test_gpu_host.py -x -q -m gpu > gpurun_out/dp_${TAG}_tests.log 2>&1; grep -a "passed\|failed" gpurun_out/dp_${TAG}_tests.log | tail -2
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/dp_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/dp_${TAG}_$name.log | tail -1 > gpurun_out/dp_${TAG}_$name.json
  python -c "
import json; d=json.load(open('gpurun_out/dp_${TAG}_$name.json')); print('$name', d['value'], d['ms_per_step'], json.dumps(d.get('dp'))[:600])" || tail -5 gpurun_out/dp_${TAG}_$name.log; }
run plain --steps 60 --warmup 10
run force_dp --force-dp --steps 60 --warmup 10
run force_dp_under_gather --force-dp --dp-score-under-gather --steps 60 --warmup 10
run plain2 --steps 60 --warmup 10
run force_dp2 --force-dp --steps 60 --warmup 10
