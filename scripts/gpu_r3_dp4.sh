#!/bin/bash
# round 3: the engine's own exchange (kprn_dp_*) against the torch.distributed hooks and the plain step; kernel timelines of each
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-e}"
timeout 900 python -m pytest tests/test_gpu_host.py -x -q -m gpu > gpurun_out/dp_${TAG}_tests.log 2>&1; grep -a "passed\|failed" gpurun_out/dp_${TAG}_tests.log | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "union or gathered or replicas or side_stream" > gpurun_out/dp_${TAG}_tests2.log 2>&1; grep -a "passed\|failed" gpurun_out/dp_${TAG}_tests2.log | tail -2
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/dp_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/dp_${TAG}_$name.log | tail -1 > gpurun_out/dp_${TAG}_$name.json
  python -c "
import json; d=json.load(open('gpurun_out/dp_${TAG}_$name.json')); print('$name', d['value'], d['ms_per_step'], json.dumps(d.get('dp'))[:900])" || tail -5 gpurun_out/dp_${TAG}_$name.log; }
run plain --steps 60 --warmup 10
run force_dp --force-dp --steps 60 --warmup 10
run force_dp_under_gather --force-dp --dp-score-under-gather --steps 60 --warmup 10
run force_dp_torch --force-dp --dp-torch-collectives --steps 60 --warmup 10
run force_dp_torch_under_gather --force-dp --dp-torch-collectives --dp-score-under-gather --steps 60 --warmup 10
run plain2 --steps 60 --warmup 10
bash scripts/gpu_timeline.sh --force-dp --no-other-configs --no-batch-sweep > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt gpurun_out/dp_${TAG}_timeline_force_dp.txt
bash scripts/gpu_timeline.sh --force-dp --dp-torch-collectives --no-other-configs --no-batch-sweep > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt gpurun_out/dp_${TAG}_timeline_force_dp_torch.txt
head -18 gpurun_out/dp_${TAG}_timeline_force_dp.txt
