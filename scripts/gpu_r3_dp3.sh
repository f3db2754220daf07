#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-d}"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "union or gathered or replicas" > gpurun_out/dp_${TAG}_tests.log 2>&1; tail -3 gpurun_out/dp_${TAG}_tests.log
bash scripts/gpu_timeline.sh --force-dp --no-other-configs --no-batch-sweep > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt gpurun_out/dp_${TAG}_timeline_force_dp.txt
bash scripts/gpu_timeline.sh --force-dp --dp-score-first --no-other-configs --no-batch-sweep > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt gpurun_out/dp_${TAG}_timeline_force_dp_score_first.txt
bash scripts/gpu_timeline.sh --no-other-configs --no-batch-sweep > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt gpurun_out/dp_${TAG}_timeline_plain.txt
wc -l gpurun_out/dp_${TAG}_timeline_*.txt
