#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-f}"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "union or gathered or replicas or side_stream" > gpurun_out/dp_${TAG}_tests2.log 2>&1; grep -a "passed\|failed" gpurun_out/dp_${TAG}_tests2.log | tail -2
bash scripts/gpu_timeline.sh --force-dp --no-other-configs --no-batch-sweep > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt gpurun_out/dp_${TAG}_timeline_force_dp.txt
bash scripts/gpu_timeline.sh --no-other-configs --no-batch-sweep > /dev/null 2>&1; cp gpurun_out/timeline/timeline.txt gpurun_out/dp_${TAG}_timeline_plain.txt
head -16 gpurun_out/dp_${TAG}_timeline_force_dp.txt; head -14 gpurun_out/dp_${TAG}_timeline_plain.txt
