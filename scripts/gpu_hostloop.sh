#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_feed.py tests/test_gpu_host.py tests/test_dp_gloo.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -v "^    " | tail -25
python scripts/gpu_train_epoch.py 2>&1 | tail -1 | tee gpurun_out/train_epoch.json
python scripts/gpu_score_files.py 2>&1 | tail -1 | tee gpurun_out/score_files.json
