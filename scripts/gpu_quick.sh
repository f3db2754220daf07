#!/bin/bash
# quick check of the fused path: a parity subset + score-only / train-only / default bench lines.   usage: scripts/gpu_quick.sh <tag>
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; TAG="${1:-q}"; cd "$REPO"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "not five_adam" 2>&1 | grep -v "^    \|^E   " | tail -6
for mode in --score-only --train-only ""; do
  timeout 300 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --batch-feed resident $mode > gpurun_out/q_$TAG.log 2>&1
  grep '^{' gpurun_out/q_$TAG.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('$mode', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], {n:round(v['ms']/max(1,v['launches']),4) for n,v in k.items() if 'lstm' in n})" || tail -5 gpurun_out/q_$TAG.log
done
