#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-a}"
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/c4d_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/c4d_${TAG}_$name.log | tail -1 > gpurun_out/c4d_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/c4d_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/c4d_${TAG}_$name.json')); k=d['kernels']
print('$name', d['value'], d['ms_per_step'], 'gates', round(k['lstm_gates_bwd_bf16']['ms']/k['lstm_gates_bwd_bf16']['launches'],4))
PY
}
for U in 1 2 4 8; do KPRN_GATES_UPW=$U run upw$U --dims C4 --steps 4 --warmup 2; done
for U in 1 8; do KPRN_GATES_NOBIAS=1 KPRN_GATES_UPW=$U run nobias_upw$U --dims C4 --steps 4 --warmup 2; done
