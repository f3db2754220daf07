#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-a}"
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/c4b_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/c4b_${TAG}_$name.log | tail -1 > gpurun_out/c4b_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/c4b_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/c4b_${TAG}_$name.json')); k=d['kernels']
print('$name', d['value'], d['ms_per_step'], {n: round(v['ms']/max(1,v['launches']),4) for n,v in k.items() if 'dw' in n})
PY
}
for S in 5 0 7 6 8 0; do KPRN_GEMM16_SX=$S run sx$S --dims C4 --steps 6 --warmup 2; done
