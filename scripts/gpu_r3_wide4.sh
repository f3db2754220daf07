#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-d}"
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/wide_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/wide_${TAG}_$name.log | tail -1 > gpurun_out/wide_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/wide_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/wide_${TAG}_$name.json')); k=d['kernels']
print('$name', d['value'], d['ms_per_step'], 'scatter', k.get('embed_scatter'), 'entity', k.get('entity_grad'))
PY
}
for G in 0 1 2 4 3 7; do KPRN_TABLE_GRAD_PPB=512 KPRN_TABLE_GRAD_DBG=$G run C4_dbg$G --dims C4 --steps 3 --warmup 2; done
