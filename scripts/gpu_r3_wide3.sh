#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-c}"
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/wide_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/wide_${TAG}_$name.log | tail -1 > gpurun_out/wide_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/wide_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/wide_${TAG}_$name.json')); k=d['kernels']
print('$name', d['value'], d['ms_per_step'], 'scatter', k.get('embed_scatter'), 'entity', k.get('entity_grad'))
PY
}
for cfg in shipped C4 B; do
KPRN_TABLE_GRAD=old run ${cfg}_old --dims $cfg --steps 4 --warmup 2
for P in 128 256 512 1024 2048; do KPRN_TABLE_GRAD_PPB=$P run ${cfg}_ppb$P --dims $cfg --steps 4 --warmup 2; done
done
