#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-a}"
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/eg_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/eg_${TAG}_$name.log | tail -1 > gpurun_out/eg_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/eg_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/eg_${TAG}_$name.json')); k=d['kernels']
print('$name', d['value'], d['ms_per_step'], {n: round(v['ms']/max(1,v['launches']),4) for n,v in k.items() if 'entity' in n})
PY
}
for B in 1024 512 256 128; do for O in 0 16; do KPRN_SG_BLOCKS=$B KPRN_EGRAD_DBG=$O run sg${B}_o$O --steps 30 --warmup 5; done; done
