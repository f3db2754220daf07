#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-e}"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_persist.py -x -q -m gpu -k "not union and not replicas" > gpurun_out/wide_${TAG}_tests.log 2>&1; grep -a "passed\|failed" gpurun_out/wide_${TAG}_tests.log | tail -2
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/wide_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/wide_${TAG}_$name.log | tail -1 > gpurun_out/wide_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/wide_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/wide_${TAG}_$name.json')); k=d['kernels']
print('$name', d['value'], d['ms_per_step'], 'scatter', k.get('embed_scatter'), 'entity', k.get('entity_grad'))
PY
}
for cfg in shipped C4 B; do
run ${cfg}_mfma --dims $cfg --steps 6 --warmup 2
KPRN_TABLE_GRAD=lds run ${cfg}_lds --dims $cfg --steps 6 --warmup 2
KPRN_TABLE_GRAD_PPB=256 run ${cfg}_mfma256 --dims $cfg --steps 6 --warmup 2
KPRN_TABLE_GRAD_PPB=1024 run ${cfg}_mfma1024 --dims $cfg --steps 6 --warmup 2
done
