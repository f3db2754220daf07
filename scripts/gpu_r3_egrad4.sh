#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-a}"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "backward or train or fullsize or full_size or prefix or variants or additive" > gpurun_out/eg4_${TAG}_tests.log 2>&1; grep -a "passed\|failed" gpurun_out/eg4_${TAG}_tests.log | tail -2
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/eg4_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/eg4_${TAG}_$name.log | tail -1 > gpurun_out/eg4_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/eg4_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/eg4_${TAG}_$name.json')); k=d['kernels']
print('$name', d['value'], d['ms_per_step'], {n: round(v['ms']/max(1,v['launches']),4) for n,v in k.items() if 'entity' in n})
PY
}
run a --steps 40 --warmup 5
run b --steps 40 --warmup 5
KPRN_SG_BLOCKS=1536 run sg1536 --steps 40 --warmup 5
run c4 --dims C4 --steps 6 --warmup 2
