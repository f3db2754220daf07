#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-a}"
timeout 900 python -m pytest tests/test_gpu_persist.py tests/test_gpu_wide.py -x -q -m gpu > gpurun_out/c4c_${TAG}_tests.log 2>&1; grep -a "passed\|failed" gpurun_out/c4c_${TAG}_tests.log | tail -2
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/c4c_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/c4c_${TAG}_$name.log | tail -1 > gpurun_out/c4c_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/c4c_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/c4c_${TAG}_$name.json')); k=d['kernels']
print('$name', d['value'], d['ms_per_step'])
for n,v in sorted(k.items(), key=lambda x:-x[1]['ms'])[:14]: print('   %-28s %.4f %d' % (n, v['ms']/max(1,v['launches']), v['launches']))
PY
}
run c4 --dims C4 --steps 8 --warmup 2
run c4b --dims C4 --steps 8 --warmup 2
