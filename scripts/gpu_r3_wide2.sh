#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-b}"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py tests/test_gpu_persist.py -x -q -m gpu -k "not union and not replicas" > gpurun_out/wide_${TAG}_tests.log 2>&1; grep -a "passed\|failed" gpurun_out/wide_${TAG}_tests.log | tail -2
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/wide_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/wide_${TAG}_$name.log | tail -1 > gpurun_out/wide_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/wide_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/wide_${TAG}_$name.json')); print('$name', d['value'], d['ms_per_step'], d.get('mfma_frac_end_to_end'))
k=d['kernels']
for n,v in sorted(k.items(), key=lambda x:-x[1]['ms'])[:12]: print('   %-28s %.4f %d %s' % (n, v['ms'], v['launches'], v.get('from')))
PY
}
run shipped --dims shipped --steps 8 --warmup 2
run dimsB --dims B --steps 8 --warmup 2
run c4 --dims C4 --steps 8 --warmup 2
