#!/bin/bash
# round 3: the generic pipeline's helper kernels (gather with the fused MaskZero mask, column sums) on the shipped config and dims B
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-a}"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide.py -x -q -m gpu -k "rnn or embedding or gather or shipped or odd or ablation or gru or wide or config4 or d128" > gpurun_out/wide_${TAG}_tests.log 2>&1; grep -a "passed\|failed" gpurun_out/wide_${TAG}_tests.log | tail -2
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/wide_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/wide_${TAG}_$name.log | tail -1 > gpurun_out/wide_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/wide_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/wide_${TAG}_$name.json')); print('$name', d['value'], d['ms_per_step'], d.get('mfma_frac_end_to_end'))
k=d['kernels']
for n,v in sorted(k.items(), key=lambda x:-x[1]['ms'])[:16]: print('   %-28s %.4f %d %s' % (n, v['ms'], v['launches'], v.get('from')))
PY
}
run shipped --dims shipped --steps 8 --warmup 2
run dimsB --dims B --steps 8 --warmup 2
