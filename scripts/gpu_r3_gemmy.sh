#!/bin/bash
# round 3: the 256 x 192 bf16 product (k_gemm16y) against the 256 x 128 one (KPRN_BF16_GEMM=x) on configs[3]'s shapes, parity tests, the C4 step
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-a}"
timeout 300 python scripts/gpu_gemm16_bench.py > gpurun_out/gemmy_${TAG}_y.txt 2>&1; cat gpurun_out/gemmy_${TAG}_y.txt | tail -6
KPRN_BF16_GEMM=x timeout 300 python scripts/gpu_gemm16_bench.py > gpurun_out/gemmy_${TAG}_x.txt 2>&1; cat gpurun_out/gemmy_${TAG}_x.txt | tail -6
timeout 900 python -m pytest tests/test_gpu_persist.py -x -q -m gpu > gpurun_out/gemmy_${TAG}_tests.log 2>&1; grep -a "passed\|failed" gpurun_out/gemmy_${TAG}_tests.log | tail -2
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/gemmy_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/gemmy_${TAG}_$name.log | tail -1 > gpurun_out/gemmy_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/gemmy_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/gemmy_${TAG}_$name.json')); k=d['kernels']
print('$name', d['value'], d['ms_per_step'], {n: round(v['ms']/max(1,v['launches']),4) for n,v in k.items() if 'gemm' in n})
PY
}
run c4_y --dims C4 --steps 6 --warmup 2
KPRN_BF16_GEMM=x run c4_x --dims C4 --steps 6 --warmup 2
