#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, smoke, a short bench. Logs -> gpurun_out/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | tail -${PYTEST_TAIL:-120} > gpurun_out/pytest.log
echo "pytest exit: $?" >> gpurun_out/pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps ${BENCH_STEPS:-12} --warmup 3 ${BENCH_ARGS} > gpurun_out/bench.log 2>&1
echo "bench exit: $?" >> gpurun_out/bench.log
tail -5 gpurun_out/pytest.log; tail -3 gpurun_out/smoke.log; tail -3 gpurun_out/bench.log
