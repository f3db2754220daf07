#!/bin/bash
# perf-only loop on the GPU box: bench summary (per-kernel avg ms), optional KPRN_TIMING breakdown
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
[ -n "$TIMING" ] && KPRN_TIMING=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-only 2>&1 | grep "kprn timing" | tail -2
timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/bench.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]); print({k:(round(v["ms"]/v["launches"],4), v["launches"]) for k,v in d["kernels"].items()})
PY
