#!/bin/bash
# quick perf+correctness loop on the GPU box: gpu tests (-x), per-stage timing, bench summary
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -3
KPRN_TIMING=1 timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-only 2>&1 | grep "kprn timing" | tail -2
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/bench.log 2>&1
python - <<PY
import json
d=json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]); print({k:(round(v["ms"]/v["launches"],4), v["launches"]) for k,v in d["kernels"].items()})
PY
