#!/bin/bash
# one gpurun call: full GPU suite (compact report) + the default bench line.   usage: scripts/gpu_all.sh <tag> [pytest args]
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; TAG="${1:-x}"; shift
cd "$REPO"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider "$@" 2>&1 | grep -v "^E    \|^    " | tail -40 > gpurun_out/tests_$TAG.txt
tail -15 gpurun_out/tests_$TAG.txt
timeout 600 python bench.py > gpurun_out/bench_$TAG.log 2>&1; echo "bench exit: $?"
grep '^{' gpurun_out/bench_$TAG.log | tail -1 > gpurun_out/bench_$TAG.json
python - <<PY
import json
try:
    d = json.load(open("gpurun_out/bench_$TAG.json"))
    for k in ("value", "ms_per_step", "roofline", "streaming", "long_run", "no_prefix_plan", "cpu_baseline", "alt_f32x6"):
        print(k, d.get(k))
    print({k: v for k, v in d["kernels"].items()})
except Exception as e:
    print("bench unreadable", e)
    print(open("gpurun_out/bench_$TAG.log").read()[-3000:])
PY
