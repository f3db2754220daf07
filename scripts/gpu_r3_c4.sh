#!/bin/bash
# round 3: the persistent bf16 layer kernel (configs[3]) -- hardware probe, parity tests, bench with and without it
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-a}"
scripts/ubench/build/mfma32_probe
timeout 900 python -m pytest tests/test_gpu_persist.py -x -q -m gpu 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_wide.py -x -q -m gpu -k "bf16" 2>&1 | tail -5
run() { name=$1; shift; timeout 900 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --batch-feed resident --steps 8 --warmup 3 "$@" > gpurun_out/c4_${TAG}_$name.log 2>&1
  grep '^{' gpurun_out/c4_${TAG}_$name.log | tail -1 > gpurun_out/c4_${TAG}_$name.json
  python - <<PY || tail -8 gpurun_out/c4_${TAG}_$name.log
import json
d = json.load(open("gpurun_out/c4_${TAG}_$name.json"))
print("$name", d["value"], d["ms_per_step"], json.dumps(d.get("roofline")))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:16]:
    print("   %-28s %8.4f ms x %3d = %8.3f ms" % (k, v["ms"] / max(1, v["launches"]), v["launches"], v["ms"]))
PY
}
run persist --dims C4
run persist_score --dims C4 --score-only
KPRN_BF16_PERSIST=0 run steps --dims C4
