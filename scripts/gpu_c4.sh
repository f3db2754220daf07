#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_wide.py -x -q -m gpu -k "bf16 or 2_pow_24" 2>&1 | tail -15
run() { name=$1; shift; timeout 900 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --batch-feed resident --steps 8 --warmup 3 "$@" > gpurun_out/c4_$name.log 2>&1
  grep '^{' gpurun_out/c4_$name.log | tail -1 > gpurun_out/c4_$name.json
  python - <<PY || tail -8 gpurun_out/c4_$name.log
import json
d = json.load(open("gpurun_out/c4_$name.json"))
print("$name", d["value"], d["ms_per_step"], d.get("roofline"))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:14]:
    print("   %-24s %8.4f ms x %3d = %8.3f ms" % (k, v["ms"] / max(1, v["launches"]), v["launches"], v["ms"]))
PY
}
run bf16 --dims C4
KPRN_BF16_TILE=small run bf16small --dims C4
