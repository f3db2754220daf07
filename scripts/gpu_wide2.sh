#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -v "^    \|^E   " | tail -25
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --batch-feed resident --steps 8 --warmup 3 "$@" > gpurun_out/w_$name.log 2>&1
  grep '^{' gpurun_out/w_$name.log | tail -1 > gpurun_out/w_$name.json
  python - <<PY || tail -5 gpurun_out/w_$name.log
import json
d = json.load(open("gpurun_out/w_$name.json"))
print("$name", d["value"], d["ms_per_step"], d.get("roofline"))
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms"])[:10]:
    print("   %-24s %8.4f ms x %3d = %8.3f ms" % (k, v["ms"] / max(1, v["launches"]), v["launches"], v["ms"]))
PY
}
run dimsB --dims B
run shipped --dims shipped
run c4_fp32 --dims C4 --c4-fp32 --steps 4 --warmup 2
timeout 300 python scripts/gpu_gemm_bench.py 2>&1 | tail -12
