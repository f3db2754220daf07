#!/bin/bash
# ONE parameterised profiling job (replaces the per-round / per-experiment profile scripts): for one state of the code and one bench command line,
#   1. rocprofv3 --kernel-trace --stats            -> rocprofv3_kernel_stats_<tag>.csv   (per-kernel average durations: must agree with bench.py's HIP events)
#   2. three rocprofv3 --pmc passes, kernel-trace only (counters in their own runs, never combined with sys / hip / hsa traces; FETCH_SIZE and
#      WRITE_SIZE cannot share a pass on gfx950)   -> pmc_summary_<tag>.json             (HBM bytes per launch with the guide's gfx950 correction,
#                                                                                          MFMA busy per SIMD, MFMA ops)
# into gpurun_out/<round>/ -- copy what DESIGN.md cites into profiles/<round>/.
#   usage: scripts/gpu_profile.sh <round> <tag> [bench.py flags ...]        e.g.  scripts/gpu_profile.sh r04 c4 --dims C4
#   env:   MOPS=<counter>  the MFMA-ops counter of the third pass (default SQ_INSTS_VALU_MFMA_MOPS_F32; configs[3]: SQ_INSTS_VALU_MFMA_MOPS_BF16)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
ROUND="${1:?round}"; TAG="${2:?tag}"; shift 2
OUT="$REPO/gpurun_out/$ROUND"; mkdir -p "$OUT"
QUICK="--steps 8 --warmup 3 --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident"
cd /tmp && export TMPDIR=/tmp
RAW="/tmp/prof_${ROUND}_${TAG}"; rm -rf "$RAW"
timeout 900 rocprofv3 --kernel-trace --stats -d "$RAW/stats" -o p --output-format csv -- python "$REPO/bench.py" $QUICK "$@" > "$OUT/bench_under_rocprof_$TAG.log" 2>&1
echo "rocprof exit: $?" >> "$OUT/bench_under_rocprof_$TAG.log"
find "$RAW/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/rocprofv3_kernel_stats_$TAG.csv" \;
head -12 "$OUT/rocprofv3_kernel_stats_$TAG.csv"
pass() {  # name, counters...
  local name="$1"; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d "$RAW/$name" -o p --output-format csv -- \
      python "$REPO/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident --no-kernel-events "${EXTRA[@]}" > "$RAW/$name.log" 2>&1
  find "$RAW/$name" -name "*counter_collection.csv" -exec cp {} "$RAW/$name.csv" \;
}
EXTRA=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES ${MOPS:-SQ_INSTS_VALU_MFMA_MOPS_F32}
python "$REPO/scripts/pmc_summary.py" "$RAW" > "$RAW/kernels.json"
python - "$RAW" "$@" <<PY > "$OUT/pmc_summary_$TAG.json"
import json, sys
d = json.load(open(sys.argv[1] + "/kernels.json"))
pps = 65536
args = sys.argv[2:]
if "--paths-per-step" in args: pps = int(args[args.index("--paths-per-step") + 1])
print(json.dumps({"paths_per_step": pps, "command": "python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident --no-kernel-events " + " ".join(args), "kernels": d}, indent=1, sort_keys=True))
PY
python - "$OUT/pmc_summary_$TAG.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0))[:8]:
    print("%-28s %8.1f MB/launch  mfma busy/SIMD %s" % (k, v.get("hbm_bytes_per_launch", 0) / 1e6, round(v.get("mfma_busy_frac_per_simd", 0), 3)))
PY
rm -rf "$RAW"
