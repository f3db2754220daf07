#!/bin/bash
# rocprofv3 PMC passes over the default bench command (counters in their own runs, kernel-trace only;
# never combined with sys/hip/hsa traces).  Output -> gpurun_out/pmc_<tag>/{fetch,write,sq}.csv + summary.json
# FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950 (TCC has 4 slots: 3 + 2).
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r01}"; shift
OUT="$REPO/gpurun_out/pmc_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|^.*SQ_BUSY_CU|SQ_WAVE_CYCLES|SQ_ACTIVE_INST_VALU|SQ_INSTS_VALU " | head -60 > "$OUT/counters_available.txt"
pass() {  # name, counters...
  local name="$1"; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/raw_$name" -o p --output-format csv -- \
      python "$REPO/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-extra-regions --batch-feed resident --no-kernel-events --no-other-configs --no-batch-sweep "${EXTRA[@]}" > "$OUT/$name.log" 2>&1
  echo "rocprof exit: $?" >> "$OUT/$name.log"
  find "$OUT/raw_$name" -name "*counter_collection.csv" -exec cp {} "$OUT/$name.csv" \;
  rm -rf "$OUT/raw_$name"
}
EXTRA=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES ${MOPS:-SQ_INSTS_VALU_MFMA_MOPS_F32}
python "$REPO/scripts/pmc_summary.py" "$OUT" > "$OUT/kernels.json"
python - "$OUT" "$@" <<PY > "$OUT/summary.json"
import json, sys
d = json.load(open(sys.argv[1] + "/kernels.json"))
pps = 65536
args = sys.argv[2:]
if "--paths-per-step" in args: pps = int(args[args.index("--paths-per-step") + 1])
print(json.dumps({"paths_per_step": pps, "command": "python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-extra-regions --batch-feed resident --no-kernel-events " + " ".join(args), "kernels": d}, indent=1, sort_keys=True))
PY
cat "$OUT/summary.json"
# the raw per-dispatch csv files are large; keep the summary + a head
for f in fetch write sq; do [ -f "$OUT/$f.csv" ] && { head -3 "$OUT/$f.csv" > "$OUT/$f.head.csv"; rm "$OUT/$f.csv"; }; done
