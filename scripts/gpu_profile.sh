#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command. Output -> gpurun_out/prof_<tag>/
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r01}"; shift
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/raw" -o bench --output-format csv -- python "$REPO/bench.py" --steps 12 --warmup 3 --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --batch-feed resident "$@" > "$OUT/bench_under_rocprof.log" 2>&1
echo "rocprof exit: $?" >> "$OUT/bench_under_rocprof.log"
find "$OUT/raw" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/raw" -name "*kernel_trace.csv" | head -1 | xargs -I{} sh -c 'head -1 {} > '"$OUT"'/kernel_trace_head.csv; wc -l {} >> '"$OUT"'/kernel_trace_head.csv'
rm -rf "$OUT/raw"
head -25 "$OUT/kernel_stats.csv"
