#!/bin/bash
# One parameterised GPU job runner (replaces the per-experiment gpu_r3_*.sh scripts): every step writes gpurun_out/<tag>_<name>.{log,json}.
#   scripts/gpu_job.sh <tag> "<step>" ["<step>" ...]
# step grammar (first word):
#   t  <name> <pytest args...>            pytest -m gpu, compact result line
#   b  <name> <bench.py args...>          bench.py with the quick flags (no CPU baseline / alt / extra regions / other configs / sweep, resident feed)
#   B  <name> <bench.py args...>          bench.py exactly as given (the driver's command when no args)
#   py <name> <script> <args...>          python <script> <args>
#   prof <name> <bench.py args...>        rocprofv3 --kernel-trace --stats of the quick bench, kernel stats csv copied next to the log
# a step may be prefixed by VAR=VALUE words (environment for that step only)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="$1"; shift
QUICK="--no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident"
for step in "$@"; do
  set -- $step
  envs=()
  while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
  kind="$1"; name="$2"; shift 2
  out="gpurun_out/${TAG}_${name}"
  case "$kind" in
    t) env "${envs[@]}" timeout 1500 python -m pytest -m gpu -q -p no:cacheprovider "$@" > "$out.log" 2>&1
       echo "[$name] $(grep -a 'passed\|failed\|error' "$out.log" | tail -1)"; grep -a "^FAILED\|^ERROR\|Error\|assert " "$out.log" | head -12 ;;
    b|B) if [ "$kind" = b ]; then extra="$QUICK"; else extra=""; fi
       env "${envs[@]}" timeout 1200 python bench.py $extra "$@" > "$out.log" 2>&1
       grep -a '^{' "$out.log" | tail -1 > "$out.json"
       python - "$out.json" "$name" <<'PY' || tail -5 "$out.log"
import json, sys
d = json.load(open(sys.argv[1])); k = d.get('kernels', {})
print(f"[{sys.argv[2]}]", d['value'], d['ms_per_step'], 'roofline', d.get('roofline', {}).get('frac'),
      {n: round(v['ms'] / max(1, v['launches']), 4) for n, v in sorted(k.items(), key=lambda kv: -kv[1]['ms'])[:14]})
PY
       ;;
    py) env "${envs[@]}" timeout 1500 python "$@" > "$out.log" 2>&1; tail -${TAILN:-30} "$out.log" ;;
    prof) cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_$name
       env "${envs[@]}" timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o p --output-format csv -- python "$REPO/bench.py" $QUICK "$@" > "$REPO/$out.log" 2>&1
       f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$REPO/$out.kernel_stats.csv" && head -12 "$f"
       cd "$REPO" ;;
    *) echo "unknown step kind $kind" ;;
  esac
done
