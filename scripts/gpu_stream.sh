#!/bin/bash
# streaming-feed experiments: resident vs streaming, feed-stream priority on/off, kernel trace of the streaming run
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; TAG="${1:-x}"
cd "$REPO"; mkdir -p gpurun_out
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --steps 60 --warmup 8 "$@" > gpurun_out/st_${TAG}_$name.log 2>&1
  grep '^{' gpurun_out/st_${TAG}_$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], (d.get('streaming') or {}).get('value'))" || tail -5 gpurun_out/st_${TAG}_$name.log; }
timeout 600 python -m pytest tests/test_gpu_feed.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | grep -v "^    " | tail -15
run resident --batch-feed resident
run streaming_host --batch-feed streaming
run streaming_host_a2 --batch-feed streaming --feed-ahead 2
run streaming_host_t2 --batch-feed streaming --feed-threads 2
run streaming_host_t16 --batch-feed streaming --feed-threads 16
run streaming_device --batch-feed streaming --feed-build device --feed-ahead 2
run streaming_trainonly --batch-feed streaming --train-only
run resident_trainonly --batch-feed resident --train-only
KPRN_FEED_TIMING=1 timeout 120 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --steps 6 --warmup 2 --batch-feed streaming 2>&1 | grep "kprn feed" | tail -12
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_st_$TAG -o p --output-format csv -- python $REPO/bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-kernel-events --steps 30 --warmup 6 --batch-feed streaming > $REPO/gpurun_out/st_${TAG}_prof.log 2>&1
cd $REPO
f=$(find gpurun_out/prof_st_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > gpurun_out/st_${TAG}_kernel_stats.csv && head -30 gpurun_out/st_${TAG}_kernel_stats.csv | cut -c1-150
t=$(find gpurun_out/prof_st_$TAG -name "*kernel_trace.csv" | head -1)
[ -n "$t" ] && python - "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last ~3 steps: timeline of kernels with stream/queue, start offset and duration (us)
t_end = int(rows[-1]["End_Timestamp"])
t_mid = (int(rows[0]["Start_Timestamp"]) + 3 * t_end) // 4
sel = [r for r in rows if t_mid < int(r["Start_Timestamp"]) < t_mid + 3_600_000]
t0 = int(sel[0]["Start_Timestamp"])
for r in sel:
    print("%9.1f %8.1f q%-4s %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
PY
rm -rf gpurun_out/prof_st_$TAG
