#!/bin/bash
# kernel timeline of steady-state steps of the default workload: start offset / duration / stream of every dispatch (rocprofv3 kernel trace)
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd /tmp && export TMPDIR=/tmp
OUT=$REPO/gpurun_out/timeline; mkdir -p $OUT
KPRN_BENCH_SOFT_EXIT=1 timeout 600 rocprofv3 --kernel-trace -d $OUT/raw -o t --output-format csv -- python $REPO/bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-alt --no-extra-regions --batch-feed resident --no-kernel-events "$@" > $OUT/log.txt 2>&1
F=$(find $OUT/raw -name "*kernel_trace.csv" | head -1)
python - "$F" > $OUT/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last full steps: find k_adam_rows occurrences (two per step: update, then next step's catch-up)
idx = [i for i, n in enumerate(names) if "k_lstm_fwd<2, true" in n or "k_lstm_fwdILi2ELb1" in n or "k_lstm_fwd_dual" in n]
if len(idx) < 4: print("few steps", len(idx)); sys.exit(0)
m = len(idx) // 2                 # (the middle of the trace = the timed region; the last steps of a --force-dp run belong to the region with events around the exchange)
a, b = idx[m], idx[m + 2]         # two whole steps, from one training forward to the one after next
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("(")[0][-46:]
    print("%9.1f us  +%7.1f us  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), nm))
    prev_end = max(prev_end, e)
PY
rm -rf $OUT/raw
tail -70 $OUT/timeline.txt
