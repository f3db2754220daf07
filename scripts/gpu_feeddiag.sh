#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
FEED="${FEED:-host}"; AH="${AH:-4}"
KPRN_BENCH_HOST_TIMING=1 timeout 300 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --steps 40 --warmup 8 --batch-feed streaming --feed-build $FEED --feed-ahead $AH 2>&1 | grep "host timing"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $REPO/gpurun_out/prof_fd -o p --output-format csv -- python $REPO/bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-kernel-events --steps 40 --warmup 8 --batch-feed streaming --feed-build $FEED --feed-ahead $AH > $REPO/gpurun_out/fd_prof.log 2>&1
cd $REPO; ls gpurun_out/prof_fd/*/ 2>/dev/null | head
python - <<'PY'
import csv, glob
kt = glob.glob("gpurun_out/prof_fd/**/*kernel_trace.csv", recursive=True)[0]
mt = glob.glob("gpurun_out/prof_fd/**/*memory_copy_trace.csv", recursive=True)
ev = []
for r in csv.DictReader(open(kt)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id", "?"), r["Kernel_Name"][:60])))
if mt:
    rows = list(csv.DictReader(open(mt[0])))
    print("memcopy columns:", list(rows[0].keys()) if rows else None)
    for r in rows:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M %s %s" % (r.get("Direction", r.get("Name", "?")), r.get("Size", r.get("Bytes", "?")))))
ev.sort()
t_end = ev[-1][1]
# steady state: a 4.4 ms window starting at the 30th-from-last backward launch
bw = [s for s, e, n in ev if "k_lstm_bwd<false" in n]
w0 = bw[-30] - 50_000; w1 = w0 + 4_400_000
t0 = None
for s, e, n in ev:
    if w0 <= s <= w1:
        if t0 is None: t0 = s
        print("%9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
rm -rf gpurun_out/prof_fd
