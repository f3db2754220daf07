#!/bin/bash
# stall breakdown of the persistent bf16 layer kernel (scoring launch): SQ wait / issue / LDS counters, separate passes, kernel-trace only
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/pmc_persist_${1:-a}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() { name="$1"; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/raw_$name" -o p --output-format csv -- \
    python "$REPO/bench.py" --dims C4 --score-only --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident --no-kernel-events --entities 2000000 > "$OUT/$name.log" 2>&1
  find "$OUT/raw_$name" -name "*counter_collection.csv" -exec cp {} "$OUT/$name.csv" \;
  rm -rf "$OUT/raw_$name"; }
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass c GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM
python - "$OUT" <<'PY'
import csv, sys, collections, json, os
out = {}
for f in "abc":
    p = os.path.join(sys.argv[1], f + ".csv")
    if not os.path.exists(p): continue
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(p)):
        if "k_lstm16_persist" not in r.get("Kernel_Name", ""): continue
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"] or 0); a[1] += 1
    for k, (s, n) in acc.items(): out[k] = s / max(n, 1)
print(json.dumps(out, indent=1, sort_keys=True))
PY
rm -f "$OUT"/*.csv
