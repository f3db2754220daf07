#!/bin/bash
# round 3: cache policy of the persistent training launch's save stores (default / nt / sc0 nt / sc0 sc1), variant libraries from scripts/build_variants.py
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
TAG="${1:-a}"
run() { name=$1; lib=$2; shift; shift; KPRN_LIB=$lib timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --no-other-configs --no-batch-sweep --batch-feed resident "$@" > gpurun_out/nt_${TAG}_$name.log 2>&1
  grep -a '^{' gpurun_out/nt_${TAG}_$name.log | tail -1 > gpurun_out/nt_${TAG}_$name.json
  python - <<PY || tail -5 gpurun_out/nt_${TAG}_$name.log
import json; d=json.load(open('gpurun_out/nt_${TAG}_$name.json')); k=d['kernels']
print('$name', d['value'], d['ms_per_step'], {n: round(v['ms']/max(1,v['launches']),4) for n,v in k.items() if 'persist' in n or 'gates' in n})
PY
}
for rep in 1 2; do
run default_$rep "" --dims C4 --steps 8 --warmup 2
run nt_$rep $REPO/kprn_amd/libkprn_nt.so --dims C4 --steps 8 --warmup 2
run nt3_$rep $REPO/kprn_amd/libkprn_nt3.so --dims C4 --steps 8 --warmup 2
run sc_$rep $REPO/kprn_amd/libkprn_sc.so --dims C4 --steps 8 --warmup 2
done
