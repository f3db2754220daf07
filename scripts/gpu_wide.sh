#!/bin/bash
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_wide.py -k "bf16 or 2_pow"  -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | grep -v "^    " | tail -25
run() { name=$1; shift; timeout 600 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --batch-feed resident --steps 8 --warmup 3 "$@" > gpurun_out/w_$name.log 2>&1
  grep '^{' gpurun_out/w_$name.log | tail -1 > gpurun_out/w_$name.json
  python -c "import sys,json; d=json.load(open('gpurun_out/w_$name.json')); print('$name', d['value'], d['ms_per_step'], d.get('roofline')); print({k:(round(v['ms']/max(1,v['launches']),4), v['launches']) for k,v in d['kernels'].items()})" || tail -5 gpurun_out/w_$name.log; }
run dimsB --dims B
run dimsB_L1 --dims B --layers 1
run shipped --dims shipped
run generic_A --impl generic
