#!/bin/bash
# host-side feed derivation: thread scaling on the GPU box's cores, then the streaming bench
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$REPO"; mkdir -p gpurun_out
g++ -O2 scripts/ubench/host_index_bench.cpp -o /tmp/hb -L$REPO/kprn_amd -lkprn -Wl,-rpath,$REPO/kprn_amd -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib -lamdhip64 -pthread
lscpu | grep -E "^CPU\(s\)|Model name|NUMA node\(s\)|Thread"
timeout 600 python -m pytest tests/test_gpu_feed.py tests/test_gpu_host.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | grep -v "^    " | tail -8
KPRN_BENCH_HOST_TIMING=1 timeout 300 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --steps 40 --warmup 8 --batch-feed streaming 2>&1 | grep "host timing"
run() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-alt --no-extra-regions --steps 60 --warmup 8 "$@" > gpurun_out/fb_$name.log 2>&1
  grep '^{' gpurun_out/fb_$name.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'])" || tail -5 gpurun_out/fb_$name.log; }
run resident --batch-feed resident
run streaming_host --batch-feed streaming
run streaming_device --batch-feed streaming --feed-build device --feed-ahead 2
run streaming_host_a2 --batch-feed streaming --feed-ahead 2
run streaming_host_a8 --batch-feed streaming --feed-ahead 8
run streaming_trainonly --batch-feed streaming --train-only
