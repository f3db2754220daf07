"""Where a small step's time goes: per-family kernel time, wall time per step, host time to queue a step.
python scripts/gpu_small_batch.py [paths_per_step ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kprn_amd import _ffi, synth  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [256, 1024, 4096]
Vt, Ve, Vr, T = 6, 2851220, 9, 6
eng = _ffi.Engine(Vt, Ve, Vr, 16, 32, 16, 64, 2)
eng.set_option("score_overlap", os.environ.get("SCORE_OVERLAP", "1"))
opt = _ffi.make_opt(method=1, lr=1e-3)
out = {}
for pps in sizes:
    pool = []
    for i, P in enumerate([1, 2, 3, 4, 5, 8]):
        idx, labels = synth.make_paths(max(1, pps // P), P, T, Ve=Ve, seed=4242 + 13 * i)
        pool.append(eng.batch(idx, labels))
    def step(i):
        b = pool[i % len(pool)]
        eng.forward_async(b, 1)
        eng.train_step(b, opt, 1, want_loss=False)
    for i in range(12):
        step(i)
    eng.sync()
    K = 600
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    t_queue = time.perf_counter() - t0
    eng.sync()
    t_all = time.perf_counter() - t0
    eng.profile_reset(); eng.set_option("profile_filter", ""); eng.profile(True)
    for i in range(60):
        step(i)
    eng.sync(); eng.profile(False)
    fam = {k: round(v[0] / 60, 5) for k, v in sorted(eng.profile_get().items(), key=lambda kv: -kv[1][0])}
    out[pps] = {"wall_ms_per_step": round(1e3 * t_all / K, 4), "host_queue_ms_per_step": round(1e3 * t_queue / K, 4),
                "kernel_ms_per_step_by_family": fam, "kernel_ms_sum": round(sum(fam.values()), 4)}
    print(pps, json.dumps(out[pps]), flush=True)
    for b in pool:
        b.free()
