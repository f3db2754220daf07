"""Repeat-run stress of the GRU launches (layer_f32_persist.hip CELL 2), whose failures would be races, not arithmetic: r * h' overwriting h_{t-1} in the LDS tile
between a step's two products (forward), the BPTT launch's counted vmcnt waits with save requests between the DMA groups, x_{t+1} requested mid-step.  bench.py's
batch (65 536 paths: four tiles per workgroup on every CU) and a ragged small one; every repeat against the FIRST repeat -- probabilities bit for bit (the forward has
no atomics), gradients to the split-K atomics' reordering -- and the first repeat against the per-step launches.  REPS=40 python scripts/gpu_stress_gru.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kprn_amd import _ffi, synth  # noqa: E402

reps = int(os.environ.get("REPS", "40"))
bad_total = 0
for dims, L, pairs, P, T in [((50, 100, 50, 250), 1, 16384, 4, 6), ((32, 64, 32, 128), 2, 16384, 4, 6), ((64, 64, 64, 192), 2, 5000, 1, 5), ((16, 32, 16, 80), 1, 4099, 3, 4)]:
    dt, de, dr, H = dims
    eng = _ffi.Engine(6, 200000, 9, dt, de, dr, H, L, rnn_type=2, param_init=0.06)
    eng.set_option("impl", "generic")
    idx, labels = synth.make_paths(pairs, P, T, Ve=200000, seed=pairs % 89)
    b = eng.batch(idx, labels)
    eng.set_option("persist_layers", "0")
    p0 = eng.forward(b, 1, want=("probs",))["probs"].astype(np.float64)
    eng.backward(b, 1)
    g0 = eng.get_flat_grads().astype(np.float64)
    eng.set_option("persist_layers", "2")
    first = None
    nbad = 0
    for rep in range(reps):
        p = eng.forward(b, 1, want=("probs",))["probs"].copy()
        eng.backward(b, 1)
        g = eng.get_flat_grads().astype(np.float64)
        if first is None:
            first = (p, g)
            ok = np.max(np.abs(p - p0)) < 2e-6 and np.max(np.abs(g - g0)) < 1e-4 * np.max(np.abs(g0))
        else:
            ok = np.array_equal(p, first[0]) and np.max(np.abs(g - first[1])) < 2e-5 * np.max(np.abs(first[1]))
        nbad += not ok
    bad_total += nbad
    print(f"D={dt + de + dr} H={H} L={L} paths={pairs * P} T={T}: bad repeats {nbad}/{reps}", flush=True)
    eng.close()
print("STRESS", "FAILED" if bad_total else "ok")
