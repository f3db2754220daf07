"""Repeat-run stress of the round-6 paths whose failures would be races, not arithmetic: LDS-DMA id staging one tile ahead (forward and bottom BPTT launch),
the time-split tile hand-over, the small tables' gradients inside the bottom launch, the scoring pass inside the training forward's launch, both layers' BPTT in one launch (the bottom layer's workgroups waiting on the top layer's flags).  Every repeat is
compared with the f64 oracle (gradients) or with the first repeat (scores: bit for bit).  REPS=40 python scripts/gpu_stress_r6.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kprn_amd import _ffi, synth  # noqa: E402
from oracle.oracle import Oracle, make_cfg  # noqa: E402

reps = int(os.environ.get("REPS", "40"))
SHAPE = dict(Vt=6, Ve=30000, Vr=9, dt=16, de=32, dr=16, H=64, L=2)


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(1e-30, np.max(np.abs(b))))


bad_total = 0
for (pairs, P, T, small, plan, real_len) in [(19200, 1, 6, "0", False, 6), (5000, 4, 6, "0", True, None), (300 * 64 - 3, 1, 2, "0", False, 2), (120, 4, 6, "1", True, None),
                                             (2100, 4, 6, "1", True, None)]:
    eng = _ffi.Engine(SHAPE["Vt"], SHAPE["Ve"], SHAPE["Vr"], SHAPE["dt"], SHAPE["de"], SHAPE["dr"], SHAPE["H"], SHAPE["L"])
    eng.set_option("small_tiles", small)
    eng.set_option("prefix_plan", "1" if plan else "0")
    eng.set_option("score_overlap", "1")
    o64 = Oracle(make_cfg(**SHAPE), np.float64)
    theta = o64.init_params(3, 0.1).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(pairs, P, T, Ve=SHAPE["Ve"], seed=pairs % 97, real_len=real_len)
    b = eng.batch(idx, labels)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    first = None
    nbad = 0
    for rep in range(reps):
        eng.forward_async(b, 1)                       # (small batches: deferred into the training forward's launch below)
        out = eng.forward(b, 1, want=("path_scores",))["path_scores"].copy() if rep % 4 == 3 else None
        loss = eng.backward(b, 1)
        probs = eng.read_probs(b.B).copy()
        g = eng.get_flat_grads()
        worst = max(rel(g[off:off + int(np.prod(shp))], og[off:off + int(np.prod(shp))]) for _, (off, shp) in eng.layout().items())
        if first is None:
            first = probs
        ok = worst < 2e-4 and abs(loss - ol) < 1e-5 * max(1.0, abs(ol)) and np.array_equal(probs, first)
        nbad += not ok
    bad_total += nbad
    print(f"pairs={pairs} P={P} T={T} small={small} plan={plan}: bad repeats {nbad}/{reps}")
    eng.close()
print("STRESS", "FAILED" if bad_total else "ok")
