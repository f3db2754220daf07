"""Measured parity margins at the benchmarked size: 20 Adam steps at lr 1e-3, per-element score error."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_fullsize import Case, PAIRS, P, T, VE
from kprn_amd import _ffi, synth
from oracle.oracle import make_opt
full = Case(PAIRS, 11)
res = {}
for cd, plan in ((0, True), (0, False), (2, True)):
    eng = full.engine(cd, "auto", plan)
    b0 = eng.batch(full.idx, full.labels)
    out = eng.forward(b0, 1, want=("path_scores", "probs"))
    err = np.abs(out["path_scores"].astype(np.float64) - full.ps)
    rel_el = err / np.maximum(np.abs(full.ps), 1e-6 * np.abs(full.ps).max())
    r = {"score_max_over_max": float(err.max() / np.abs(full.ps).max()), "score_per_element_rel_floor1e-6": float(rel_el.max()),
         "probs_rel": float(np.max(np.abs(out["probs"] - full.probs[:, 0]) / full.probs[:, 0]))}
    idx2, lab2 = synth.make_paths(PAIRS, P, T, Ve=VE, seed=501)
    batches = [(full.idx, full.labels), (idx2, lab2)]
    gb = [b0, eng.batch(idx2, lab2)]
    th = full.theta.copy(); st = full.o64.new_state()
    oopt = make_opt(method=1, lr=1e-3); gopt = _ffi.make_opt(method=1, lr=1e-3)
    dl = []
    t0 = time.time()
    for s in range(20):
        i, l = batches[s & 1]
        ol, _ = full.o64.train_step(th, st, oopt, i, l)
        gl = eng.train_step(gb[s & 1], gopt)
        dl.append(abs(gl - ol) / max(1.0, abs(ol)))
    got = eng.get_flat_params()
    r["adam20_lr1e-3_max_dtheta"] = float(np.max(np.abs(got - th)))
    r["adam20_loss_rel_max"] = float(max(dl))
    o2 = eng.forward(gb[0], 1, want=("probs",))
    _, _, probs = full.o64.forward(th, full.idx)
    r["probs_after_training_rel"] = float(np.max(np.abs(o2["probs"] - probs[:, 0]) / probs[:, 0]))
    r["seconds"] = round(time.time() - t0, 1)
    res[f"cd{cd}_plan{int(plan)}"] = r
    print(cd, plan, json.dumps(r), flush=True)
    eng.close()
