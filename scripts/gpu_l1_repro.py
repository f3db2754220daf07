"""GPU diagnostics: L = 1 fused path, are two scoring passes over the same data bit-identical?  Which rows differ (by their pad-prefix length), and which pass
is right (against the no-plan scores of the same engine)?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from kprn_amd import _ffi, synth

T = 6
for L, npaths in ((1, 300 * 64 - 7), (1, 100 * 64), (1, 260 * 64), (2, 300 * 64 - 7)):
    idx, labels = synth.make_paths(npaths, 1, T, Ve=30000, seed=200, real_len=None)
    ent = idx[:, 0, :, 1]
    pad = (ent == 30000).sum(axis=1)
    eng = _ffi.Engine(6, 30000, 9, 16, 32, 16, 64, L)
    eng.set_option("small_tiles", "0"); eng.set_option("tile_handover", "0")
    rng = np.random.default_rng(3)
    eng.set_flat_params((rng.random(eng.n_params) * 0.2 - 0.1).astype(np.float32))
    eng.set_option("prefix_plan", "0")
    ref = eng.forward(eng.batch(idx, labels), 1, want=("path_scores",))["path_scores"].copy()
    ref2 = eng.forward(eng.batch(idx, labels), 1, want=("path_scores",))["path_scores"].copy()
    eng.set_option("prefix_plan", "1")
    outs = []
    for rep in range(3):
        b = eng.batch(idx, labels)
        outs.append(eng.forward(b, 1, want=("path_scores",))["path_scores"].copy())
    d01 = np.where(np.any(outs[0] != outs[1], axis=1))[0]
    print(f"L={L} paths={npaths}: no-plan passes identical {np.array_equal(ref, ref2)}; plan: rows differing pass0/1 {d01.size}, pad lengths of those rows {np.bincount(pad[d01], minlength=4).tolist()} "
          f"(all rows {np.bincount(pad, minlength=4).tolist()}); max |pass - noplan|: {[float(np.max(np.abs(o - ref))) for o in outs]}; rows off by > 1e-5 vs no-plan: {[int(np.sum(np.any(np.abs(o - ref) > 1e-5, axis=1))) for o in outs]}")
    eng.close()
