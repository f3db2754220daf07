"""GPU diagnostics of the time-split tile hand-over (tests/test_gpu_handover.py): which rows / gradients differ between tile_handover on and off."""
import sys
import numpy as np
sys.path.insert(0, ".")
from kprn_amd import _ffi, synth

SHAPE = dict(Vt=6, Ve=30000, Vr=9, dt=16, de=32, dr=16, H=64, L=2)
T = 6


def mk(handover, plan, L=2):
    eng = _ffi.Engine(SHAPE["Vt"], SHAPE["Ve"], SHAPE["Vr"], SHAPE["dt"], SHAPE["de"], SHAPE["dr"], SHAPE["H"], L)
    eng.set_option("small_tiles", "0")
    eng.set_option("prefix_plan", "1" if plan else "0")
    eng.set_option("tile_handover", "1" if handover else "0")
    rng = np.random.default_rng(3)
    eng.set_flat_params((rng.random(eng.n_params) * 0.2 - 0.1).astype(np.float32))
    return eng


for pairs, P, plan, real_len in ((19200, 1, False, 6), (16384, 4, True, None)):
    idx, labels = synth.make_paths(pairs, P, T, Ve=SHAPE["Ve"], seed=11, real_len=real_len)
    res = {}
    for on in (True, False):
        eng = mk(on, plan)
        b = eng.batch(idx, labels)
        st = b.handover_stats
        out = eng.forward(b, 1, want=("path_scores",))
        ps = out["path_scores"].copy()
        out2 = eng.forward(b, 1, want=("path_scores",))
        loss = eng.backward(b, 1)
        g = eng.get_flat_grads().copy()
        res[on] = (ps, out2["path_scores"].copy(), loss, g, st, eng.layout())
        eng.close()
    ps1, ps0 = res[True][0], res[False][0]
    bad = np.where(np.any(ps1 != ps0, axis=1))[0]
    print(f"case pairs={pairs} P={P} plan={plan}: stats on={res[True][4]} off={res[False][4]}; rows differing {bad.size} of {ps1.shape[0]};"
          f" second pass equal to first: on {np.array_equal(res[True][0], res[True][1])} off {np.array_equal(res[False][0], res[False][1])}")
    if bad.size:
        tiles = np.unique(bad // 64)
        print("  tiles with differing rows:", tiles[:40].tolist(), "count", tiles.size, " max |diff|", float(np.max(np.abs(ps1 - ps0))),
              " rows per tile:", np.bincount(bad // 64)[tiles][:20].tolist())
        print("  nan/inf in on:", int(np.sum(~np.isfinite(ps1))), " first bad rows:", bad[:10].tolist())
    print("  loss on/off", res[True][2], res[False][2])
    g1, g0 = res[True][3].astype(np.float64), res[False][3].astype(np.float64)
    for nm, (off, shp) in res[True][5].items():
        n = int(np.prod(shp))
        d = np.max(np.abs(g1[off:off + n] - g0[off:off + n])) / max(1e-30, np.max(np.abs(g0[off:off + n])))
        print(f"    grad {nm:24s} rel diff {d:.3e}")
