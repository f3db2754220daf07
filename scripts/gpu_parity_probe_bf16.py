"""Measured parity margins of the bf16 pipeline (configs[3], compute_dtype = 1) against the float64 oracle: per tensor max / rms error,
cosine and sign agreement of the gradients, and a 20-step Adam loss curve.  tests/test_gpu_persist.py's bars are set from this output
(one order above the measured margins).   python scripts/gpu_parity_probe_bf16.py [pairs P T ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kprn_amd import _ffi, synth  # noqa: E402
from oracle.oracle import Oracle, make_cfg, make_opt  # noqa: E402

DIMS = (128, 128, 128)


def case(pairs, P, T, Ve=700, Vr=100, seed=4, init=0.05):
    dt, de, dr = DIMS
    eng = _ffi.Engine(6, Ve, Vr, dt, de, dr, 384, 1, compute_dtype=1)
    o64 = Oracle(make_cfg(Vt=6, Ve=Ve, Vr=Vr, dt=dt, de=de, dr=dr, H=384, L=1), np.float64)
    theta = o64.init_params(seed, init).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(pairs, P, T, Ve=Ve, Vr=Vr, seed=seed + 1)
    return eng, o64, theta, idx, labels


def tensor_stats(got, want):
    got, want = np.asarray(got, np.float64).ravel(), np.asarray(want, np.float64).ravel()
    scale = max(1e-30, np.max(np.abs(want)))
    d = got - want
    big = np.abs(want) > 0.05 * scale        # elements above the bf16 noise floor of the tensor
    return {"max_over_max": float(np.max(np.abs(d)) / scale), "rms_over_max": float(np.sqrt(np.mean(d * d)) / scale),
            "cos": float(got @ want / max(1e-300, np.linalg.norm(got) * np.linalg.norm(want))),
            "sign_agree_big": float(np.mean(np.sign(got[big]) == np.sign(want[big]))) if big.any() else 1.0, "n_big": int(big.sum())}


cases = [(150, 2, 6), (333, 3, 6), (143, 3, 4), (2000, 5, 6)]
if len(sys.argv) > 3:
    cases = [tuple(int(x) for x in sys.argv[i:i + 3]) for i in range(1, len(sys.argv) - 2, 3)]
for pairs, P, T in cases:
    eng, o64, theta, idx, labels = case(pairs, P, T)
    b = eng.batch(idx, labels)
    out = eng.forward(b, 1, want=("probs", "path_scores"))
    ps, _, probs = o64.forward(theta, idx)
    r = {"case": [pairs, P, T], "scores": tensor_stats(out["path_scores"], ps), "probs_abs": float(np.max(np.abs(out["probs"] - probs[:, 0])))}
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    r["loss_rel"] = float(abs(loss - ol) / max(1.0, abs(ol)))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        r["g_" + nm] = tensor_stats(g[off:off + n], og[off:off + n])
    print(json.dumps(r), flush=True)
    eng.close()

# 20 Adam steps (lr 1e-3, the reference default) on two alternating batches: loss curve and parameter walk against the oracle
eng, o64, theta, idx, labels = case(512, 3, 6, Ve=5000, seed=12)
idx2, lab2 = synth.make_paths(512, 3, 6, Ve=5000, Vr=100, seed=77)
gb = [eng.batch(idx, labels), eng.batch(idx2, lab2)]
ob = [(idx, labels), (idx2, lab2)]
th0 = theta.copy(); th = theta.copy(); st = o64.new_state()
oopt, gopt = make_opt(method=1, lr=1e-3), _ffi.make_opt(method=1, lr=1e-3)
dl = []
for s in range(20):
    ol, _ = o64.train_step(th, st, oopt, *ob[s & 1])
    gl = eng.train_step(gb[s & 1], gopt)
    dl.append(abs(gl - ol) / max(1.0, abs(ol)))
got = eng.get_flat_params().astype(np.float64)
r = {"adam20_loss_rel_max": float(max(dl)), "adam20_loss_rel": [round(float(x), 5) for x in dl]}
for nm, (off, shp) in eng.layout().items():
    n = int(np.prod(shp))
    de_, do_ = got[off:off + n] - th0[off:off + n], th[off:off + n] - th0[off:off + n]
    moved = np.abs(do_) > 0.25 * np.max(np.abs(do_))
    r["walk_" + nm] = {"cos": float(de_ @ do_ / max(1e-300, np.linalg.norm(de_) * np.linalg.norm(do_))),
                       "sign_agree_moved": float(np.mean(np.sign(de_[moved]) == np.sign(do_[moved]))), "n_moved": int(moved.sum()),
                       "max_abs_diff": float(np.max(np.abs(de_ - do_))), "rms_diff": float(np.sqrt(np.mean((de_ - do_) ** 2)))}
print(json.dumps(r), flush=True)
