"""GRU through the persistent layer launches (layer_f32_persist.hip CELL 2): parity against the per-step launches and the f64 oracle on small shapes,
then the launch times at bench.py's batch (65 536 paths, D = 200, H = 250, T = 6).  python scripts/gpu_probe_gru_persist.py [quick]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kprn_amd import _ffi, synth  # noqa: E402
from oracle.oracle import Oracle, make_cfg  # noqa: E402


def rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(1e-30, np.max(np.abs(b))))


out = {}
for dims, L, pairs, P, T in [] if (len(sys.argv) > 1 and sys.argv[1] == "time") else [((50, 100, 50, 250), 1, 150, 2, 6), ((64, 64, 64, 192), 2, 129, 1, 4), ((16, 32, 16, 80), 1, 100, 2, 3), ((16, 32, 16, 64), 2, 300, 3, 6), ((64, 64, 64, 192), 1, 40, 1, 1)]:
    dt, de, dr, H = dims
    eng = _ffi.Engine(6, 800, 9, dt, de, dr, H, L, rnn_type=2, param_init=0.07)
    eng.set_option("impl", "generic")
    eng.set_option("persist_layers", "2")
    o64 = Oracle(make_cfg(Vt=6, Ve=800, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L, rnn_type=2), np.float64)
    theta = o64.init_params(5, 0.07).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(pairs, P, T, Ve=800, seed=pairs + T)
    b = eng.batch(idx, labels)
    eng.profile(True)
    o = eng.forward(b, 1, want=("probs", "path_scores"))
    loss = eng.backward(b, 1)
    fam = sorted(eng.profile_get())
    g = eng.get_flat_grads().astype(np.float64)
    ps, _, probs = o64.forward(theta, idx)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    eng.set_option("persist_layers", "0")
    o0 = eng.forward(b, 1, want=("probs", "path_scores"))
    eng.backward(b, 1)
    g0 = eng.get_flat_grads().astype(np.float64)
    worst = max(rel_inf(g[off:off + int(np.prod(shp))], og[off:off + int(np.prod(shp))]) for nm, (off, shp) in eng.layout().items())
    worst0 = max(rel_inf(g[off:off + int(np.prod(shp))], g0[off:off + int(np.prod(shp))]) for nm, (off, shp) in eng.layout().items())
    key = "D%d_H%d_L%d_N%d_T%d" % (dt + de + dr, H, L, pairs * P, T)
    out[key] = {"scores_vs_oracle": rel_inf(o["path_scores"], ps), "scores_vs_steps": rel_inf(o["path_scores"], o0["path_scores"].astype(np.float64)),
                "loss_err": abs(loss - ol), "grads_vs_oracle": worst, "grads_vs_steps": worst0, "families": [f for f in fam if "gru" in f or "o2g" in f]}
    print(key, json.dumps(out[key]), flush=True)
    eng.close()

if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)
eng = _ffi.Engine(6, 200000, 9, 50, 100, 50, 250, 1, rnn_type=2, param_init=0.06)
eng.set_option("impl", "generic")
idx, labels = synth.make_paths(16384, 4, 6, Ve=200000, seed=19)
b = eng.batch(idx, labels)
times = {}
for mode in ("1", "0", "1", "0"):
    eng.set_option("persist_layers", mode)
    eng.forward(b, 1); eng.backward(b, 1)
    eng.profile_reset(); eng.profile(True)
    for _ in range(4):
        eng.forward(b, 1)
        eng.backward(b, 1)
    eng.sync(); eng.profile(False)
    pg = eng.profile_get()
    times.setdefault(mode, []).append({k: round(v[0] / 4, 4) for k, v in sorted(pg.items(), key=lambda kv: -kv[1][0])[:14]})
    print("persist_layers", mode, json.dumps(times[mode][-1]), flush=True)
out["times_ms_per_step"] = times
print(json.dumps(out))
