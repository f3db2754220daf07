"""Diagnosis (round 5): three Adam steps of the shipped rnn shape through the persistent layer launch ("persist_layers" = 2) and through the
per-step launches (= 0), each against the float64 oracle, per tensor: max |delta|, rms, and how many elements are off by more than 2e-4."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kprn_amd import _ffi, synth  # noqa: E402
from oracle.oracle import Oracle, make_cfg, make_opt  # noqa: E402

dt, de, dr, H, L = 50, 100, 50, 250, 1
o64 = Oracle(make_cfg(Vt=6, Ve=800, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L, rnn_type=1, use_relu=1), np.float64)
theta0 = o64.init_params(5, 0.07).astype(np.float32).astype(np.float64)
o64.zero_pad(theta0)
idx, labels = synth.make_paths(200, 3, 6, Ve=800, seed=77)
th, st = theta0.copy(), o64.new_state()
oopt = make_opt(method=1, lr=5e-3)
ol = [o64.train_step(th, st, oopt, idx, labels)[0] for _ in range(3)]
res = {}
for mode in ("2", "0"):
    eng = _ffi.Engine(6, 800, 9, dt, de, dr, H, L, rnn_type=1, use_relu=1, param_init=0.07)
    eng.set_option("impl", "generic")
    eng.set_option("persist_layers", mode)
    eng.set_flat_params(theta0.astype(np.float32))
    b = eng.batch(idx, labels)
    gl = [eng.train_step(b, _ffi.make_opt(method=1, lr=5e-3)) for _ in range(3)]
    res[mode] = eng.get_flat_params().astype(np.float64)
    print("mode", mode, "loss deltas", [f"{a - b_:.2e}" for a, b_ in zip(gl, ol)])
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        d = np.abs(res[mode][off:off + n] - th[off:off + n])
        mv = np.abs(th[off:off + n] - theta0[off:off + n])
        print(f"   {nm:14s} max {d.max():.2e} rms {np.sqrt(np.mean(d * d)):.2e}  > 2e-4: {int((d > 2e-4).sum())} of {n}   (oracle walk max {mv.max():.2e})")
    eng.close()
d = np.abs(res["2"] - res["0"])
print("persist vs steps: max", d.max(), "count > 2e-4", int((d > 2e-4).sum()))
