import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_parity import mk, rel_inf
from kprn_amd import synth
reps = int(os.environ.get("REPS", "30"))
for (L, P, npairs, seed) in [(2, 1, 32, 21), (2, 3, 32, 23), (2, 2, 32, 22), (1, 2, 32, 22), (2, 4, 700, 5)]:
    eng, o64, theta = mk(L=L, reducer=2, K=2, impl="auto")
    idx, labels = synth.make_paths(npairs, P, 6, Ve=300, seed=seed)
    b = eng.batch(idx, labels)
    ol, og, _ = o64.forward_backward(theta, idx, labels, class_id=1)
    worst = {}
    nbad = 0
    for rep in range(reps):
        loss = eng.backward(b, 1)
        g = eng.get_flat_grads()
        bad = False
        for nm, (off, shp) in eng.layout().items():
            n = int(np.prod(shp))
            r = rel_inf(g[off:off + n], og[off:off + n])
            worst[nm] = max(worst.get(nm, 0), r)
            if r > 2e-4: bad = True
        nbad += bad
    print(f"L={L} P={P} pairs={npairs}: bad runs {nbad}/{reps}; worst:", {k: f"{v:.1e}" for k, v in worst.items() if v > 1e-5})
