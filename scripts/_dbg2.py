import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_parity import mk, rel_inf
from kprn_amd import synth
for (L, red, P, npairs, seed) in [(1, 0, 5, 23, 6), (1, 2, 5, 23, 6), (1, 2, 2, 41, 7), (2, 2, 4, 41, 7), (1, 2, 1, 64, 3), (1, 2, 1, 16, 3)]:
    eng, o64, theta = mk(L=L, reducer=red, K=2, impl="auto")
    idx, labels = synth.make_paths(npairs, P, 6, Ve=300, seed=seed)
    b = eng.batch(idx, labels)
    loss = eng.backward(b, 3)
    ol, og, _ = o64.forward_backward(theta, idx, labels, class_id=3)
    g = eng.get_flat_grads()
    print(f"L={L} reducer={red} P={P} pairs={npairs}: loss {loss:.6f} vs {ol:.6f}")
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        print(f"   {nm:22s} rel_inf={rel_inf(g[off:off + n], og[off:off + n]):.3e} max={np.max(np.abs(og[off:off+n])):.3e}")
