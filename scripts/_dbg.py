import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kprn_amd import _ffi
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
for name in ["c2_small.npz"]:
    z = np.load(os.path.join(GOLD, name))
    c = [int(x) for x in z["cfg"]]
    print("cfg", c)
    eng = _ffi.Engine(c[0], c[1], c[2], c[3], c[4], c[5], c[8], c[9], F=c[6], num_types=c[7], C_=c[10], reducer=c[11], K=c[12])
    eng.set_flat_params(z["theta"])
    b = eng.batch(z["idx"], z["labels"])
    print("idx shape", z["idx"].shape)
    loss = eng.backward(b, 1)
    g = eng.get_flat_grads(); ref = z["grad"]
    Vt, Ve, Vr, dt, de, dr, F, nT, H, L, C = c[:11]
    D = dt * 1 + de + dr
    off = 0
    blocks = [("Wt", Vt * dt), ("We", Ve * de), ("Wr", Vr * dr)]
    for l in range(L):
        din = D if l == 0 else H
        blocks += [(f"l{l}.Wi", 4 * H * din), (f"l{l}.bi", 4 * H), (f"l{l}.Wo", 4 * H * H)]
    blocks += [("outW", C * H), ("outb", C)]
    for nm, n in blocks:
        gg, rr = g[off:off + n], ref[off:off + n]
        print(f"{nm:8s} n={n:8d} nan={int(np.isnan(gg).sum()):6d} maxerr={np.nanmax(np.abs(gg - rr)):.3e} refmax={np.max(np.abs(rr)):.3e}")
        off += n
