#!/usr/bin/env python3
"""Soak of the streaming feed: random minibatch shapes through a ring of slots (host- and device-built, labels / no labels, rows feed),
training and scoring interleaved, every slot's results compared with a batch made by kprn_batch_create from the same ids on a second
engine that takes the same steps (to 1e-5: the two engines' parameters drift apart by fp32 atomics' ordering, ~1e-7).
usage: gpu_feed_soak.py [iterations] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kprn_amd import _ffi, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
Ve = 20000
files = [synth.make_paths(4000, P, 6, Ve=Ve, seed=100 + P) for P in (1, 2, 3, 5, 8)]
bad = 0
t0 = time.time()
for build in ("host", "device"):
    eng = _ffi.Engine(6, Ve, 9, 16, 32, 16, 64, 2, seed=3)
    ref = _ffi.Engine(6, Ve, 9, 16, 32, 16, 64, 2, seed=3)
    eng.set_option("feed_build", build)
    eng.set_option("score_overlap", "1")
    opt = _ffi.make_opt(method=1, lr=1e-3)
    NS = 4
    slots = [None] * NS
    pend = []     # (slot index, idx, labels or None)
    for it in range(iters):
        data, labels = files[rng.integers(len(files))]
        n = int(rng.integers(1, 1500))
        rows = rng.permutation(data.shape[0])[:n].astype(np.int64)
        k = it % NS
        mode = rng.integers(3)
        if mode == 0:
            slots[k] = eng.feed_rows(data, labels, rows, slot=slots[k]); item = (k, data[rows], labels[rows])
        elif mode == 1:
            idx = np.ascontiguousarray(data[rows]); lab = np.ascontiguousarray(labels[rows])
            slots[k] = eng.feed(idx, lab, slot=slots[k]); item = (k, idx, lab)
        else:
            idx = np.ascontiguousarray(data[rows])
            slots[k] = eng.feed(idx, None, slot=slots[k]); item = (k, idx, None)
        pend.append(item)
        if len(pend) >= NS - 1:           # use the oldest fed slot
            kk, idx, lab = pend.pop(0)
            b = slots[kk]
            rb = ref.batch(idx, lab)
            if lab is not None and rng.integers(2):
                eng.forward_async(b, 1)
                got = None
                la = eng.train_step(b, opt)
                got = eng.read_probs(b.B)
                want = ref.forward(rb, 1)["probs"]
                lb = ref.train_step(rb, opt)
                ok = np.allclose(got, want, rtol=1e-5, atol=1e-6) and abs(la - lb) <= 1e-5 * max(1.0, abs(lb))
            else:
                got = eng.forward(b, 1, want=("probs", "path_scores"))
                want = ref.forward(rb, 1, want=("probs", "path_scores"))
                ok = np.allclose(got["probs"], want["probs"], rtol=1e-5, atol=1e-6) and np.allclose(got["path_scores"], want["path_scores"], rtol=1e-4, atol=1e-5)
            if not ok:
                bad += 1
                print("MISMATCH build", build, "iter", it, "B", b.B, "P", b.P, "labels", lab is not None, flush=True)
            rb.free()
    d = float(np.abs(eng.get_flat_params() - ref.get_flat_params()).max())
    print("build", build, "iterations", iters, "max |param diff| vs the resident-batch engine", d, flush=True)
    if d > 1e-5: bad += 1
    eng.close(); ref.close()
print("soak done in %.1f s, failures: %d" % (time.time() - t0, bad))
sys.exit(1 if bad else 0)
