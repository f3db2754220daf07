#!/usr/bin/env python3
"""One-GPU estimate of the data-parallel exchange cost at world sizes 2/4/8 (the dev box has one GPU; the driver runs the
real N-GPU bench).  Every simulated rank is a batch of its own: backward + pack fills that rank's slot of the gathered
buffer (a device copy stands in for the all-gather), then merge + optimiser step are timed on the full buffer."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from kprn_amd import _ffi, synth, dp

dev = "cuda:0"
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
Ve = 2851220
eng = _ffi.Engine(6, Ve, 9, 16, 32, 16, 64, 2, stream=stream, seed=12345, param_init=0.1)
opt = _ffi.make_opt(method=1, lr=1e-3)
paths = int(os.environ.get("PATHS", "65536"))
WMAX = 8
batches = []
for r in range(WMAX):
    idx, labels = synth.make_paths(paths // 2, 2, 6, Ve=Ve, seed=1000 + 7919 * r)
    batches.append(eng.batch(idx, labels))
cap = max(b.n_uniq for b in batches)
print("rows per rank:", [b.n_uniq for b in batches], "capacity", cap, "buffer MB per rank", (4 + cap * 33) * 4 / 1e6)
for W in (1, 2, 4, 8):
    words = 4 + cap * 33
    allb = torch.empty(words * W, dtype=torch.int32, device=dev)
    ts = {"merge": [], "update": []}
    for it in range(4):
        for r in range(W):
            eng.backward(batches[r], 1, False, 1.0 / (W * batches[r].B), want_loss=False)
            ptr, n = eng.sparse_grad_pack(cap)
            allb[r * words:(r + 1) * words].copy_(dp.wrap_device(ptr, n, "i32", torch.device(dev)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sparse_grad_merge(allb.data_ptr(), W, cap)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.apply_update(opt)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it:
            ts["merge"].append((t1 - t0) * 1e3); ts["update"].append((t2 - t1) * 1e3)
    print(f"world {W}: merge {np.mean(ts['merge']):.3f} ms, optimiser step on the union {np.mean(ts['update']):.3f} ms, "
          f"all-gather volume received per rank {(W - 1) * words * 4 / 1e6:.1f} MB")
