#!/usr/bin/env python3
"""One-GPU measurement of the data-parallel exchange at the payloads of world sizes 2 / 4 / 8 (the dev box has one GPU; the driver runs
the real N-GPU bench).  Every simulated rank is a batch of its own: backward + pack fills that rank's slot of the gathered buffer (a
device copy stands in for the all-gather's arrival), then merge + optimiser step run on the full buffer.  pack / merge / update are timed
by the engine's HIP events (kernel families dp_pack_rows, dp_merge_rows, adam_*), the gather stand-in by torch events; the real
all-gather's time is bounded from the payload: every rank receives (W - 1) packed buffers over its xGMI links.
  FUSED=1 python scripts/gpu_dp_sim.py > profiles/r03/dp_exchange_sim_fused.json   (FUSED=0: the separate merge)"""
import json, os, sys
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
cap = (max(b.n_uniq for b in batches) + 3) // 4 * 4
FUSED = os.environ.get("FUSED", "1")   # union inside the row update (dp_fused_update) or the separate marking merge
eng.set_option("dp_fused_update", FUSED)
eng.stream()   # (the exchange hooks want a caller that knows the engine's stream)
words = 4 + cap * 33
out = {"dp_fused_update": int(FUSED), "paths_per_rank_per_step": paths, "rows_per_rank": [b.n_uniq for b in batches], "capacity_rows": cap,
       "packed_MB_per_rank": round(words * 4 / 1e6, 2), "worlds": {}}
XGMI_GBS = 153.0   # per link, MI355X_MICROARCH.md; a rank receives from each peer over its own link
for W in (1, 2, 4, 8):
    allb = torch.empty(words * W, dtype=torch.int32, device=dev)
    eng.profile_reset()
    eng.set_option("profile_filter", "")
    copy_ms = []
    iters = 6
    for it in range(iters + 1):
        if it == 1:
            eng.sync(); eng.profile_reset(); eng.profile(True)
        for r in range(W):
            eng.backward(batches[r], 1, False, 1.0 / (W * batches[r].B), want_loss=False)
            ptr, n = eng.sparse_grad_pack(cap)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            allb[r * words:(r + 1) * words].copy_(dp.wrap_device(ptr, n, "i32", torch.device(dev)))
            e1.record()
            if it >= 1:
                copy_ms.append((e0, e1))
        eng.sparse_grad_merge(allb.data_ptr(), W, cap)
        eng.apply_update(opt)
    eng.sync()
    eng.profile(False)
    fam = eng.profile_get()
    per = lambda k: round(fam[k][0] / max(1, fam[k][1]), 4) if k in fam else None
    torch.cuda.synchronize()
    out["worlds"][str(W)] = {
        "pack_rows_ms": per("dp_pack_rows"), "merge_rows_ms": per("dp_merge_rows"),
        "optimizer_step_on_union_ms": round((fam.get("adam_entity_rows", (0, 1))[0] + fam.get("adam_dense", (0, 1))[0]) / iters, 4),
        "device_copy_per_rank_buffer_ms": round(float(np.mean([a.elapsed_time(b) for a, b in copy_ms])), 4),
        "allgather_received_MB_per_rank": round((W - 1) * words * 4 / 1e6, 1),
        "allgather_lower_bound_ms_one_link_per_peer": round(words * 4 / 1e6 / XGMI_GBS, 4) if W > 1 else 0.0,
    }
print(json.dumps(out, indent=1))
