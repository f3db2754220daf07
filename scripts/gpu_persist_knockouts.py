"""Times the persistent bf16 layer kernel's measurement variants on the configs[3] shape (65 536 paths, T = 6, D = H = 384), one process:
KPRN_LIB=kprn_amd/libkprn_variants.so python scripts/gpu_persist_knockouts.py   (build: python scripts/build_variants.py)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kprn_amd import _ffi, synth  # noqa: E402

Ve = 2_000_000
eng = _ffi.Engine(6, Ve, 100, 128, 128, 128, 384, 1, compute_dtype=1, param_init=0.05)
idx, labels = synth.make_paths(16384, 4, 6, Ve=Ve, Vr=100, seed=3)
b = eng.batch(idx, labels)
N, T, D, H = 65536, 6, 384, 384
flops = N * 2 * 4 * H * (T * D + (T - 1) * H)
BWD = len(sys.argv) > 1 and sys.argv[1] == "bwd"   # the persistent BPTT launch (lstm_bf16_bwd_persist.hip) instead of the forward
variants = [("8 waves pf8 la1 (default)", {}), ("8 waves pf12 la1", {"KPRN_PERSIST_PF": "12"}), ("8 waves pf8 la2", {"KPRN_PERSIST_LA": "2"}), ("8 waves pf12 la2", {"KPRN_PERSIST_PF": "12", "KPRN_PERSIST_LA": "2"}),
            ("4 waves pf12 la2", {"KPRN_PERSIST_NW": "4", "KPRN_PERSIST_PF": "12", "KPRN_PERSIST_LA": "2"}), ("4 waves pf24 la2", {"KPRN_PERSIST_NW": "4", "KPRN_PERSIST_PF": "24", "KPRN_PERSIST_LA": "2"}),
            ("no cell", {"KPRN_PERSIST_DBG": "1"}), ("no weight stream", {"KPRN_PERSIST_DBG": "2"}), ("no LDS reads", {"KPRN_PERSIST_DBG": "4"}),
            ("cell math without its loads / stores", {"KPRN_PERSIST_DBG": "16"}),
            ("no weights, no LDS", {"KPRN_PERSIST_DBG": "6"}), ("MFMA only", {"KPRN_PERSIST_DBG": "7"}),
            ("no cell, no MFMA", {"KPRN_PERSIST_DBG": "9"}), ("weight stream only", {"KPRN_PERSIST_DBG": "13"}),
            ("skeleton (barriers, DMA, ids)", {"KPRN_PERSIST_DBG": "15"})]
if BWD:
    flops = N * 2 * 4 * H * (T - 1) * H
    variants = [("full", {}), ("no dA^T / bias pass", {"KPRN_PERSIST_BWD_DBG": "1"}), ("no row-major dA stores", {"KPRN_PERSIST_BWD_DBG": "2"}),
                ("no outputs at all", {"KPRN_PERSIST_BWD_DBG": "3"}), ("no product", {"KPRN_PERSIST_BWD_DBG": "4"}), ("no outputs, no product", {"KPRN_PERSIST_BWD_DBG": "7"}),
                ("no save loads", {"KPRN_PERSIST_BWD_DBG": "8"}), ("cell arithmetic + barriers only", {"KPRN_PERSIST_BWD_DBG": "15"})]
    eng.forward(b, 1)
    eng.backward(b, 1)
ROUNDS = 3
res = {name: [] for name, _ in variants}
for rnd in range(ROUNDS):          # interleaved rounds: a drift of the box (clocks, temperature) hits every variant alike
    for name, env in variants:
        for k in ("KPRN_PERSIST_DBG", "KPRN_PERSIST_PF", "KPRN_PERSIST_LA", "KPRN_PERSIST_NW", "KPRN_PERSIST_BWD_DBG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        if BWD:
            eng.backward(b, 1)
            eng.profile_reset(); eng.profile(True)
            for _ in range(4):
                eng.backward(b, 1)
        else:
            eng.forward(b, 1)
            eng.profile_reset(); eng.profile(True)
            for _ in range(4):
                eng.forward(b, 1)
        eng.sync(); eng.profile(False)
        ms, n = eng.profile_get()["lstm_persist_bf16_bwd" if BWD else "lstm_persist_bf16_score"]
        res[name].append(ms / n)
out = {}
for name, _ in variants:
    v = sorted(res[name])
    out[name] = {"ms_min": round(v[0], 4), "ms_median": round(v[len(v) // 2], 4), "frac_of_bf16_peak_at_median": round(flops / (v[len(v) // 2] * 1e-3) / 2.5e15, 4)}
    print("%-24s min %8.4f  median %8.4f ms  %.3f of 2.5 PF" % (name, v[0], v[len(v) // 2], out[name]["frac_of_bf16_peak_at_median"]), flush=True)
print(json.dumps(out))
