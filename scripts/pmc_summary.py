#!/usr/bin/env python3
"""Per-kernel averages of the rocprofv3 PMC passes written by scripts/gpu_profile.sh.

Units / gfx950 corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are reported in KiB-like units of the TCC_EA request counters; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so wide coalesced reads are DOUBLED here
("fetch_bytes_corrected"); WRITE_SIZE is uncalibrated and reported as-is (x1024).
"""
import csv
import re
import json
import os
import sys
from collections import defaultdict


def load(path):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    if not os.path.exists(path):
        return acc
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "?")
            c = row.get("Counter_Name", "?")
            v = float(row.get("Counter_Value", "0") or 0)
            a = acc[k][c]
            a[0] += v
            a[1] += 1
    return acc


def short(name):
    """kernel family as bench.py's profiler names it"""
    if "k_lstm16_bwd_persist" in name:   # persistent bf16 BPTT kernel
        return "lstm_persist_bf16_bwd"
    if "k_lstm16_persist" in name:   # persistent bf16 layer kernel: <KX, KH, SAVE, ...>
        return "lstm_persist_bf16_train" if "24, 24, true" in name.replace("(bool)1", "true") else "lstm_persist_bf16_score"
    if "lp32::k_layer<" in name:      # one wide fp32 layer, all T steps (layer_f32_persist.hip): <CELL, NCH, SAVE>
        cell = "rnn" if "k_layer<1" in name else ("gru" if "k_layer<2" in name else "lstm")
        return cell + ("_layer_fwd_train" if "true>" in name.replace("(bool)1", "true") else "_layer_fwd")
    if "lp32::k_bptt<" in name:
        return ("rnn" if "k_bptt<1" in name else ("gru" if "k_bptt<2" in name else "lstm")) + "_layer_bwd"
    if "k_lstm_fwd_mc" in name:
        return "lstm_mc_fwd_train" if "true>" in name else "lstm_mc_fwd"
    if "k_lstm_fwd" in name:   # k_lstm_fwd<L, SAVE, NMT>: the training (SAVE) and the scoring launch are separate families
        return "lstm_fused_fwd_train" if re.search(r"k_lstm_fwd<\d+, (true|\(bool\)1)", name) else "lstm_fused_fwd"
    if "k_lstm_bwd" in name:
        return "lstm_fused_bwd"
    n = name.split("(")[0]
    return n.split("::")[-1]


def main(d):
    out = {}
    for fname in ("fetch", "write", "sq"):
        acc = load(os.path.join(d, fname + ".csv"))
        merged = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))  # template variants of one family are pooled
        for k, cs in acc.items():
            for c, (s_, n_) in cs.items():
                m = merged[short(k)][c]
                m[0] += s_
                m[1] += n_
        for k, cs in merged.items():
            o = out.setdefault(k, {})
            for c, (s_, n_) in cs.items():
                o[c + "_avg"] = s_ / max(n_, 1)
                o["dispatches"] = n_
    for k, o in out.items():
        if "FETCH_SIZE_avg" in o:
            o["fetch_bytes_raw"] = o["FETCH_SIZE_avg"] * 1024.0
            o["fetch_bytes_corrected"] = o["FETCH_SIZE_avg"] * 1024.0 * 2.0
        if "WRITE_SIZE_avg" in o:
            o["write_bytes_raw"] = o["WRITE_SIZE_avg"] * 1024.0
        if "SQ_VALU_MFMA_BUSY_CYCLES_avg" in o and o.get("SQ_BUSY_CU_CYCLES_avg"):
            o["mfma_busy_over_cu_busy"] = o["SQ_VALU_MFMA_BUSY_CYCLES_avg"] / o["SQ_BUSY_CU_CYCLES_avg"]
            o["mfma_busy_frac_per_simd"] = o["mfma_busy_over_cu_busy"] / 4.0  # MFMA-busy is summed over the CU's 4 SIMDs
    for k, o in out.items():
        if "fetch_bytes_corrected" in o or "write_bytes_raw" in o:
            o["hbm_bytes_per_launch"] = o.get("fetch_bytes_corrected", 0.0) + o.get("write_bytes_raw", 0.0)
    keep = {k: v for k, v in out.items() if "lstm" in k or "rnn_layer" in k or "entity_grad" in k or "adam" in k or "loss" in k or "gemm16" in k or "gemm_tiled" in k
            or "gemm_kernel" in k}
    print(json.dumps(keep, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1])
