#!/usr/bin/env python3
"""The bf16 pipeline's products on configs[3]'s shapes through kprn_debug_gemm (what 5 / 6): ms per launch, TFLOP/s, fraction of 2.5 PF.
KPRN_BF16_GEMM=old python scripts/gpu_gemm16_bench.py  -> the round-2 kernel (k_gemm16) on the same shapes."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kprn_amd import _ffi
eng = _ffi.Engine(6, 1000, 9, 16, 32, 16, 64, 1)
L = eng.L
L.kprn_debug_gemm.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_float)]
NP = 65536
cases = [("dh  [65k,1536] x [384,1536]^T", 5, NP, 384, 1536), ("dx  [393k,1536] x [384,1536]^T", 5, NP * 6, 384, 1536),
         ("dW i2g [1536,393k] x [384,393k]^T split-K", 6, 1536, 384, NP * 6), ("dW o2g [1536,328k] x [384,328k]^T split-K", 6, 1536, 384, NP * 5),
         ("square 8192^3", 5, 8192, 8192, 8192), ("4096^3", 5, 4096, 4096, 4096)]
for name, what, M, N, K in cases:
    ms = C.c_float()
    eng._ck(L.kprn_debug_gemm(eng.h, what, M, N, K, 10, C.byref(ms)))
    fl = 2.0 * M * N * K
    print("%-46s %8.4f ms  %7.1f TFLOP/s  %.3f of 2.5 PF" % (name, ms.value, fl / ms.value / 1e9, fl / ms.value / 1e9 / 2500.0), flush=True)
