#!/usr/bin/env python3
"""GEMM shapes of the wide configurations through kprn_debug_gemm: ms per launch and TFLOP/s (fp32 MFMA peak 157.3)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kprn_amd import _ffi
eng = _ffi.Engine(6, 1000, 9, 16, 32, 16, 64, 1)
L = eng.L
L.kprn_debug_gemm.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_float)]
NP = 65536
cases = [
    ("B fwd step (H=192, Din=192)", 3, NP, 192, 192, lambda M, N, K: 2 * M * 4 * N * (K + N)),
    ("B i2g fwd  [393k,192]x[192,768]", 0, NP * 6, 768, 192, None),
    ("B dx       [393k,768]x[768,192]", 1, NP * 6, 192, 768, None),
    ("B dh       [65k,768]x[768,192]", 1, NP, 192, 768, None),
    ("B dW       [393k,768]^T x [393k,192]", 2, NP * 6, 768, 192, None),
    ("C4 fwd step (H=384, Din=384)", 3, NP, 384, 384, lambda M, N, K: 2 * M * 4 * N * (K + N)),
    ("C4 dW      [393k,1536]^T x [393k,384]", 2, NP * 6, 1536, 384, None),
    ("shipped rnn step (H=250, Din=200)", 4, NP, 250, 200, lambda M, N, K: 2 * M * N * (K + N)),
    ("shipped dx [393k,250]x[250,200]", 1, NP * 6, 200, 250, None),
    ("probe rnn step (H=256, Din=200)", 4, NP, 256, 200, lambda M, N, K: 2 * M * N * (K + N)),
    ("probe rnn step (H=256, Din=192)", 4, NP, 256, 192, lambda M, N, K: 2 * M * N * (K + N)),
    ("probe rnn step (H=252, Din=200)", 4, NP, 252, 200, lambda M, N, K: 2 * M * N * (K + N)),
    ("probe rnn step (H=512, Din=512)", 4, NP, 512, 512, lambda M, N, K: 2 * M * N * (K + N)),
    ("A generic fwd step (H=64)", 3, NP, 64, 64, lambda M, N, K: 2 * M * 4 * N * (K + N)),
]
sel = sys.argv[1:]
for name, what, M, N, K, fl in cases:
    if sel and not any(x in name for x in sel):
        continue
    ms = C.c_float()
    eng._ck(L.kprn_debug_gemm(eng.h, what, M, N, K, 10, C.byref(ms)))
    flops = fl(M, N, K) if fl else 2.0 * M * N * K
    print("%-42s %8.4f ms  %7.1f TFLOP/s  %.3f of peak" % (name, ms.value, flops / ms.value / 1e9, flops / ms.value / 1e9 / 157.3))
