#!/usr/bin/env python3
"""Knock-out builds of the split-K bf16 product gx::k_gemm16x on configs[3]'s merged dW shape (1 536 x 640, K = 393 216) through kprn_debug_gemm (what 6):
what does the launch cost without its MFMAs / its fragment reads / its DMA / its epilogue?  Needs the measurement build:
  KPRN_VARIANT_FILES=lstm_bf16.hip python scripts/build_variants.py && KPRN_LIB=kprn_amd/libkprn_variants.so python scripts/gpu_gemm16_knockouts.py
(each mask runs in its own process: the switch is read once)."""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from kprn_amd import _ffi
    eng = _ffi.Engine(6, 1000, 9, 16, 32, 16, 64, 1)
    L = eng.L
    L.kprn_debug_gemm.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_float)]
    out = {}
    for name, M, N, K in (("merged_dW_1536x640", 1536, 640, 65536 * 6), ("dW_1536x384", 1536, 384, 65536 * 6)):
        ms = C.c_float()
        eng._ck(L.kprn_debug_gemm(eng.h, 6, M, N, K, 8, C.byref(ms)))
        out[name] = round(ms.value, 4)
    print(json.dumps(out))
    sys.exit(0)
MASKS = [(0, "full"), (1, "no MFMAs"), (3, "no MFMAs, no fragment reads (DMA + waits + barriers + epilogue)"), (4, "no DMA"), (5, "no DMA, no MFMAs (fragment reads only)"),
         (7, "skeleton: barriers + epilogue"), (8, "no epilogue"), (9, "no MFMAs, no epilogue"), (12, "no DMA, no epilogue (MFMAs + fragment reads)"),
         (16, "full, sources as a K-blocked layout [K/64][rows][64] would have them"), (19, "DMA + waits + barriers + epilogue, K-blocked sources"),
         (32, "full, every chunk from the first 1 KB of its rows (all L2 hits)"), (35, "DMA + waits + barriers + epilogue, all L2 hits")]
ONLY = [int(x) for x in os.environ.get("KPRN_KNOCKOUT_MASKS", "").split(",") if x]
for rep in range(2):
    for m, what in MASKS:
        if ONLY and m not in ONLY:
            continue
        env = dict(os.environ, KPRN_GEMM16_DBG=str(m))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=env, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(f"DBG {m:2d} {what:70s}", line[-1] if line else ("FAILED " + r.stderr[-300:]), flush=True)
