"""Measurement build: kprn_amd/libkprn_variants.so = libkprn.so with the persistent bf16 layer kernel's knock-out / tuning
variants compiled in (-DKPRN_PERSIST_VARIANTS: lstm_bf16_persist.hip, lstm_bf16_bwd_persist.hip).  Loaded with KPRN_LIB=<path> (kprn_amd/_ffi.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kprn_amd import build as kb  # noqa: E402

kb.build()
# (KPRN_VARIANT_FILES: other sources to recompile with KPRN_VARIANT_DEFS, e.g. batch_index.hip with -DKPRN_EG_OCC=0)
VARIANT_FILES = os.environ.get("KPRN_VARIANT_FILES", "lstm_bf16_persist.hip lstm_bf16_bwd_persist.hip").split()
objs = []
for src in kb.sources():
    obj = os.path.join(kb.HERE, "build", os.path.basename(src) + ".o")
    if os.path.basename(src) in VARIANT_FILES:
        obj = os.path.join(kb.HERE, "build", os.path.basename(src)[:-4] + ".variants.o")
        subprocess.check_call([kb.HIPCC] + kb.FLAGS + kb.file_flags(src) + ["-DKPRN_PERSIST_VARIANTS"] + os.environ.get("KPRN_VARIANT_DEFS", "").split() + ["-c", src, "-o", obj])
    objs.append(obj)
out = os.path.join(kb.HERE, os.environ.get("KPRN_VARIANT_NAME", "libkprn_variants.so"))
subprocess.check_call([kb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
