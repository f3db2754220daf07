"""Where a host-buffer minibatch step's time goes: the same 128-pair minibatches through (a) resident batches, loss read every
step, (b) kprn_batch_feed_async + train_step_batch, (c) kprn_train_step from host buffers; per-family kernel time of (c).
python scripts/gpu_dropin_probe.py"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kprn_amd import _ffi, synth  # noqa: E402

Vt, Ve, Vr, T = 6, 2851220, 9, 6
eng = _ffi.Engine(Vt, Ve, Vr, 16, 32, 16, 64, 2)
opt = _ffi.make_opt(method=1, lr=1e-3)
rng = np.random.default_rng(99)
mb = []
for i in range(32):
    P = int(min(rng.geometric(0.57), 28))
    idx, labels = synth.make_paths(128, P, T, Ve=Ve, seed=9100 + i)
    mb.append((np.ascontiguousarray(idx, np.int32), np.ascontiguousarray(labels, np.float32)))
res = {}


def timed(name, fn, k=640):
    for i in range(64):   # (two passes over the 32 minibatches: every slot has seen its largest P)
        fn(i)
    eng.sync()
    t0 = time.perf_counter()
    for i in range(k):
        fn(i)
    eng.sync()
    res[name] = round(1e3 * (time.perf_counter() - t0) / k, 4)


resident = [eng.batch(i, l) for i, l in mb]
timed("resident_loss_every_step_ms", lambda i: eng.train_step(resident[i % 32], opt, 1, want_loss=True))
timed("resident_no_loss_ms", lambda i: eng.train_step(resident[i % 32], opt, 1, want_loss=False))
slots = [None, None]
def fed(i):
    slots[i & 1] = eng.feed(*mb[i % 32], slot=slots[i & 1])
    eng.train_step(slots[i & 1], opt, 1, want_loss=True)
timed("feed_async_then_step_ms", fed)
timed("host_buffer_entry_ms", lambda i: eng.train_step_host(*mb[i % 32], opt))
for rep in ("", "_again"):
    eng.set_option("train_step_return", "drain")   # rounds 1-4: the call returns after the whole step
    timed("host_buffer_entry_drain_ms" + rep, lambda i: eng.train_step_host(*mb[i % 32], opt))
    timed("resident_loss_every_step_drain_ms" + rep, lambda i: eng.train_step(resident[i % 32], opt, 1, want_loss=True))
    eng.set_option("train_step_return", "loss")
    timed("host_buffer_entry_loss_ms" + rep, lambda i: eng.train_step_host(*mb[i % 32], opt))
    timed("resident_loss_every_step_loss_ms" + rep, lambda i: eng.train_step(resident[i % 32], opt, 1, want_loss=True))
for rep in ("", "_again"):   # the inline feed's upload: on the upload stream beside the previous step's backward (default) or in stream order
    eng.set_option("inline_upload", "main")
    timed("host_buffer_entry_upload_in_stream_order_ms" + rep, lambda i: eng.train_step_host(*mb[i % 32], opt))
    eng.set_option("inline_upload", "side")
    timed("host_buffer_entry_upload_beside_ms" + rep, lambda i: eng.train_step_host(*mb[i % 32], opt))
sc = []
for i in range(16):
    P = int(min(rng.geometric(0.57), 28))
    sc.append(np.ascontiguousarray(synth.make_paths(512, P, T, Ve=Ve, seed=9300 + i)[0], np.int32))
timed("score_512_probs_only_ms", lambda i: eng.forward_host(sc[i % 16], 1, want_all=False))
timed("score_512_probs_and_all_classes_ms", lambda i: eng.forward_host(sc[i % 16], 1, want_all=True))
t0 = time.perf_counter()
for i in range(2000):
    eng.sync()
res["python_sync_call_us"] = round(1e6 * (time.perf_counter() - t0) / 2000, 2)
eng.profile_reset(); eng.set_option("profile_filter", ""); eng.profile(True)
for i in range(64):
    eng.train_step_host(*mb[i % 32], opt)
eng.sync(); eng.profile(False)
fam = {k: round(v[0] / 64, 5) for k, v in sorted(eng.profile_get().items(), key=lambda kv: -kv[1][0])}
res["kernel_ms_per_step_by_family"] = fam
res["kernel_ms_sum"] = round(sum(fam.values()), 4)
print(json.dumps(res))
