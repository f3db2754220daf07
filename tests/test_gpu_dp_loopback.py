"""-m gpu: the engine's OWN world > 1 exchange (kprn_dp_exchange_begin / _finish: in-place slots of the gathered buffer, the dp_comm_stream
hand-over, k_union_adam against W real id lists, a scoring pass queued between begin and finish) on ONE GPU, through a loopback
communicator.

No multi-GPU node was available to any round of this build, so the RCCL function table of kprn_api.hip (dlopen'ed from the path the caller
hands kprn_dp_init) is pointed at tests/loopback_rccl.cpp: an in-process stand-in whose ncclAllGather is a rendezvous of W host threads
followed by device-to-device copies of the peers' slots.  Everything on the engine's side of the six ncclXxx entry points runs exactly as it
will under RCCL at W = 2 / 4 / 8.  Checked after several Adam steps on two alternating batch sets (rows skip steps: the lazy replay runs):
  * every replica holds bit-identical parameters AND Adam state (rank-ordered sums: DESIGN.md section 4);
  * they equal ONE handle fed the global minibatch to fp32 reordering (MapReduce.lua:24-47 pairs are independent, MyOptimizer.lua:184-218);
  * the collective really ran once per step with W ranks (the stub counts), and kprn_dp_comm_size reports W.
Each case runs in its own process: librccl is bound once per process (the other GPU tests bind the real one)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "loopback_rccl.cpp")
LIB = os.path.join(ROOT, "tests", "_build", "librccl_loopback.so")


def build_loopback():
    if os.path.exists(LIB) and os.path.getmtime(LIB) > os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950", SRC, "-o", LIB])
    return LIB


_RUN = textwrap.dedent("""
    import sys, json, threading, ctypes, numpy as np
    sys.path.insert(0, %(root)r)
    from kprn_amd import _ffi, synth
    W, steps, comm_stream, score_under, cfgname = %(W)d, %(steps)d, %(comm_stream)d, %(score_under)d, %(cfg)r
    LOOP = %(lib)r
    if cfgname == "c2":
        shape = dict(Vt=6, Ve=3000, Vr=9, dt=16, de=32, dr=16, H=64, L=2)
    else:   # the generic pipeline (another forward / backward / union path): one layer, de = 64
        shape = dict(Vt=6, Ve=3000, Vr=9, dt=16, de=64, dr=16, H=64, L=1)
    per = 10                                     # pairs per rank per step
    sets = [synth.make_paths(per * W, 3, 6, Ve=shape["Ve"], seed=41 + k) for k in range(2)]
    opt = _ffi.make_opt(method=1, lr=1e-2)
    def mk(rank, world):
        e = _ffi.Engine(shape["Vt"], shape["Ve"], shape["Vr"], shape["dt"], shape["de"], shape["dr"], shape["H"], shape["L"], rank=rank, world=world,
                        param_init=0.1, seed=777)
        if cfgname != "c2":
            e.set_option("impl", "generic")
        return e
    ref = mk(0, 1)
    theta = ref.get_flat_params()
    reps = [mk(r, W) for r in range(W)]
    for e in reps:
        e.set_flat_params(theta)
    uid = _ffi.dp_unique_id(LOOP)
    for r, e in enumerate(reps):
        e.dp_init(uid, r, W, LOOP)
        e.set_option("dp_comm_stream", str(comm_stream))
        if score_under:
            e.set_option("score_overlap", "1")
            e.set_option("reserve_cus", "16")
    assert all(e.dp_comm_size() == W for e in reps)
    bref = [ref.batch(idx, lab) for idx, lab in sets]
    shards = [[reps[r].batch(idx[r * per:(r + 1) * per], lab[r * per:(r + 1) * per]) for r in range(W)] for idx, lab in sets]
    cap = (max(b.n_uniq for s in shards for b in s) + 3) // 4 * 4
    errors = []
    def rank_step(r, which):
        try:
            e, b = reps[r], shards[which][r]
            e.zero_pad_tokens()                                                   # MyOptimizer.lua:181
            e.backward(b, 1, False, 1.0 / (per * W), want_loss=False)             # loss scaled by the GLOBAL minibatch
            e.dp_exchange_begin(cap)                                              # pack into this rank's slot + the all-gather, in place
            if score_under:
                e.forward_async(b, 1)                                             # a scoring pass under the collective, as bench.py --gpus N queues it
            e.dp_exchange_finish(opt)                                             # dense sum in rank order + union inside the row update
            e.sync()
        except Exception as ex:   # noqa: BLE001
            errors.append((r, repr(ex)))
    probs_ref = None
    for step in range(steps):
        which = 0 if step %% 3 != 1 else 1
        ref.train_step(bref[which], opt)
        th = [threading.Thread(target=rank_step, args=(r, which)) for r in range(W)]
        for t in th: t.start()
        for t in th: t.join()
        assert not errors, errors
    out = {"W": W}
    flats = [e.get_flat_params() for e in reps]
    out["replicas_bit_identical"] = bool(all(np.array_equal(flats[0], f) for f in flats[1:]))
    ms = [(e.get_flat_opt_state(0), e.get_flat_opt_state(1)) for e in reps]
    out["adam_state_bit_identical"] = bool(all(np.array_equal(ms[0][0], m) and np.array_equal(ms[0][1], v) for m, v in ms[1:]))
    a = ref.get_flat_params()
    out["max_abs_vs_one_handle"] = float(np.max(np.abs(a - flats[0])))
    out["moved"] = float(np.max(np.abs(a - theta)))
    out["finite"] = bool(np.all(np.isfinite(flats[0])))
    out["allgathers"] = int(ctypes.CDLL(LOOP).kprn_loopback_allgathers())
    # ... and a scoring pass on every replica gives the probabilities the single handle gives on the same pairs
    pr = ref.forward(bref[0], 1)["probs"]
    got = np.concatenate([reps[r].forward(shards[0][r], 1)["probs"] for r in range(W)])
    out["probs_max_abs"] = float(np.max(np.abs(pr - got)))
    for e in reps:
        e.dp_shutdown()
    print(json.dumps(out))
""")


@pytest.mark.parametrize("W,comm_stream,score_under,cfg", [(2, 0, 0, "c2"), (2, 1, 1, "c2"), (4, 1, 1, "c2"), (8, 1, 1, "c2"), (8, 0, 0, "c2"), (3, 1, 0, "generic")])
def test_world_gt_1_exchange_through_a_loopback_communicator(W, comm_stream, score_under, cfg):
    lib = build_loopback()
    steps = 6
    code = _RUN % dict(root=ROOT, W=W, steps=steps, comm_stream=comm_stream, score_under=score_under, cfg=cfg, lib=lib)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2500:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["finite"] and out["moved"] > 1e-3                       # the steps did move the parameters
    assert out["replicas_bit_identical"], out                          # rank-ordered sums: identical bits on every replica
    assert out["adam_state_bit_identical"], out
    assert out["allgathers"] == steps, out                             # one collective per step, all W ranks in it
    assert out["max_abs_vs_one_handle"] < 3e-5, out                    # = one handle fed the global minibatch, to fp32 reordering
    assert out["probs_max_abs"] < 1e-5, out
