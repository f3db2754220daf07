"""CPU: the host-side derivation of a batch's identical-prefix plan and entity-occurrence index (kprn_amd/csrc/host_feed.hip, what
the streaming feed's worker threads compute; kprn_host_batch_index needs no GPU) against a numpy restatement of the device
builders' definitions (kprn_amd/csrc/batch_index.hip: k_find_ref, k_prefix_len, stable sort by prefix length, k_prefix_apply,
k_keys, stable radix sort by entity row, run-length encode, k_drop_sentinel).  tests/test_gpu_feed.py checks on the GPU that a
host-fed slot and a device-built batch behave identically."""
import ctypes as C

import numpy as np
import pytest

from kprn_amd import _ffi, synth

KCAP = 8


def host_index(idx, nT, Vt, Ve, Vr, plan, threads=3):
    L = _ffi.lib()
    idx = np.ascontiguousarray(idx, np.int32)
    B, P, T, F = idx.shape
    N, n_index = B * P, B * P * T + (KCAP if plan else 0)
    out = {"idx_s": np.zeros((N, T, F), np.int32), "perm": np.zeros(N, np.int32), "slot_of": np.zeros(N, np.int32),
           "tile_k": np.zeros((N + 63) // 64, np.int32), "pmeta": np.zeros(24, np.int32), "key_sorted": np.zeros(n_index, np.int32),
           "pos_sorted": np.zeros(n_index, np.int32), "uniq": np.zeros(n_index + 4, np.int32)}
    summary = np.zeros(4, np.int64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.kprn_host_batch_index(p(idx), B, P, T, F, nT, Vt, Ve, Vr, int(plan), threads, p(out["idx_s"]), p(out["perm"]), p(out["slot_of"]),
                                 p(out["tile_k"]), p(out["pmeta"]), p(out["key_sorted"]), p(out["pos_sorted"]), p(out["uniq"]), p(summary))
    assert rc == 0
    out.update(bad=bool(summary[0]), kmax=int(summary[1]), n_uniq=int(summary[2]), exec_steps=int(summary[3]))
    return out


def numpy_index(idx, nT, Ve, plan):
    B, P, T, F = idx.shape
    N = B * P
    flat = idx.reshape(N, T, F).astype(np.int64)
    c0 = F - nT - 2
    r = {}
    src = flat
    tile_k = None
    kmax_batch = 0
    qent = 0
    if plan:
        same = (flat[:, 0, c0:] == flat[:, 1, c0:]).all(axis=1)
        k = np.zeros(N, np.int64)
        pmeta = np.zeros(24, np.int64)
        pmeta[1] = 0x7fffffff
        if same.any():
            ref = int(np.argmax(same))
            q = flat[ref, 0]
            pmeta[1] = ref
            pmeta[8:8 + F] = q
            lim = min(T - 2, KCAP)
            alive = np.ones(N, bool)
            for t in range(lim):
                alive &= (flat[:, t, c0:] == q[c0:]).all(axis=1)
                k += alive
        perm = np.argsort(k, kind="stable")
        ks = k[perm]
        src = flat[perm]
        slot_of = np.empty(N, np.int64)
        slot_of[perm] = np.arange(N)
        tile_k = ks[::64]
        pmeta[0] = ks[-1]
        kmax_batch = int(ks[-1])
        qent = int(pmeta[8 + F - 2])
        r.update(idx_s=src, perm=perm, slot_of=slot_of, tile_k=tile_k, pmeta=pmeta)
    ent = src[:, :, F - 2] - 1
    pos = np.arange(N * T).reshape(N, T)
    if plan:
        skipped = np.arange(T)[None, :] < np.repeat(tile_k, 64)[:N, None]
        ent = np.where(skipped, Ve, ent)
    keys = ent.ravel()
    vals = pos.ravel()
    if plan:
        npad = (N + 63) // 64 * 64
        vk = np.array([qent - 1 if t < kmax_batch else Ve for t in range(KCAP)], np.int64)
        keys = np.concatenate([keys, vk])
        vals = np.concatenate([vals, npad * T + np.arange(KCAP)])
    order = np.argsort(keys, kind="stable")
    r["key_sorted"] = keys[order]
    r["pos_sorted"] = vals[order]
    u = np.unique(keys)
    r["uniq"] = u[u != Ve]
    r["exec_steps"] = N * T - (int((np.repeat(tile_k, 64)[:N]).sum()) if plan else 0)
    r["kmax"] = kmax_batch
    return r


def _compare(idx, nT=1, Vt=6, Ve=300, Vr=9, plan=True, threads=3):
    got = host_index(idx, nT, Vt, Ve, Vr, plan, threads)
    want = numpy_index(idx, nT, Ve, plan)
    assert not got["bad"]
    names = ["key_sorted", "pos_sorted"] + (["idx_s", "perm", "slot_of", "tile_k"] if plan else [])
    for nm in names:
        assert np.array_equal(got[nm].reshape(-1), np.asarray(want[nm]).reshape(-1)), nm
    if plan:
        F = idx.shape[3]
        assert np.array_equal(got["pmeta"][:2], want["pmeta"][:2]) and np.array_equal(got["pmeta"][8:8 + F], want["pmeta"][8:8 + F])
    assert got["n_uniq"] == len(want["uniq"]) and np.array_equal(got["uniq"][:got["n_uniq"]], want["uniq"])
    assert got["exec_steps"] == want["exec_steps"] and got["kmax"] == want["kmax"]
    return got


@pytest.mark.parametrize("plan", [True, False])
@pytest.mark.parametrize("pairs,P,T", [(700, 3, 6), (37, 7, 3), (1, 1, 6), (130, 2, 12), (64, 1, 2), (33, 28, 4)])
def test_host_index_matches_the_device_builders_definition(pairs, P, T, plan):
    idx, _ = synth.make_paths(pairs, P, T, Ve=300, seed=pairs + P + T)
    got = _compare(idx, plan=plan)
    if plan and T >= 4:
        assert got["kmax"] > 0 and got["exec_steps"] <= pairs * P * T


def test_no_padded_path_means_no_plan_effect():
    idx, _ = synth.make_paths(200, 2, 6, Ve=3000000, seed=5, real_len=6)
    idx[..., 1] = 1 + np.arange(400 * 6).reshape(200, 2, 6)   # (no step repeats the ids of the one before it, hub entities included)
    got = _compare(idx, Ve=3000000, plan=True)
    assert got["kmax"] == 0 and got["exec_steps"] == 200 * 2 * 6 and got["pmeta"][1] == 0x7fffffff
    assert np.array_equal(got["perm"], np.arange(400))


def test_deep_padding_is_capped_and_two_type_slots():
    idx, _ = synth.make_paths(150, 3, 12, F=4, Ve=300, num_types=2, seed=8, real_len=2)   # 10 pad steps: capped at min(T-2, 8) = 8
    got = _compare(idx, nT=2, plan=True)
    assert got["kmax"] == 8
    idx, _ = synth.make_paths(90, 2, 6, F=6, Ve=300, num_types=2, seed=9)                  # leading feature columns are ignored
    _compare(idx, nT=2, plan=True)


def test_thread_count_does_not_change_the_result():
    idx, _ = synth.make_paths(900, 4, 6, Ve=50000, seed=3)
    a = host_index(idx, 1, 6, 50000, 9, True, threads=1)
    for th in (2, 5, 8):
        b = host_index(idx, 1, 6, 50000, 9, True, threads=th)
        for nm in ("key_sorted", "pos_sorted", "idx_s", "perm", "slot_of", "tile_k", "uniq"):
            assert np.array_equal(a[nm], b[nm]), (nm, th)


def test_ids_above_2_pow_24_and_out_of_range_ids():
    Ve = 20_000_000   # BASELINE configs[3]: beyond float32's exact integers
    idx, _ = synth.make_paths(300, 2, 6, Ve=Ve, Vr=100, seed=4)
    idx[5, 1, 4, 1] = Ve - 3
    idx[6, 0, 5, 1] = (1 << 24) + 1
    got = _compare(idx, Ve=Ve, Vr=100, plan=True)
    assert (1 << 24) in got["uniq"][:got["n_uniq"]]
    for col, v in ((1, Ve + 1), (1, 0), (0, 7), (2, 101)):
        bad = idx.copy()
        bad[2, 1, 3, col] = v
        assert host_index(bad, 1, 6, Ve, 100, True)["bad"]
