"""-m gpu: parity of the HIP path (through the C ABI, include/kprn.h) against the CPU oracle.

Bars (BASELINE.json north_star): indexing bit-exact; fp32 scores within 1e-4 relative of the
float64 oracle.  Gradients / parameters after training are compared relative to the tensor's
largest magnitude (fp32 accumulation order differs; atomics make the order run-dependent).
"""
import os

import numpy as np
import pytest

from kprn_amd import _ffi, synth
from oracle.oracle import Oracle, make_cfg, make_opt

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _sixty_four_path_tiles(monkeypatch):
    """This module pins the 64-path tiles and their identical-prefix plan on SMALL shapes (fast, every edge case); left alone the engine runs batches
    of <= 8 192 paths on tiles of one 16-row m-tile without a plan -- tests/test_gpu_small_tiles.py covers that mode."""
    monkeypatch.setenv("KPRN_SMALL_TILES", "0")

SCORE_RTOL = 1e-4   # north_star: fp32 scores within 1e-4 relative
GRAD_RTOL = 2e-4    # max|g_gpu - g_f64| / max|g_f64| per tensor


def mk(Vt=6, Ve=300, Vr=9, dt=16, de=32, dr=16, H=64, L=2, F=3, nT=1, reducer=2, K=5, impl="auto", seed=1, init=0.1, compute_dtype=0):
    eng = _ffi.Engine(Vt, Ve, Vr, dt, de, dr, H, L, F=F, num_types=nT, reducer=reducer, K=K, compute_dtype=compute_dtype)
    eng.set_option("impl", impl)
    ocfg = make_cfg(Vt=Vt, Ve=Ve, Vr=Vr, dt=dt, de=de, dr=dr, F=F, numTypes=nT, H=H, L=L, reducer=reducer, K=K)
    o64 = Oracle(ocfg, np.float64)
    theta = o64.init_params(seed, init)
    eng.set_flat_params(theta.astype(np.float32))
    theta = theta.astype(np.float32).astype(np.float64)  # the oracle sees exactly the fp32 values
    return eng, o64, theta


def rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(1e-30, np.max(np.abs(b))))


IMPLS = ["generic", "auto"]


def test_param_roundtrip_and_layout():
    eng, o64, theta = mk()
    got = eng.get_flat_params()
    assert np.array_equal(got, theta.astype(np.float32))
    lay_e, lay_o = eng.layout(), o64.layout()
    assert list(lay_e.keys()) == list(lay_o.keys())
    for nm in lay_e:
        assert lay_e[nm] == lay_o[nm]
        off, shp = lay_e[nm]
        n = int(np.prod(shp))
        assert np.array_equal(eng.get_param(nm).ravel(), theta[off:off + n].astype(np.float32))
    w = np.arange(46 * 64, dtype=np.float32).reshape(46, 64)
    eng.set_param("out.weight", w)
    assert np.array_equal(eng.get_param("out.weight"), w)
    off = lay_e["out.weight"][0]
    assert np.array_equal(eng.get_flat_params()[off:off + w.size], w.ravel())


@pytest.mark.parametrize("nT,F", [(1, 3), (2, 4), (2, 6)])
def test_embedding_gather_is_bit_exact(nT, F):
    eng, o64, theta = mk(F=F, nT=nT, L=1)
    idx, _ = synth.make_paths(40, 3, 6, F=F, Ve=300, num_types=nT, seed=4)
    x = eng.embed(idx)
    want = Oracle(o64.cfg, np.float32).embed(theta.astype(np.float32), idx)
    assert np.array_equal(x, want)  # indexing + concat order bit-exact (FeatureEmbedding.lua:118)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("L,P,T", [(1, 1, 6), (2, 3, 6), (2, 7, 3), (1, 28, 4)])
def test_forward_matches_oracle(impl, L, P, T):
    eng, o64, theta = mk(L=L, impl=impl)
    idx, _ = synth.make_paths(37, P, T, Ve=300, seed=5)
    out = eng.forward(eng.batch(idx), 1, want=("probs", "all_probs", "pooled", "path_scores"))
    ps, pooled, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5
    np.testing.assert_allclose(out["pooled"], pooled, rtol=SCORE_RTOL, atol=2e-6)
    np.testing.assert_allclose(out["all_probs"], probs, rtol=SCORE_RTOL)
    np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=SCORE_RTOL)
    # the host-pointer entry point gives the same answer
    p2, all2 = eng.forward_host(idx, 1)
    assert np.array_equal(p2, out["probs"]) and np.array_equal(all2, out["all_probs"])


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("reducer", [0, 1, 2])
def test_reducers(impl, reducer):
    eng, o64, theta = mk(L=1, reducer=reducer, K=2, impl=impl)
    idx, labels = synth.make_paths(23, 5, 6, Ve=300, seed=6)
    b = eng.batch(idx, labels)
    out = eng.forward(b, 3, want=("probs", "pooled"))
    ps, pooled, probs = o64.forward(theta, idx)
    np.testing.assert_allclose(out["pooled"], pooled, rtol=SCORE_RTOL, atol=2e-6)
    np.testing.assert_allclose(out["probs"], probs[:, 2], rtol=SCORE_RTOL)
    loss = eng.backward(b, 3)
    ol, og, _ = o64.forward_backward(theta, idx, labels, class_id=3)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    assert rel_inf(g, og) < GRAD_RTOL


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("L,P,nT,F", [(1, 2, 1, 3), (2, 4, 1, 3), (2, 3, 2, 4)])
def test_backward_matches_oracle(impl, L, P, nT, F):
    eng, o64, theta = mk(L=L, F=F, nT=nT, impl=impl)
    idx, labels = synth.make_paths(41, P, 6, F=F, Ve=300, num_types=nT, seed=7)
    b = eng.batch(idx, labels)
    for literal in (False, True):
        loss = eng.backward(b, 1, bce_literal=literal)
        ol, og, _ = o64.forward_backward(theta, idx, labels, class_id=1, bce_literal=literal)
        assert abs(loss - ol) < 1e-5 * max(1, abs(ol)), (loss, ol)
        g = eng.get_flat_grads()
        for nm, (off, shp) in eng.layout().items():
            n = int(np.prod(shp))
            r = rel_inf(g[off:off + n], og[off:off + n])
            assert r < GRAD_RTOL, (nm, r, literal)
        # zeroGradParameters: a second backward does not accumulate (MyOptimizer.lua:186)
    g2 = eng.get_grad("entity_emb")
    eng.backward(b, 1, bce_literal=True)
    g3 = eng.get_grad("entity_emb")
    assert rel_inf(g3, g2.astype(np.float64)) < 1e-5


@pytest.mark.parametrize("Vt,Vr,nT,F,dt,de,dr,H", [(40, 70, 2, 4, 16, 32, 16, 64), (100, 9, 1, 3, 48, 32, 16, 96), (6, 128, 3, 5, 20, 40, 36, 96)])
def test_table_gradients_on_the_matrix_cores_for_larger_tables_and_several_type_slots(Vt, Vr, nT, F, dt, de, dr, H):
    """kk::k_table_grad_mfma (generic pipelines): tables of more than 16 rows take several 16-row tiles (tiles nobody's id falls into are
    skipped), several type slots send the same dx to every slot's row, slices that are not a multiple of 16 columns mask their tail lanes --
    type_emb / relation_emb gradients against the f64 oracle (the named configs only exercise <= 16 rows or one slot)."""
    eng, o64, theta = mk(Vt=Vt, Vr=Vr, dt=dt, de=de, dr=dr, H=H, L=1, F=F, nT=nT, impl="generic")
    idx, labels = synth.make_paths(300, 3, 6, F=F, Vt=Vt, Ve=300, Vr=Vr, num_types=nT, seed=17)
    b = eng.batch(idx, labels)
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels, class_id=1)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < GRAD_RTOL, nm
    # row by row: a misplaced row tile would hide in a matrix norm
    for nm in ("type_emb", "relation_emb"):
        off, shp = eng.layout()[nm]
        ge = g[off:off + int(np.prod(shp))].reshape(shp).astype(np.float64)
        go = og[off:off + int(np.prod(shp))].reshape(shp)
        scale = np.abs(go).max()
        assert np.max(np.abs(ge - go)) < 2e-5 * scale, nm
        assert np.count_nonzero(np.abs(go).sum(axis=1)) >= min(shp[0], 8) - 2  # (several rows really received gradient; the last id of a vocabulary is unused by the generator)


@pytest.mark.parametrize("impl", IMPLS)
def test_odd_sizes_shipped_config_shape(impl):
    """run_scripts/config.sh: d = 50/100/50, H = 250 (not a multiple of 16), L = 1."""
    eng, o64, theta = mk(dt=50, de=100, dr=50, H=250, L=1, Ve=120, impl=impl, init=0.05)
    idx, labels = synth.make_paths(9, 3, 6, Ve=120, seed=8)
    b = eng.batch(idx, labels)
    out = eng.forward(b, 1, want=("probs", "path_scores"))
    ps, pooled, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 3e-5
    np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=SCORE_RTOL)
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert rel_inf(eng.get_flat_grads(), og) < GRAD_RTOL


def _train_compare(eng, o64, theta, batches, opt_kw, steps, tol):
    oopt = make_opt(**{k: v for k, v in opt_kw.items() if k != "entity_update"})
    gopt = _ffi.make_opt(**opt_kw)
    th = theta.copy()
    st = o64.new_state()
    gb = [eng.batch(i, l) for i, l in batches]
    for s in range(steps):
        i, l = batches[s % len(batches)]
        ol, _ = o64.train_step(th, st, oopt, i, l)
        gl = eng.train_step(gb[s % len(batches)], gopt)
        assert abs(gl - ol) < 2e-4 * max(1.0, abs(ol)), (s, gl, ol)
    got = eng.get_flat_params()
    d = float(np.max(np.abs(got - th)))
    assert d < tol, d
    return got, th


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("method,regularize", [(1, 0), (0, 0), (1, 1), (0, 1)])
def test_train_steps_match_oracle(impl, method, regularize):
    eng, o64, theta = mk(L=2, impl=impl)
    batches = [synth.make_paths(32, P, 6, Ve=300, seed=20 + P) for P in (1, 3, 2)]
    kw = dict(method=method, lr=1e-2, lr_decay=0.0167, regularize=regularize, use_grad_clip=1, grad_clip_norm=0.02, l2=1e-3)
    # 30 steps with lr 1e-2: parameters move by up to ~0.3; fp32-vs-f64 drift stays well below 1e-3
    _train_compare(eng, o64, theta, batches, kw, 30, 2e-4)
    lay = eng.layout()
    for nm, V in (("type_emb", 6), ("entity_emb", 300), ("relation_emb", 9)):
        assert np.all(eng.get_param(nm)[V - 1] == 0)  # zeroPadTokens (MyOptimizer.lua:74-93)
    assert lay


@pytest.mark.parametrize("impl", IMPLS)
def test_lazy_entity_update_matches_the_dense_sweep_to_rounding(impl):
    """the lazy-exact row update against optim.adam's dense sweep over 24 steps, including rows that are touched once and then coast
    on momentum for many steps: equal to fp32 rounding (2e-6) -- the gradients come from fp32 atomics whose order differs run to run;
    the BITWISE statement is test_lazy_replay_exactness_without_atomics below."""
    res = []
    for mode in (0, 1):
        eng, o64, theta = mk(L=1, impl="generic", Ve=400)  # deterministic generic path for the bitwise check
        opt = _ffi.make_opt(method=1, lr=5e-3, entity_update=mode)
        batches = [eng.batch(*synth.make_paths(8, 2, 6, Ve=400, seed=40 + k)) for k in range(6)]
        # entity grads are accumulated with atomics: order can differ run to run, so bitwise identity is
        # asserted on the optimiser given identical gradients -> use single-path pairs with distinct rows
        for s in range(24):
            eng.train_step(batches[s % 6], opt)
        res.append((eng.get_param("entity_emb"), eng.get_flat_opt_state(0), eng.get_flat_opt_state(1)))
    W0, m0, v0 = res[0]
    W1, m1, v1 = res[1]
    # gradients come from fp32 atomics (run-dependent order), so compare to rounding, not bitwise;
    # the bitwise statement is test_lazy_replay_exactness_without_atomics below
    np.testing.assert_allclose(W0, W1, rtol=0, atol=2e-6)
    np.testing.assert_allclose(m0, m1, rtol=0, atol=2e-6 * float(np.max(np.abs(m1))))
    np.testing.assert_allclose(v0, v1, rtol=0, atol=2e-6 * float(np.max(np.abs(v1))))
    assert impl in IMPLS


def test_lazy_replay_exactness_without_atomics():
    """one pair, one path, distinct entities per step => no atomic reordering => the lazy and the
    dense entity update must agree BITWISE after rows coast for 40 steps."""
    outs = []
    for mode in (0, 1):
        eng, o64, theta = mk(L=1, impl="generic", Ve=64)
        opt = _ffi.make_opt(method=1, lr=5e-3, entity_update=mode)
        idx0, l0 = synth.make_paths(1, 1, 6, Ve=30, seed=1)
        idx1, l1 = synth.make_paths(1, 1, 6, Ve=30, seed=2)
        idx0 = idx0.copy(); idx0[0, 0, :, 1] = np.arange(1, 7)      # six distinct rows: no atomic ever sees > 1 addend
        idx1 = idx1.copy(); idx1[0, 0, :, 1] = np.arange(31, 37)    # a disjoint set
        b0, b1 = eng.batch(idx0, l0), eng.batch(idx1, l1)
        eng.train_step(b0, opt)
        for _ in range(40):
            eng.train_step(b1, opt)
        eng.train_step(b0, opt)
        outs.append((eng.get_param("entity_emb"), eng.get_flat_opt_state(0), eng.get_flat_opt_state(1)))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)


def test_errors_are_codes_not_crashes():
    eng, o64, theta = mk(L=1)
    idx, labels = synth.make_paths(4, 2, 6, Ve=300, seed=3)
    bad = idx.copy(); bad[0, 0, 0, 1] = 301
    with pytest.raises(_ffi.KprnError) as e:
        eng.batch(bad, labels)
    assert e.value.code == _ffi.E_INDEX
    bad = idx.copy(); bad[1, 1, 2, 2] = 0
    with pytest.raises(_ffi.KprnError) as e:
        eng.batch(bad, labels)
    assert e.value.code == _ffi.E_INDEX
    with pytest.raises(_ffi.KprnError) as e:
        eng.forward(eng.batch(idx), 47)
    assert e.value.code == _ffi.E_ARG
    with pytest.raises(_ffi.KprnError) as e:
        eng.backward(eng.batch(idx), 1)  # no targets: assert(targets) MyOptimizer.lua:179
    assert e.value.code == _ffi.E_ARG
    with pytest.raises(_ffi.KprnError) as e:
        _ffi.Engine(6, 100, 9, 4, 8, 4, 32, L=2)  # D != H with 2 layers (OneModel.lua:236,270-273)
    assert e.value.code == _ffi.E_ARG
    with pytest.raises(_ffi.KprnError) as e:
        _ffi.Engine(6, 100, 9, 4, 8, 4, 16, rnn_type=3)  # no such cell
    assert e.value.code == _ffi.E_ARG
    with pytest.raises(_ffi.KprnError) as e:
        eng.get_param("nope")
    assert e.value.code == _ffi.E_ARG
    # the handle is still usable after errors
    assert eng.forward(eng.batch(idx), 1)["probs"].shape == (4,)


def test_checkpoint_roundtrip(tmp_path):
    eng, o64, theta = mk(L=2)
    idx, labels = synth.make_paths(16, 2, 6, Ve=300, seed=3)
    b = eng.batch(idx, labels)
    opt = _ffi.make_opt()
    for _ in range(3):
        eng.train_step(b, opt)
    p = os.path.join(tmp_path, "model-latest.kprn")
    eng.save(p)
    want = eng.forward(b, 1)["probs"]
    flat = eng.get_flat_params()
    eng2, _, _ = mk(L=2, seed=99)
    eng2.load(p)
    assert np.array_equal(eng2.get_flat_params(), flat)
    assert np.array_equal(eng2.forward(eng2.batch(idx), 1)["probs"], want)
    eng3 = _ffi.Engine(6, 300, 9, 16, 32, 16, 64, 1)
    with pytest.raises(_ffi.KprnError) as e:
        eng3.load(p)
    assert e.value.code == _ffi.E_IO


@pytest.mark.parametrize("impl", IMPLS)
def test_two_replicas_exchange_equals_one_big_batch(impl):
    """data-parallel hooks on ONE GPU: two handles each take half the pairs, exchange gradients
    through the pack / merge API (device pointers), and must end where a single handle fed
    the whole minibatch ends.  (The RCCL transport itself is exercised by the driver's N>1 runs.)"""
    import torch
    from kprn_amd import dp
    idx, labels = synth.make_paths(24, 3, 6, Ve=300, seed=11)
    opt = _ffi.make_opt(method=1, lr=1e-2)
    ref, o64, theta = mk(L=2, impl=impl)
    bref = ref.batch(idx, labels)
    reps = [mk(L=2, impl=impl)[0] for _ in range(2)]
    for e in reps:
        e.stream()   # (the exchange hooks want a caller that knows the engine's stream; this test orders by full synchronisation)
    halves = [reps[r].batch(idx[r * 12:(r + 1) * 12], labels[r * 12:(r + 1) * 12]) for r in range(2)]
    dev = "cuda:0"
    for step in range(4):
        ref.train_step(bref, opt)
        cap = 0
        packed = []
        for r in range(2):
            reps[r].zero_pad_tokens()
            reps[r].backward(halves[r], 1, False, 1.0 / 24.0, want_loss=False)
            cap = max(cap, reps[r].sparse_grad_capacity())
        # dense all-reduce by hand
        dens = []
        for r in range(2):
            reps[r].sync()
            ptr, n = reps[r].dense_grad_buffer()
            dens.append(dp.wrap_device(ptr, n, "f32", dev))
        torch.cuda.synchronize()
        tot = dens[0] + dens[1]
        for r in range(2):
            dens[r].copy_(tot)
        torch.cuda.synchronize()
        for r in range(2):
            ptr, n_words = reps[r].sparse_grad_pack(cap)
            reps[r].sync()
            packed.append(dp.wrap_device(ptr, n_words, "i32", dev).clone())
        torch.cuda.synchronize()
        allbuf = torch.cat(packed)  # what the all-gather delivers on every rank
        torch.cuda.synchronize()
        for r in range(2):
            reps[r].sparse_grad_merge(allbuf.data_ptr(), 2, cap)
            reps[r].apply_update(opt)
            reps[r].sync()
    a, b, c = ref.get_flat_params(), reps[0].get_flat_params(), reps[1].get_flat_params()
    assert np.array_equal(b, c)  # replicas stay bit-identical
    assert float(np.max(np.abs(a - b))) < 2e-5


def _exchange_steps(world, steps, de=32, dense_in_pack=True, **mkkw):
    """Two groups of `world` replicas on ONE GPU -- group A updates with the union inside the row kernel (dp_fused_update), group B with the
    separate marking merge -- in lockstep: every replica runs its own backward on its slice of the pairs (two alternating sets of
    slices, so rows skip steps and the lazy replay runs) and packs; BOTH groups then merge the buffer gathered from group A (a backward is
    reproducible only to rounding -- fp32 atomics in the weight-gradient reduce -- so the comparison hands both paths the same bits)."""
    import torch
    from kprn_amd import dp
    per = 6
    sets = [synth.make_paths(per * world, 3, 6, Ve=300, seed=21 + k) for k in range(2)]
    opt = _ffi.make_opt(method=1, lr=1e-2)
    groups = []
    for fused in (True, False):
        reps = [mk(de=de, **mkkw)[0] for _ in range(world)]
        for e in reps:
            e.stream()
            e.set_option("dp_dense_in_pack", "1" if dense_in_pack else "0")
            e.set_option("dp_fused_update", "1" if fused else "0")
        batches = [[reps[r].batch(idx[r * per:(r + 1) * per], lab[r * per:(r + 1) * per]) for r in range(world)] for idx, lab in sets]
        groups.append((reps, batches))
    dev = "cuda:0"
    for step in range(steps):
        which = 0 if step % 3 != 1 else 1
        cap = 0
        for reps, batches in groups:
            for r in range(world):
                reps[r].zero_pad_tokens()
                reps[r].backward(batches[which][r], 1, False, 1.0 / (per * world), want_loss=False)
                cap = max(cap, reps[r].sparse_grad_capacity())
        cap = (cap + 3) // 4 * 4
        if not dense_in_pack:   # the dense all-reduce by hand: group A's sum, handed to everybody
            dens = []
            for reps, _ in groups:
                for r in range(world):
                    reps[r].sync()
                    ptr, n = reps[r].dense_grad_buffer()
                    dens.append(dp.wrap_device(ptr, n, "f32", dev))
            tot = dens[0].clone()
            for r in range(1, world):
                tot += dens[r]
            for d in dens:
                d.copy_(tot)
            torch.cuda.synchronize()
        gathered = None
        for reps, _ in groups:
            packed = []
            for r in range(world):
                ptr, n_words = reps[r].sparse_grad_pack(cap)
                reps[r].sync()
                packed.append(dp.wrap_device(ptr, n_words, "i32", dev).clone())
            torch.cuda.synchronize()
            if gathered is None:
                gathered = torch.cat(packed)   # what the all-gather delivers; alive until every update has run (the fused update reads it)
        torch.cuda.synchronize()
        for reps, _ in groups:
            for r in range(world):
                reps[r].sparse_grad_merge(gathered.data_ptr(), world, cap)
                reps[r].apply_update(opt)
                reps[r].sync()
    return [[e.get_flat_params() for e in reps] for reps, _ in groups]


@pytest.mark.parametrize("world,de,kw", [(2, 32, dict(L=2)), (3, 32, dict(L=2)), (8, 32, dict(L=2, impl="generic")),
                                         (3, 64, dict(dt=16, dr=16, H=64, L=1)), (5, 32, dict(L=2, dense_in_pack=False))])
def test_union_inside_the_row_update_equals_the_separate_merge_bitwise(world, de, kw):
    """kprn_set_option("dp_fused_update"): the lowest rank's entry owns a row, finds the other ranks' contributions by binary search in their
    ascending id lists and adds them in rank order -- the same additions in the same order as the marking merge + row update, so the
    parameters (entity rows, Adam state through the lazy replay, dense arena) must agree BIT FOR BIT, on every replica."""
    kw = dict(kw)
    dip = kw.pop("dense_in_pack", True)
    a, b = _exchange_steps(world, 5, de=de, dense_in_pack=dip, **kw)
    for r in range(world):
        assert np.array_equal(a[r], a[0]) and np.array_equal(b[r], b[0])     # replicas identical
    assert np.array_equal(a[0], b[0])          # fused == separate merge
    assert np.all(np.isfinite(a[0]))
    theta0 = mk(de=de, **kw)[2].astype(np.float32)
    assert float(np.max(np.abs(a[0] - theta0))) > 1e-3   # (and they trained)


def test_a_gathered_gradient_can_still_be_read_or_dropped_before_the_update():
    """the fused path only RECORDS the union; a caller that reads the gradient (kprn_get_grad) or starts another step instead of updating
    must see what the separate merge would have left"""
    import torch
    from kprn_amd import dp
    outs = []
    for fused in (True, False):
        eng = mk(L=2)[0]
        eng.stream()
        eng.set_option("dp_dense_in_pack", "1")
        eng.set_option("dp_fused_update", "1" if fused else "0")
        idx, lab = synth.make_paths(12, 3, 6, Ve=300, seed=31)
        b = eng.batch(idx, lab)
        eng.backward(b, 1, False, 1.0 / 24.0, want_loss=False)
        cap = (eng.sparse_grad_capacity() + 3) // 4 * 4
        ptr, n_words = eng.sparse_grad_pack(cap)
        eng.sync()
        one = dp.wrap_device(ptr, n_words, "i32", "cuda:0").clone()
        allbuf = torch.cat([one, one])    # "two ranks" with the same rows: the union doubles every gradient
        torch.cuda.synchronize()
        eng.sparse_grad_merge(allbuf.data_ptr(), 2, cap)
        g = eng.get_flat_grads()
        # dropped: the next backward starts from clean accumulators
        eng.backward(b, 1, False, 1.0 / 24.0, want_loss=False)
        g2 = eng.get_flat_grads()
        # gathered again and dropped WITHOUT anybody reading it
        ptr, n_words = eng.sparse_grad_pack(cap)
        eng.sync()
        one = dp.wrap_device(ptr, n_words, "i32", "cuda:0").clone()
        allbuf = torch.cat([one, one])
        torch.cuda.synchronize()
        eng.sparse_grad_merge(allbuf.data_ptr(), 2, cap)
        eng.backward(b, 1, False, 1.0 / 24.0, want_loss=False)
        g3 = eng.get_flat_grads()
        outs.append((g, g2, g3))
    # (a backward is reproducible to rounding only -- fp32 atomics in the weight-gradient reduce -- hence tolerances, not bits)
    close = lambda x, y: np.allclose(x, y, rtol=1e-4, atol=1e-9)
    assert close(outs[0][0], outs[1][0]) and close(outs[0][1], outs[1][1])
    assert close(outs[0][2], outs[0][1]) and close(outs[1][2], outs[1][1])
    for g, g2, _ in outs:
        ent = slice(6 * 16, 6 * 16 + 300 * 32)   # entity rows: doubled by the two-rank union; (the dense arena too, dp_dense_in_pack)
        assert np.any(g2[ent] != 0) and close(g[ent], 2.0 * g2[ent])
        assert close(g, 2.0 * g2)


def test_large_batch_properties():
    """BASELINE-size shapes (T=6, D=H=64, L=2, KKBox-size entity table), checked through
    size-independent properties: (1) scoring a pair does not depend on which other pairs share
    the batch; (2) the LSE pool of a single path is the path score; (3) permuting the paths of
    a pair does not change its score."""
    eng = _ffi.Engine(6, 2851220, 9, 16, 32, 16, 64, 2)
    idx, labels = synth.make_paths(4096, 4, 6, Ve=2851220, seed=12)
    big = eng.forward(eng.batch(idx), 1, want=("probs", "pooled", "path_scores"))
    sub = eng.forward(eng.batch(idx[100:164]), 1, want=("probs",))
    np.testing.assert_allclose(big["probs"][100:164], sub["probs"], rtol=1e-6)
    one = eng.forward(eng.batch(idx[:64, :1]), 1, want=("pooled", "path_scores"))
    np.testing.assert_allclose(one["pooled"], one["path_scores"], rtol=1e-6, atol=1e-7)
    perm = idx[:64, ::-1].copy()
    pp = eng.forward(eng.batch(perm), 1, want=("probs",))
    np.testing.assert_allclose(pp["probs"], big["probs"][:64], rtol=2e-6)
    # a train step on the full-size table runs and keeps untouched rows untouched
    before = eng.get_param("entity_emb")
    opt = _ffi.make_opt()
    b = eng.batch(idx, labels)
    l0 = eng.train_step(b, opt)
    for _ in range(5):
        l1 = eng.train_step(b, opt)
    assert np.isfinite(l0) and l1 < l0
    after = eng.get_param("entity_emb")
    touched = np.zeros(2851220, bool)
    touched[np.unique(idx[..., 1]) - 1] = True
    assert np.array_equal(before[~touched], after[~touched])
    assert np.any(before[touched] != after[touched])


# ---- rnnType "rnn" (the shipped config.sh default): nn.Recurrence + nn.MaskZero, generic pipeline -------------
def mk_rnn(use_relu, L, H=48, dt=8, de=24, dr=16, Ve=300, rnn_init=False, seed=5, init=0.2):
    D = dt + de + dr
    if L > 1:
        assert D == H  # numLayers > 1 stacks Recurrence(D -> H) modules of one shape (OneModel.lua:268-273)
    eng = _ffi.Engine(6, Ve, 9, dt, de, dr, H, L, rnn_type=1, use_relu=use_relu, rnn_init=1 if rnn_init else 0, param_init=init)
    ocfg = make_cfg(Vt=6, Ve=Ve, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L, rnn_type=1, use_relu=use_relu)
    o64 = Oracle(ocfg, np.float64)
    return eng, o64


@pytest.mark.parametrize("use_relu,L", [(1, 1), (0, 1), (1, 2), (0, 2)])
def test_rnn_cell_forward_backward_match_oracle(use_relu, L):
    eng, o64 = mk_rnn(use_relu, L)
    assert list(eng.layout().keys()) == list(o64.layout().keys())
    theta = o64.init_params(5, 0.2).astype(np.float32).astype(np.float64)
    o64.zero_pad(theta)  # pad embeddings are zero => pad steps are masked (MaskZero) until the first real step
    eng.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(37, 3, 6, Ve=300, seed=8)  # 73 % of the paths carry two LEFT pad steps
    b = eng.batch(idx, labels)
    out = eng.forward(b, 1, want=("probs", "path_scores"))
    ps, pooled, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5
    np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=SCORE_RTOL)
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels, class_id=1)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < GRAD_RTOL, nm


def test_rnn_identity_init_and_training_steps():
    """-rnnInitialization 1: eye / zero-bias init (OneModel.lua:310-322); then 12 Adam steps track the oracle"""
    eng, o64 = mk_rnn(1, 2, rnn_init=True, init=0.1)
    got = eng.get_flat_params()
    lay = eng.layout()
    for l in (1, 2):
        off, (H, din) = lay[f"rnn{l}.i2h.weight"]
        assert np.array_equal(got[off:off + H * din], np.eye(din, H, dtype=np.float32).ravel())
        off, _ = lay[f"rnn{l}.h2h.weight"]
        assert np.array_equal(got[off:off + H * H], np.eye(H, dtype=np.float32).ravel())
        for nm in ("i2h.bias", "h2h.bias"):
            off, shp = lay[f"rnn{l}.{nm}"]
            assert not got[off:off + shp[0]].any()
    theta = got.astype(np.float64)
    th = theta.copy()
    st = o64.new_state()
    batches = [synth.make_paths(24, P, 6, Ve=300, seed=30 + P) for P in (2, 3)]
    gb = [eng.batch(i, l) for i, l in batches]
    opt, oopt = _ffi.make_opt(method=1, lr=2e-3), make_opt(method=1, lr=2e-3)
    for s in range(12):
        i, l = batches[s % 2]
        ol, _ = o64.train_step(th, st, oopt, i, l)
        gl = eng.train_step(gb[s % 2], opt)
        assert abs(gl - ol) < 2e-4 * max(1, abs(ol)), (s, gl, ol)
    assert float(np.max(np.abs(eng.get_flat_params() - th))) < 2e-4


@pytest.mark.parametrize("L", [1, 2])
def test_gru_cell_forward_backward_and_training_match_oracle(L):
    """rnnType gru (nn.GRU, OneModel.lua:237-238) on the generic pipeline"""
    H = 48
    eng = _ffi.Engine(6, 300, 9, 8, 24, 16, H, L, rnn_type=2, param_init=0.2)
    o64 = Oracle(make_cfg(Vt=6, Ve=300, Vr=9, dt=8, de=24, dr=16, H=H, L=L, rnn_type=2), np.float64)
    assert list(eng.layout().keys()) == list(o64.layout().keys())
    theta = o64.init_params(6, 0.2).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(37, 3, 6, Ve=300, seed=9)
    b = eng.batch(idx, labels)
    out = eng.forward(b, 1, want=("probs", "path_scores"))
    ps, _, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5
    np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=SCORE_RTOL)
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels, class_id=1)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < GRAD_RTOL, nm
    th, st = theta.copy(), o64.new_state()
    opt, oopt = _ffi.make_opt(method=1, lr=2e-3), make_opt(method=1, lr=2e-3)
    for s_ in range(8):
        ol, _ = o64.train_step(th, st, oopt, idx, labels)
        gl = eng.train_step(b, opt)
        assert abs(gl - ol) < 2e-4 * max(1, abs(ol)), (s_, gl, ol)
    assert float(np.max(np.abs(eng.get_flat_params() - th))) < 2e-4


# ---- BASELINE.json configs as parity cases (the bench line is configs[1]; the others are checked here) ---------------
def test_config4_shape_d128_fp32_generic():
    """configs[3] shape: d = 128 per slice => D = H = 384 (here L = 1, small vocabulary).  The bf16 MFMA variant of that
    config is not built yet; the fp32 generic pipeline must already be correct at this shape (it is what a bf16 path
    will be checked against)."""
    eng = _ffi.Engine(6, 500, 100, 128, 128, 128, 384, 1)
    ocfg = make_cfg(Vt=6, Ve=500, Vr=100, dt=128, de=128, dr=128, H=384, L=1)
    o64 = Oracle(ocfg, np.float64)
    theta = o64.init_params(9, 0.05).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(40, 2, 6, Ve=500, Vr=100, seed=14)
    b = eng.batch(idx, labels)
    out = eng.forward(b, 1, want=("probs", "path_scores"))
    ps, _, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5
    np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=SCORE_RTOL)
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    assert rel_inf(eng.get_flat_grads(), og) < GRAD_RTOL


@pytest.mark.parametrize("impl", IMPLS)
def test_config5_variable_length_buckets_inference(impl):
    """configs[4]: variable path length <= 7, inference only.  Paths are bucketed by identical T (left-pad semantics are
    kept inside a bucket: pads are NOT no-ops in the reference LSTM), d = 64, each bucket scored on its own."""
    eng, o64, theta = mk(L=2, impl=impl)
    for T in (3, 4, 5, 6, 7):
        idx, _ = synth.make_paths(50, 2, T, Ve=300, seed=40 + T)
        out = eng.forward(eng.batch(idx), 1, want=("probs", "path_scores"))
        ps, _, probs = o64.forward(theta, idx)
        assert rel_inf(out["path_scores"], ps) < 2e-5, T
        np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=SCORE_RTOL)


def test_config4_bf16_compute_is_tolerance_gated_against_the_f64_oracle():
    """configs[3] "bf16 MFMA LSTM": compute_dtype = 1 multiplies in bf16 (operands rounded to nearest even), accumulates in f32.
    Tolerance-gated against the float64 oracle (bf16 has 8 mantissa bits; K = 768-term dot products): scores within 3e-2
    relative of the largest score, probabilities within 3e-2, gradients within 5e-2 of each tensor's largest magnitude --
    and measurably different from the fp32 path (so the test cannot pass by silently running fp32)."""
    cfg = dict(Vt=6, Ve=500, Vr=100, dt=128, de=128, dr=128, H=384, L=1)
    ocfg = make_cfg(**cfg)
    o64 = Oracle(ocfg, np.float64)
    theta = o64.init_params(9, 0.05).astype(np.float32).astype(np.float64)
    idx, labels = synth.make_paths(40, 2, 6, Ve=500, Vr=100, seed=14)
    ps, _, probs = o64.forward(theta, idx)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    outs = {}
    for mode in (0, 1):
        eng = _ffi.Engine(6, 500, 100, 128, 128, 128, 384, 1, compute_dtype=mode)
        eng.set_flat_params(theta.astype(np.float32))
        b = eng.batch(idx, labels)
        out = eng.forward(b, 1, want=("probs", "path_scores"))
        loss = eng.backward(b, 1)
        outs[mode] = (out["path_scores"].copy(), out["probs"].copy(), loss, eng.get_flat_grads())
    s_bf, p_bf, l_bf, g_bf = outs[1]
    assert rel_inf(s_bf, ps) < 3e-2
    np.testing.assert_allclose(p_bf, probs[:, 0], rtol=3e-2)
    assert abs(l_bf - ol) < 3e-2 * max(1, abs(ol))
    for nm, (off, shp) in o64.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g_bf[off:off + n], og[off:off + n]) < 5e-2, nm
    assert rel_inf(outs[0][0], ps) < 2e-5            # the fp32 path at the same shape
    assert rel_inf(s_bf, ps) > 20 * rel_inf(outs[0][0], ps)  # bf16 really ran


def test_full_size_backward_is_additive_over_pairs():
    """BASELINE-size shapes (T=6, D=H=64, L=2, KKBox-size entity table, ~16k paths per call): with the loss scale held
    fixed, the gradient of a minibatch is the SUM of the gradients of any partition of its pairs (pairs are independent
    units, MapReduce.lua:24-47) -- a size-independent check of the whole fused backward, the index-based embedding
    gradient and the head, at a size the CPU oracle cannot reach in test time."""
    eng = _ffi.Engine(6, 2851220, 9, 16, 32, 16, 64, 2)
    idx, labels = synth.make_paths(4096, 4, 6, Ve=2851220, seed=77)
    inv = 1.0 / 4096.0
    lay = eng.layout()
    dense_names = [n for n in lay if n != "entity_emb"]

    def grads(sl):
        b = eng.batch(idx[sl], labels[sl])
        loss = eng.backward(b, 1, False, inv)
        g = {n: eng.get_grad(n).astype(np.float64) for n in dense_names}
        rows = np.unique(idx[sl][..., 1]) - 1
        ge = eng.get_grad("entity_emb")[rows].astype(np.float64)
        return loss, g, rows, ge

    l_all, g_all, r_all, e_all = grads(slice(0, 4096))
    l_a, g_a, r_a, e_a = grads(slice(0, 1500))
    l_b, g_b, r_b, e_b = grads(slice(1500, 4096))
    assert abs(l_all - (l_a + l_b)) < 1e-5 * max(1.0, abs(l_all))
    for n in dense_names:
        ref = g_all[n]
        assert np.max(np.abs(ref - (g_a[n] + g_b[n]))) < 2e-5 * max(1e-30, np.max(np.abs(ref))), n
    acc = {int(r): e_a[i].copy() for i, r in enumerate(r_a)}
    for i, r in enumerate(r_b):
        acc[int(r)] = acc.get(int(r), 0) + e_b[i]
    tot = np.stack([acc[int(r)] for r in r_all])
    assert np.max(np.abs(e_all - tot)) < 2e-5 * np.max(np.abs(e_all))


@pytest.mark.parametrize("T,L", [(2, 1), (2, 2), (16, 2)])
def test_fused_kernels_at_the_edges_of_their_step_range(T, L):
    """the fused kernels cover 2 <= T <= 16 (LDS id tile); T = 2 has a single recurrent step, T = 16 fills the id tile"""
    eng, o64, theta = mk(L=L, impl="auto")
    idx, labels = synth.make_paths(45, 3, T, Ve=300, seed=50 + T)
    b = eng.batch(idx, labels)
    out = eng.forward(b, 1, want=("probs", "path_scores"))
    ps, _, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5
    np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=SCORE_RTOL)
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < GRAD_RTOL, (nm, T, L)


def test_embedding_backward_variants_agree(monkeypatch):
    """the three forms of the embedding backward are interchangeable: one-hot MFMA small tables + index gather-reduce
    (default), general scatter kernel for the small tables (KPRN_DBG=8) and for everything (KPRN_DBG=24, atomics);
    with and without the identical-prefix plan (KPRN_DBG=64: every step of every path executed)"""
    import subprocess, sys, json, textwrap
    code = textwrap.dedent("""
        import sys, json, numpy as np
        sys.path.insert(0, %r)
        from tests.test_gpu_parity import mk
        from kprn_amd import synth
        eng, o64, theta = mk(L=2, impl="auto")
        idx, labels = synth.make_paths(300, 3, 6, Ve=300, seed=61)
        eng.backward(eng.batch(idx, labels), 1)
        g = eng.get_flat_grads()
        lay = eng.layout()
        out = {}
        for nm in ("type_emb", "entity_emb", "relation_emb"):
            off, shp = lay[nm]
            out[nm] = g[off:off + int(np.prod(shp))].astype(float).tolist()
        print(json.dumps(out))
    """) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for dbg in ("0", "8", "24", "64", "16"):
        env = dict(os.environ, KPRN_DBG=dbg)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        res[dbg] = {k: np.array(v) for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items()}
    for nm in res["0"]:
        ref = res["0"][nm]
        for dbg in ("8", "24", "64", "16"):  # 64: no identical-prefix plan; 16: plan + atomic entity scatter
            assert np.max(np.abs(res[dbg][nm] - ref)) < 1e-5 * max(1e-30, np.max(np.abs(ref))), (nm, dbg)


def test_nothing_depends_on_what_hipmalloc_returns():
    """KPRN_POISON_ALLOC=1 fills every new device allocation with 0xFF bytes (NaN / -1).  Fresh pages of a new process are zero, recycled
    blocks of a long-lived one are not: scores, loss, gradients, a few training steps and a fed slot of every pipeline (fused with and
    without the identical-prefix plan, generic, wide lstm, rnn, gru) must come out the same, twice (the second pass reuses every cache)."""
    import subprocess, sys, json, textwrap
    code = textwrap.dedent("""
        import sys, json
        sys.path.insert(0, %r)
        import numpy as np
        from kprn_amd import _ffi, synth
        from oracle.oracle import Oracle, make_cfg
        out = {}
        def case(name, shape, oshape, n_pairs, **opts):
            o64 = Oracle(make_cfg(**oshape), np.float64)
            theta = o64.init_params(3, 0.1).astype(np.float32).astype(np.float64)
            idx, labels = synth.make_paths(n_pairs, 3, 6, Ve=oshape["Ve"], Vr=oshape["Vr"], seed=5)
            ps, _, probs = o64.forward(theta, idx)
            ol, og, _ = o64.forward_backward(theta, idx, labels)
            eng = _ffi.Engine(*shape, **{k: v for k, v in opts.items() if k not in ("impl", "plan")})
            eng.set_option("impl", opts.get("impl", "auto")); eng.set_option("prefix_plan", opts.get("plan", "1"))
            eng.set_flat_params(theta.astype(np.float32))
            b = eng.batch(idx, labels)
            r = []
            for _ in range(2):
                sc = eng.forward(b, 1, want=("probs", "path_scores"))["path_scores"]
                loss = eng.backward(b, 1)
                g = eng.get_flat_grads()
                r.append([float(np.abs(sc - ps).max() / np.abs(ps).max()), float(abs(loss - ol)), float(np.abs(g - og).max() / np.abs(og).max())])
            opt = _ffi.make_opt(method=1, lr=1e-2)
            tl = [eng.train_step(b, opt) for _ in range(3)]
            slot = eng.feed(idx, labels)
            fed = eng.forward(slot, 1)["probs"]
            r.append([float(np.isfinite(tl).all()), float(np.isfinite(fed).all()), float(np.abs(fed - eng.forward(b, 1)["probs"]).max())])
            out[name] = r
            eng.close()
        A = (6, 5000, 9, 16, 32, 16, 64, 2); OA = dict(Vt=6, Ve=5000, Vr=9, dt=16, de=32, dr=16, H=64, L=2)
        case("fused+plan", A, OA, 3000)
        case("fused", A, OA, 3000, plan="0")
        case("generic", A, OA, 3000, impl="generic")
        W = (6, 700, 9, 24, 40, 24, 72, 1); OW = dict(Vt=6, Ve=700, Vr=9, dt=24, de=40, dr=24, H=72, L=1)
        case("wide lstm", W, OW, 200)
        case("rnn", W, dict(OW, rnn_type=1, use_relu=0), 200, rnn_type=1, use_relu=0)
        case("gru", W, dict(OW, rnn_type=2), 200, rnn_type=2)
        print(json.dumps(out))
    """) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, KPRN_POISON_ALLOC="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    for name, (first, second, tail) in res.items():
        for e_sc, e_loss, e_g in (first, second):
            assert e_sc < 2e-5 and e_loss < 1e-5 and e_g < 2e-4, (name, first, second)
        assert tail[0] == 1.0 and tail[1] == 1.0 and tail[2] == 0.0, (name, tail)


def test_shipped_config_sh_shape_rnn_h250():
    """run_scripts/config.sh as shipped: rnnType rnn, rnnHidSize 250, embedding dims 50/100/50 (D = 200), 1 layer, ReLU,
    identity initialisation, LogSumExp pool, Adam -- odd sizes for every GEMM edge (small vocabulary here)."""
    eng = _ffi.Engine(6, 400, 9, 50, 100, 50, 250, 1, rnn_type=1, use_relu=1, rnn_init=1, param_init=0.1, reducer=2)
    o64 = Oracle(make_cfg(Vt=6, Ve=400, Vr=9, dt=50, de=100, dr=50, H=250, L=1, rnn_type=1, use_relu=1), np.float64)
    theta = eng.get_flat_params().astype(np.float64)
    idx, labels = synth.make_paths(64, 2, 6, Ve=400, seed=71)   # batchSize=128 paths-ish
    b = eng.batch(idx, labels)
    out = eng.forward(b, 1, want=("probs", "path_scores"))
    ps, _, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5
    np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=SCORE_RTOL)
    th, st = theta.copy(), o64.new_state()
    opt, oopt = _ffi.make_opt(method=1, lr=1e-3), make_opt(method=1, lr=1e-3)
    for s_ in range(6):
        ol, _ = o64.train_step(th, st, oopt, idx, labels)
        gl = eng.train_step(b, opt)
        assert abs(gl - ol) < 2e-4 * max(1, abs(ol)), (s_, gl, ol)
    assert float(np.max(np.abs(eng.get_flat_params() - th))) < 2e-4


# ---- identical-prefix plan (batch_index.hip prefix_plan, lstm_fused_prefix.hip) ---------------------------------
def _plan_executed_steps(idx, nT=1, kcap=8):
    """numpy restatement of the plan: reference step = step 0 of the first path whose steps 0 and 1 carry the same ids;
    k_n = leading steps equal to it (<= min(T-2, kcap)); paths sorted by k (stable); a 64-path tile skips its smallest k."""
    B, P, T, F = idx.shape
    rows = idx.reshape(B * P, T, F)[:, :, F - nT - 2:]
    same = np.all(rows[:, 0] == rows[:, 1], axis=1)
    if not same.any():
        return B * P * T
    ref = rows[np.argmax(same), 0]
    eq = np.all(rows == ref[None, None, :], axis=2)
    k = np.where(eq.all(axis=1), T, np.argmin(eq, axis=1))
    k = np.minimum(k, min(T - 2, kcap))
    ks = np.sort(k, kind="stable")
    skipped = sum(int(ks[i]) * len(ks[i:i + 64]) for i in range(0, len(ks), 64))
    return B * P * T - skipped


def _pad_left(idx, pads, Vt=6, Ve=300, Vr=9):
    """overwrite the first pads[n] steps of path n with the pad tuple (synth.make_paths conventions)"""
    B, P, T, F = idx.shape
    flat = idx.reshape(B * P, T, F)
    for n, k in enumerate(pads):
        flat[n, :k, :] = (Vt - 1, Ve, Vr - 1)
    return idx


@pytest.mark.parametrize("case", ["all_padded", "no_pads", "mixed_ragged", "T3", "deep_T8", "capped_T12", "L1"])
def test_identical_prefix_plan_matches_oracle(case):
    """Paths that share leading (pad) steps are started behind them from the state computed once per batch; forward,
    every gradient (including the pad rows of the three tables, fed by the prefix backward) and one Adam step must match
    the oracle, which runs every step of every path."""
    rng = np.random.default_rng(7)
    L, T, pairs, P = 2, 6, 70, 3
    if case == "T3":
        T = 3
    elif case == "deep_T8":
        T = 8
    elif case == "capped_T12":
        T = 12
    elif case == "L1":
        L = 1
    if case == "mixed_ragged":
        pairs, P = 67, 3   # 201 paths: ragged last tile, tiles that mix prefix lengths
    eng, o64, theta = mk(L=L, impl="auto")
    idx, labels = synth.make_paths(pairs, P, T, Ve=300, seed=90 + T, real_len=T)
    N = pairs * P
    if case == "all_padded":
        pads = rng.integers(1, 3, size=N)
    elif case == "no_pads":
        pads = np.zeros(N, dtype=int)
    elif case == "T3":
        pads = rng.integers(0, 3, size=N)   # two pads of three steps: one is skipped (two steps always run)
    elif case == "deep_T8":
        pads = rng.integers(0, 7, size=N)
    elif case == "capped_T12":
        pads = rng.integers(0, 11, size=N)   # up to 10 pad steps: only 8 are skipped, the rest run as ordinary steps
    else:
        pads = rng.integers(0, 4, size=N)
    idx = _pad_left(idx, pads)
    b = eng.batch(idx, labels)
    assert b.executed_steps == _plan_executed_steps(idx)
    if case == "no_pads":
        assert b.executed_steps == N * T
    else:
        assert b.executed_steps < N * T
    assert b.n_uniq == len(np.unique(idx[..., 1]))
    out = eng.forward(b, 1, want=("probs", "path_scores"))
    ps, _, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5
    np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=SCORE_RTOL)
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < GRAD_RTOL, (nm, case)
    # the pad rows themselves (their gradient comes from the prefix backward alone when every occurrence is skipped)
    ge = eng.get_grad("entity_emb").astype(np.float64)
    off, shp = eng.layout()["entity_emb"]
    oge = og[off:off + int(np.prod(shp))].reshape(shp)
    assert np.max(np.abs(ge[299] - oge[299])) <= GRAD_RTOL * max(1e-30, np.max(np.abs(oge)))
    # a few optimiser steps on the same batch (prefix table recomputed after every update)
    _train_compare(eng, o64, theta, [(idx, labels)], dict(method=1, lr=1e-2), 4, 2e-4)


def test_identical_prefix_plan_full_size_equals_no_plan(monkeypatch):
    """BASELINE size (65 536 paths, T = 6, L = 2): scores and gradients with the plan equal those of the same library
    with every step executed (KPRN_DBG=64), to fp32 re-association."""
    import subprocess, sys, json, textwrap
    code = textwrap.dedent("""
        import sys, json, numpy as np
        sys.path.insert(0, %r)
        from kprn_amd import _ffi, synth
        eng = _ffi.Engine(6, 200000, 9, 16, 32, 16, 64, 2)
        idx, labels = synth.make_paths(16384, 4, 6, Ve=200000, seed=3)
        b = eng.batch(idx, labels)
        out = eng.forward(b, 1, want=("probs",))
        loss = eng.backward(b, 1)
        g = eng.get_flat_grads()
        lay = eng.layout()
        res = {"loss": float(loss), "steps": b.executed_steps, "probs": out["probs"][:2000].astype(float).tolist()}
        for nm in lay:
            off, shp = lay[nm]
            v = g[off:off + int(np.prod(shp))].astype(np.float64)
            res[nm] = [float(np.abs(v).max()), float(v.sum()), float((v * np.cos(np.arange(v.size) * 0.37)).sum()), float(np.abs(v).sum())]
        print(json.dumps(res))
    """) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for dbg in ("0", "64"):
        env = dict(os.environ, KPRN_DBG=dbg)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        res[dbg] = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["64"]["steps"] == 65536 * 6 and res["0"]["steps"] < 0.8 * 65536 * 6
    assert abs(res["0"]["loss"] - res["64"]["loss"]) < 1e-6 * max(1.0, abs(res["64"]["loss"]))
    np.testing.assert_allclose(res["0"]["probs"], res["64"]["probs"], rtol=1e-5)
    for nm, ref in res["64"].items():
        if nm in ("loss", "steps", "probs"):
            continue
        got = res["0"][nm]
        scale = max(1e-30, ref[0])
        assert abs(got[0] - ref[0]) < 1e-4 * scale, nm
        # plain and cosine-weighted sums of the tensor: fp32 re-association moves them by ~1e-6 of the sum of magnitudes
        tol = 5e-5 * ref[3] + 1e-12
        assert abs(got[1] - ref[1]) < tol and abs(got[2] - ref[2]) < tol, (nm, got, ref)


@pytest.mark.parametrize("compute_dtype", [0, 2, 3])
def test_scoring_pass_on_the_side_stream_changes_nothing(compute_dtype):
    """score_overlap: kprn_forward_batch_async on a second stream, sharing the chip with the train step enqueued behind it.
    Scores are those of the parameters BEFORE the step's update, and the training trajectory is untouched."""
    outs = []
    for overlap in ("0", "1"):
        eng, o64, theta = mk(L=2, impl="auto", compute_dtype=compute_dtype)
        eng.set_option("score_overlap", overlap)
        batches = [synth.make_paths(300, P, 6, Ve=300, seed=40 + P) for P in (2, 3)]
        gb = [eng.batch(i, l) for i, l in batches]
        opt = _ffi.make_opt(method=1, lr=1e-2)
        probs, losses = [], []
        eng.train_step(gb[0], opt, 1)   # (the first trainBatch zeroes the pad rows, MyOptimizer.lua:181; scoring never does)
        for s_ in range(6):
            b = gb[s_ % 2]
            eng.forward_async(b, 1)
            losses.append(eng.train_step(b, opt, 1))
            probs.append(eng.read_probs(b.B))      # the pass enqueued before the step: pre-update parameters
        outs.append((np.concatenate(probs), np.array(losses), eng.get_flat_params()))
    # (without overlap read_probs returns what the step's own training forward left: same parameters, the SAVE variant of the kernel)
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=1e-6)
    assert np.max(np.abs(outs[0][2] - outs[1][2])) < 1e-6   # (atomics in the embedding backward: run-to-run noise only)


def test_identical_prefix_plan_random_shapes_agree_with_every_step_executed():
    """Random shapes and pad patterns (all paths identical, everything padded, nothing padded, ragged tiles, deep prefixes):
    the same engine, the same data, once with the plan and once with every step executed."""
    rng = np.random.default_rng(123)
    for trial in range(14):
        L = int(rng.integers(1, 3))
        T = int(rng.integers(2, 13))
        pairs, P = int(rng.integers(1, 160)), int(rng.integers(1, 6))
        eng, o64, theta = mk(L=L, impl="auto", seed=10 + trial)
        idx, labels = synth.make_paths(pairs, P, T, Ve=300, seed=500 + trial, real_len=T)
        N = pairs * P
        mode = trial % 5
        if mode == 0:
            pads = rng.integers(0, T, size=N)           # anything, including fully padded paths
        elif mode == 1:
            pads = np.full(N, min(T - 1, 3))            # one class only
        elif mode == 2:
            pads = np.zeros(N, dtype=int)
            if N > 3:
                pads[N // 2:N // 2 + 2] = min(T, 2)     # two padded paths in the middle of a tile
        elif mode == 3:
            pads = rng.integers(0, min(T, 3), size=N)
            idx[:] = idx[:1, :1]                        # every path identical
        else:
            pads = np.where(rng.random(N) < 0.73, min(T - 1, 2), 0)
        idx = _pad_left(idx, pads)
        res = []
        for plan in ("1", "0"):
            eng.set_option("prefix_plan", plan)
            b = eng.batch(idx, labels)
            out = eng.forward(b, 1, want=("probs", "path_scores"))
            loss = eng.backward(b, 1)
            res.append((b.executed_steps, out["path_scores"].astype(np.float64), float(loss), eng.get_flat_grads().astype(np.float64)))
        assert res[1][0] == N * T and res[0][0] <= N * T
        assert res[0][0] == _plan_executed_steps(idx), (trial, T, N)
        assert rel_inf(res[0][1], res[1][1]) < 1e-5, (trial, "scores")
        assert abs(res[0][2] - res[1][2]) < 1e-5 * max(1.0, abs(res[1][2])), (trial, "loss")
        for nm, (off, shp) in eng.layout().items():
            n = int(np.prod(shp))
            ref = res[1][3][off:off + n]
            assert np.max(np.abs(res[0][3][off:off + n] - ref)) < 2e-5 * max(1e-30, np.max(np.abs(ref))), (trial, nm, T, L, N)
        if trial < 3:   # and against the oracle
            ol, og, _ = o64.forward_backward(theta, idx, labels)
            assert rel_inf(res[0][3], og) < GRAD_RTOL


# ---- forward on the bf16 matrix cores (lstm_fused_fwd_mc.hip) ---------------------------------------------------------
@pytest.mark.parametrize("compute_dtype", [2, 3])
@pytest.mark.parametrize("L,P,T,pairs", [(1, 1, 6, 70), (2, 3, 6, 45), (2, 7, 3, 37), (1, 28, 4, 11), (2, 2, 12, 130)])
def test_f32x6_forward_backward_hold_the_fp32_bars(L, P, T, pairs, compute_dtype):
    """compute_dtype = 2: fp32 operands split exactly into three bf16 pieces, six partial products per term on the matrix
    cores, fp32 accumulation (3: two fp16 pieces of the pre-scaled operands, three products).  Same tolerances as the fp32-MFMA
    path: scores 2e-5, gradients 2e-4 of the tensor's largest."""
    eng, o64, theta = mk(L=L, impl="auto", compute_dtype=compute_dtype)
    idx, labels = synth.make_paths(pairs, P, T, Ve=300, seed=300 + T + P)
    b = eng.batch(idx, labels)
    out = eng.forward(b, 1, want=("probs", "all_probs", "path_scores"))
    ps, pooled, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5
    np.testing.assert_allclose(out["all_probs"], probs, rtol=SCORE_RTOL)
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < GRAD_RTOL, (nm, L, P, T)
    _train_compare(eng, o64, theta, [(idx, labels)], dict(method=1, lr=1e-2), 4, 2e-4)


def test_f32x6_is_as_close_to_the_f64_oracle_as_fp32_mfma():
    """the split products are exact and the dropped ones are below 2^-24: the f32x6 scores must not be further from the
    float64 oracle than the fp32-MFMA scores are (both measured on the same parameters and paths)."""
    errs = {}
    for dt in (0, 2, 3):
        eng, o64, theta = mk(L=2, impl="auto", compute_dtype=dt)
        idx, _ = synth.make_paths(400, 3, 6, Ve=300, seed=77)
        out = eng.forward(eng.batch(idx), 1, want=("path_scores",))
        ps, _, _ = o64.forward(theta, idx)
        errs[dt] = rel_inf(out["path_scores"], ps)
    print(errs)
    assert errs[2] < 2e-6 and errs[2] < 3 * errs[0] + 1e-7, errs
    assert errs[3] < 2e-6 and errs[3] < 3 * errs[0] + 1e-7, errs


def test_bf16_fused_scoring_is_tolerance_gated():
    """compute_dtype = 1 at D = H = 64: scoring on the fused matrix-core forward (operands rounded to bf16, fp32 accumulation)"""
    eng, o64, theta = mk(L=2, impl="auto", compute_dtype=1)
    idx, _ = synth.make_paths(200, 3, 6, Ve=300, seed=78)
    out = eng.forward(eng.batch(idx), 1, want=("probs", "path_scores"))
    ps, _, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 3e-2      # bf16 has 8 mantissa bits; 12 layer-steps deep
    np.testing.assert_allclose(out["probs"], probs[:, 0], atol=2e-2)
    # and it agrees with the generic bf16 pipeline (same rounding of the operands) much more closely than with the oracle
    eng.set_option("impl", "generic")
    out_g = eng.forward(eng.batch(idx), 1, want=("path_scores",))
    assert rel_inf(out["path_scores"], out_g["path_scores"].astype(np.float64)) < 1e-2


@pytest.mark.parametrize("dt,de,rnn_type", [(0, 24, 0), (16, 0, 0), (0, 0, 0), (0, 24, 1), (16, 0, 2)])
@pytest.mark.parametrize("pairs", [33, 150])
def test_embedding_ablations_match_oracle(dt, de, rnn_type, pairs):
    """OneModel.lua:207-219 / FeatureEmbedding.lua:26-34,83-110: -includeEntityTypes 0 (x_t = [entities | relations]), -includeEntity 0
    ([types | relations]), both 0 ([relations]); a left-out table is a table of width 0.  Forward, backward and Adam steps against the
    oracle at a small and at a tiled-GEMM size; the id columns of the absent tables are ignored, as the reference ignores them."""
    Ve, dr, H = 300, 8, 40
    eng = _ffi.Engine(6, Ve, 9, dt, de, dr, H, 1, rnn_type=rnn_type, param_init=0.2)
    o64 = Oracle(make_cfg(Vt=6, Ve=Ve, Vr=9, dt=dt, de=de, dr=dr, H=H, L=1, rnn_type=rnn_type), np.float64)
    assert eng.n_params == o64.n and eng.D == dt + de + dr
    theta = o64.init_params(11, 0.2).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    assert np.array_equal(eng.get_flat_params(), theta.astype(np.float32))
    idx, labels = synth.make_paths(pairs, 3, 5, Ve=Ve, seed=13)
    b = eng.batch(idx, labels)
    out = eng.forward(b, 1, want=("probs", "path_scores"))
    ps, _, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5
    np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=SCORE_RTOL)
    x = eng.embed(idx)
    assert x.shape[-1] == dt + de + dr and np.array_equal(x, Oracle(o64.cfg, np.float32).embed(theta.astype(np.float32), idx))
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        if n:
            assert rel_inf(g[off:off + n], og[off:off + n]) < GRAD_RTOL, nm
    _train_compare(eng, o64, theta, [(idx, labels)], dict(method=1, lr=1e-2), 5, 2e-4)
