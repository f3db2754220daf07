"""-m gpu: the streaming batch feed (kprn_batch_feed_async) -- the engine's counterpart of BatcherFileList's GPU double buffer
(release/songPathRnn/model/batcher/BatcherFileList.lua:53-96: tensors preallocated once, every minibatch :copy()'d into them).
A slot refilled on the feed stream must behave exactly like a batch made by kprn_batch_create."""
import numpy as np
import pytest

from kprn_amd import _ffi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _sixty_four_path_tiles(monkeypatch):
    """This module pins the 64-path tiles and their identical-prefix plan on SMALL shapes (fast, every edge case); left alone the engine runs batches
    of <= 8 192 paths on tiles of one 16-row m-tile without a plan -- tests/test_gpu_small_tiles.py covers that mode."""
    monkeypatch.setenv("KPRN_SMALL_TILES", "0")

SHAPE = (6, 5000, 9, 16, 32, 16, 64, 2)


def _pinned(eng, idx, labels):
    hi = eng.host_array(idx.shape, np.int32)
    hi[...] = idx
    hl = eng.host_array(labels.shape, np.float32)
    hl[...] = labels
    return hi, hl


def _batches(n, seed=0):
    out = []
    for i in range(n):
        P = [1, 3, 2, 5, 4, 8][i % 6]
        pairs = [700, 300, 900, 150, 1100, 90][i % 6]   # the slots must grow (realloc) and shrink (reuse)
        out.append(synth.make_paths(pairs, P, 6, Ve=5000, seed=seed + 11 * i))
    return out


@pytest.mark.parametrize("build", ["host", "device"])
@pytest.mark.parametrize("impl", ["auto", "generic"])
def test_fed_slot_equals_created_batch(impl, build):
    eng = _ffi.Engine(*SHAPE)
    eng.set_option("impl", impl)
    eng.set_option("feed_build", build)
    idx, labels = synth.make_paths(500, 3, 6, Ve=5000, seed=3)
    ref = eng.batch(idx, labels)
    hi, hl = _pinned(eng, idx, labels)
    slot = eng.feed(hi, hl)
    assert slot.n_uniq == ref.n_uniq and slot.executed_steps == ref.executed_steps
    a = eng.forward(ref, 1, want=("probs", "path_scores"))
    b = eng.forward(slot, 1, want=("probs", "path_scores"))
    assert np.array_equal(a["probs"], b["probs"]) and np.array_equal(a["path_scores"], b["path_scores"])
    la = eng.backward(ref, 1)
    ga = eng.get_flat_grads()
    lb = eng.backward(slot, 1)
    gb = eng.get_flat_grads()
    assert abs(la - lb) < 1e-6 * max(1.0, abs(la))
    assert np.max(np.abs(ga - gb)) <= 2e-6 * np.max(np.abs(ga))   # (a few fp32 atomics in the embedding backward)
    eng.close()


@pytest.mark.parametrize("build", ["host", "device"])
@pytest.mark.parametrize("score_overlap", ["0", "1"])
def test_slot_ring_trains_like_resident_batches(score_overlap, build):
    """the loop bench.py --batch-feed streaming runs: feed the batches of the next steps, then score + train on batch i"""
    data = _batches(9, seed=100)
    opt = _ffi.make_opt(method=1, lr=1e-2)
    res = []
    for streaming in (False, True):
        eng = _ffi.Engine(*SHAPE, seed=7)
        eng.set_option("score_overlap", score_overlap)
        eng.set_option("feed_build", build)
        probs = []
        if streaming:
            host = [_pinned(eng, i, l) for i, l in data]
            NS = 3
            slots = [None] * NS
            fed = -1
            for s in range(len(data)):
                while fed < min(s + NS - 1, len(data) - 1):
                    fed += 1
                    slots[fed % NS] = eng.feed(host[fed][0], host[fed][1], slot=slots[fed % NS])
                b = slots[s % NS]
                eng.forward_async(b, 1)
                eng.train_step(b, opt, want_loss=False)
                probs.append(eng.read_probs(b.B))
        else:
            for i, l in data:
                b = eng.batch(i, l)
                eng.forward_async(b, 1)
                eng.train_step(b, opt, want_loss=False)
                probs.append(eng.read_probs(b.B))
        res.append((eng.get_flat_params(), probs, eng.read_loss()))
        eng.close()
    (pa, qa, la), (pb, qb, lb) = res
    assert abs(la - lb) < 1e-5 * max(1.0, abs(la))
    assert np.max(np.abs(pa - pb)) < 2e-6
    for x, y in zip(qa, qb):
        np.testing.assert_allclose(x, y, rtol=1e-5)


@pytest.mark.parametrize("build", ["host", "device"])
def test_bad_id_surfaces_at_first_use_and_the_slot_recovers(build):
    eng = _ffi.Engine(*SHAPE)
    eng.set_option("feed_build", build)
    idx, labels = synth.make_paths(64, 2, 6, Ve=5000, seed=5)
    bad = idx.copy()
    bad[3, 1, 2, 1] = 5001   # entity id outside 1..Ve
    slot = eng.feed(bad, labels)
    with pytest.raises(_ffi.KprnError) as e:
        eng.forward(slot, 1)
    assert e.value.code == _ffi.E_INDEX
    with pytest.raises(_ffi.KprnError):   # and it stays an error until the slot is refilled
        eng.train_step(slot, _ffi.make_opt())
    eng.feed(idx, labels, slot=slot)
    ref = eng.batch(idx, labels)
    assert np.array_equal(eng.forward(slot, 1)["probs"], eng.forward(ref, 1)["probs"])
    # argument errors are reported by the feed call itself
    with pytest.raises(_ffi.KprnError) as e:
        eng.feed(idx[..., :2], labels)
    assert e.value.code == _ffi.E_ARG
    eng.close()


@pytest.mark.parametrize("build", ["host", "device"])
def test_prefix_plan_toggle_between_refills(build):
    """a slot allocated with a plan and refilled with plans switched off (and back) must not read stale plan buffers"""
    eng = _ffi.Engine(*SHAPE)
    eng.set_option("feed_build", build)
    idx, labels = synth.make_paths(400, 4, 6, Ve=5000, seed=9)
    slot = eng.feed(idx, labels)
    p1 = eng.forward(slot, 1)["probs"]
    n1 = slot.executed_steps
    eng.set_option("prefix_plan", "0")
    eng.feed(idx, labels, slot=slot)
    p0 = eng.forward(slot, 1)["probs"]
    assert slot.executed_steps == 400 * 4 * 6 and n1 < slot.executed_steps
    np.testing.assert_allclose(p1, p0, rtol=1e-5)
    eng.set_option("prefix_plan", "1")
    eng.feed(idx, labels, slot=slot)
    assert slot.executed_steps == n1
    assert np.array_equal(eng.forward(slot, 1)["probs"], p1)
    eng.close()


def test_label_less_slot_fed_before_training_scores_with_current_rows():
    """a scoring-only batch (no labels) fed while no lazy row update is pending carries no occurrence index; if a training step
    then leaves entity rows behind, scoring that slot must still see every row up to date (the whole table is flushed instead)"""
    eng = _ffi.Engine(*SHAPE)
    ia, la = synth.make_paths(300, 3, 6, Ve=5000, seed=21)
    ib, _ = synth.make_paths(200, 2, 6, Ve=5000, seed=22)
    slot = eng.feed(ib)                      # no labels, nothing pending -> plan only
    assert slot.n_uniq == 0
    ta = eng.batch(ia, la)
    opt = _ffi.make_opt(method=1, lr=1e-2)
    for _ in range(3):
        eng.train_step(ta, opt)              # lazy-exact Adam: rows of B not in A fall behind
    got = eng.forward(slot, 1, want=("probs", "path_scores"))
    ref = eng.forward(eng.batch(ib), 1, want=("probs", "path_scores"))
    assert np.array_equal(got["probs"], ref["probs"]) and np.array_equal(got["path_scores"], ref["path_scores"])
    # fed while rows are pending: the slot carries its row list
    eng.train_step(ta, opt)
    slot2 = eng.feed(ib)
    assert slot2.n_uniq == ref_uniq(ib)
    assert np.array_equal(eng.forward(slot2, 1)["probs"], eng.forward(eng.batch(ib), 1)["probs"])
    with pytest.raises(_ffi.KprnError):      # no labels -> no training on it
        eng.train_step(slot, opt)
    eng.close()


@pytest.mark.parametrize("build", ["host", "device"])
def test_rows_feed_gathers_a_shuffled_minibatch_inside_the_engine(build):
    """kprn_batch_feed_rows_async: pair i = row rows[i] of the file's arrays (Batcher.lua:35-41's shuffle without moving the file)"""
    eng = _ffi.Engine(*SHAPE)
    eng.set_option("feed_build", build)
    data, labels = synth.make_paths(900, 3, 6, Ve=5000, seed=31)
    rng = np.random.default_rng(2)
    slot = None
    for n in (256, 100, 611):                               # the slot grows and shrinks between refills
        rows = rng.permutation(900)[:n].astype(np.int64)
        slot = eng.feed_rows(data, labels, rows, slot=slot)
        ref = eng.batch(data[rows], labels[rows])
        assert slot.B == n and slot.n_uniq == ref.n_uniq
        a = eng.forward(ref, 1, want=("probs", "path_scores"))
        b = eng.forward(slot, 1, want=("probs", "path_scores"))
        assert np.array_equal(a["probs"], b["probs"]) and np.array_equal(a["path_scores"], b["path_scores"])
        la = eng.backward(ref, 1)
        lb = eng.backward(slot, 1)
        assert abs(la - lb) < 1e-6 * max(1.0, abs(la))
    with pytest.raises(_ffi.KprnError) as e:                # a row outside the file
        eng.feed_rows(data, labels, np.array([0, 900], np.int64), slot=slot)
    assert e.value.code == _ffi.E_ARG
    eng.close()


def test_loss_accumulator_sums_the_steps_without_a_sync_per_step():
    """loss_accumulate = 1 + kprn_read_loss_sum: MyOptimizer's totalError (MyOptimizer.lua:199) formed on the device"""
    eng = _ffi.Engine(*SHAPE, seed=3)
    ref = _ffi.Engine(*SHAPE, seed=3)
    opt = _ffi.make_opt(method=1, lr=1e-2)
    data = _batches(5, seed=40)
    eng.set_option("loss_accumulate", "1")
    want = 0.0
    for i, l in data:
        eng.train_step(eng.batch(i, l), opt, want_loss=False)
        want += ref.train_step(ref.batch(i, l), opt)
    s, n = eng.loss_sum(reset=True)
    assert n == 5 and abs(s - want) < 1e-5 * abs(want)
    assert eng.loss_sum() == (0.0, 0)
    eng.train_step(eng.batch(*data[0]), opt, want_loss=False)
    s1, n1 = eng.loss_sum(reset=False)
    assert n1 == 1 and abs(s1 - eng.read_loss()) < 1e-6 * max(1.0, abs(s1))
    eng.close(); ref.close()


def test_soak_random_shapes_through_a_slot_ring():
    """scripts/gpu_feed_soak.py: 300 random minibatches (1 .. 1 500 pairs, P in {1,2,3,5,8}; rows feed / plain feed / label-less) through
    a 4-slot ring with training and second-stream scoring interleaved, host- and device-built, against an engine fed resident batches"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "gpu_feed_soak.py"), "150", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "failures: 0" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def ref_uniq(idx):
    return len(np.unique(idx[..., 1]))
