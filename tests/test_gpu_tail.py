"""-m gpu: the serial tail of a training step: launches merged into one must leave the SAME bits as the launches they replace.

optim.adam sweeps the whole flat vector (MyOptimizer.lua:218); here the touched entity rows (lazy-exact replay) and the dense arena
are updated by one launch (option "adam_merged", default on) instead of two -- the per-element arithmetic is the same device function, so parameters
and both moments must be bit-identical to the two-launch form, at the bench's shape and at the reference's minibatch size (config.sh:38)."""
import numpy as np
import pytest

from kprn_amd import _ffi, synth

pytestmark = pytest.mark.gpu
T = 6


def _distinct_paths(seed, lo):
    """one pair, one path, six DISTINCT entity rows (no gradient atomic ever sees two addends: the step's gradients are reproducible bit for bit)"""
    idx, labels = synth.make_paths(1, 1, T, Ve=2000, seed=seed)
    idx = idx.copy()
    idx[0, 0, :, 1] = np.arange(lo, lo + T)
    return idx, labels


@pytest.mark.parametrize("de,H", [(32, 64), (64, 96), (128, 160)])
def test_merged_adam_launch_is_bit_identical_to_two_launches(de, H):
    """rows of 32 / 64 / 128 floats (the three vector widths of the row kernel); D = H = 64 runs the fused kernels, the others the generic pipeline"""
    res = []
    for merged in ("1", "0"):
        eng = _ffi.Engine(6, 2000, 9, 16, de, 16, H, 2)
        eng.set_option("adam_merged", merged)
        rng = np.random.default_rng(1)
        eng.set_flat_params((rng.random(eng.n_params) * 0.2 - 0.1).astype(np.float32))
        opt = _ffi.make_opt(method=1, lr=5e-3)
        b0, b1 = eng.batch(*_distinct_paths(1, 1)), eng.batch(*_distinct_paths(2, 101))
        losses = [eng.train_step(b0, opt)]
        losses += [eng.train_step(b1, opt) for _ in range(9)]     # b0's rows coast for nine steps: the lazy replay runs when they come back
        eng.profile(True)
        losses.append(eng.train_step(b0, opt))
        fam = eng.profile_get()
        assert ("adam_step" in fam) == (merged == "1") and ("adam_dense" in fam) == (merged == "0"), sorted(fam)
        res.append((eng.get_flat_params(), eng.get_flat_opt_state(0), eng.get_flat_opt_state(1), losses))
        eng.close()
    for a, b in zip(res[0][:3], res[1][:3]):
        assert np.array_equal(a, b)
    assert res[0][3] == res[1][3]


def test_merged_adam_launch_at_the_bench_shape_agrees_to_rounding():
    """65 536-path batches: hub entities' gradients are summed with atomics in any order (run to run), so parameters are compared to rounding"""
    batches = [synth.make_paths(8192, 4, T, Ve=50000, seed=70 + k) for k in range(2)]
    res = []
    for merged in ("1", "0"):
        eng = _ffi.Engine(6, 50000, 9, 16, 32, 16, 64, 2)
        eng.set_option("adam_merged", merged)
        rng = np.random.default_rng(1)
        eng.set_flat_params((rng.random(eng.n_params) * 0.2 - 0.1).astype(np.float32))
        opt = _ffi.make_opt(method=1, lr=1e-3)
        bs = [eng.batch(i, l) for i, l in batches]
        for k in range(5):
            eng.train_step(bs[k % 2], opt)
        res.append(eng.get_flat_params().astype(np.float64))
        eng.close()
    assert float(np.max(np.abs(res[0] - res[1]))) < 2e-6


@pytest.mark.parametrize("pairs,P,small,plan", [(4000, 4, "0", True), (19200, 1, "0", False), (60, 4, "1", True)])
def test_small_table_gradients_in_the_bptt_launch_equal_the_passenger_job(pairs, P, small, plan):
    """nn.LookupTable backward of the type / relation tables (net/FeatureEmbedding.lua:86,112-121): summed in LDS tables inside the bottom layer's BPTT launch
    (option "fused_small_tables", default) or by the entity-gradient launch's passenger job from a second reading of dx -- the same sums in another order.
    64-path tiles with more tiles than workgroups (hand-over pieces included), and the 16-row tiles of small batches; every other gradient is untouched."""
    idx, labels = synth.make_paths(pairs, P, T, Ve=30000, seed=11 + P)
    grads = []
    for on in ("1", "0"):
        eng = _ffi.Engine(6, 30000, 9, 16, 32, 16, 64, 2)
        eng.set_option("small_tiles", small)
        eng.set_option("prefix_plan", "1" if plan else "0")
        eng.set_option("fused_small_tables", on)
        rng = np.random.default_rng(3)
        eng.set_flat_params((rng.random(eng.n_params) * 0.2 - 0.1).astype(np.float32))
        b = eng.batch(idx, labels)
        eng.forward(b, 1, want=("probs",))
        eng.profile(True)
        eng.backward(b, 1)
        fam = eng.profile_get()
        grads.append((eng.get_flat_grads().astype(np.float64), eng.layout()))
        eng.close()
    (g1, lay), (g0, _) = grads
    for nm, (off, shp) in lay.items():
        n = int(np.prod(shp))
        a, c = g1[off:off + n], g0[off:off + n]
        scale = max(1e-30, float(np.max(np.abs(c))))
        if "type" in nm.lower() or "rel" in nm.lower():
            assert float(np.max(np.abs(a - c))) / scale < 2e-5, nm
            assert float(np.max(np.abs(c))) > 0, nm
        else:
            assert float(np.max(np.abs(a - c))) / scale < 2e-6, nm   # (atomics of the entity / weight-slab sums re-associate run to run)


def test_catch_up_and_prefix_table_in_one_launch_is_bit_identical_to_two():
    """the stretch between the optimiser step and the next forward: the lazy-exact catch-up of the next batch's entity rows
    (optimizer/MyOptimizer.lua:218 as a replay) and that batch's identical-prefix table (model/OneModel.lua:236 on the shared padded steps) as ONE launch
    (option "catchup_prefix", default) or as two.  Same arithmetic either way: after Adam steps whose gradients are reproducible bit for bit (one path of
    distinct rows: no atomic sees two addends), the scores of a padded batch that shares rows with them -- rows that coast, rows that replay, the pad row --
    and the table's own values are equal bit for bit; the one-launch form runs the table's own kernel less often."""
    big = synth.make_paths(900, 4, T, Ve=2000, seed=301)
    res = []
    for merged in ("1", "0"):
        eng = _ffi.Engine(6, 2000, 9, 16, 32, 16, 64, 2)
        eng.set_option("small_tiles", "0")
        eng.set_option("catchup_prefix", merged)
        rng = np.random.default_rng(2)
        eng.set_flat_params((rng.random(eng.n_params) * 0.2 - 0.1).astype(np.float32))
        opt = _ffi.make_opt(method=1, lr=5e-3)
        b0, b1, bb = eng.batch(*_distinct_paths(1, 1)), eng.batch(*_distinct_paths(2, 101)), eng.batch(*big)
        scores = []
        eng.profile(True)
        for k in range(6):
            eng.train_step(b0 if k % 3 == 0 else b1, opt)
            scores.append(eng.forward(bb, 1, want=("path_scores",))["path_scores"].copy())   # catch-up of bb's rows (+ its prefix table: parameters moved)
        fam = eng.profile_get()
        res.append((scores, eng.get_flat_params(), fam.get("prefix_fwd", (0.0, 0))[1], fam.get("adam_rows_catchup", (0.0, 0))[1]))
        eng.close()
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(res[0][1], res[1][1])
    assert res[1][2] >= 6 and res[0][2] < res[1][2], (res[0][2:], res[1][2:])


@pytest.mark.parametrize("pairs,P", [(3000, 4), (60, 4), (20000, 1)])
def test_scoring_pass_in_the_training_forwards_launch_is_bit_identical(pairs, P):
    """option "score_dual" (with "score_overlap"): a pass queued by kprn_forward_batch_async waits for the training forward that follows and runs as a second
    branch of its kernel (eval/test_from_checkpoint.lua:109's forward and MyOptimizer.lua:177-221's, one launch).  64-path tiles, the 16-row tiles of small
    batches, more tiles than workgroups; the pass's probabilities bit for bit those of the side-stream pass, the training step's loss and parameters equal
    (the same kernels' arithmetic; entity-gradient atomics re-associate run to run).  A pass nobody trains behind is run the usual way when its result is read."""
    idx, labels = synth.make_paths(pairs, P, T, Ve=30000, seed=21 + P)
    res = []
    for dual in ("1", "0"):
        eng = _ffi.Engine(6, 30000, 9, 16, 32, 16, 64, 2)
        eng.set_option("score_overlap", "1")
        eng.set_option("score_dual", dual)
        rng = np.random.default_rng(4)
        eng.set_flat_params((rng.random(eng.n_params) * 0.2 - 0.1).astype(np.float32))
        opt = _ffi.make_opt(method=1, lr=1e-3)
        b = eng.batch(idx, labels)
        probs, losses = [], []
        eng.profile(True)
        for k in range(3):
            eng.forward_async(b, 1)
            losses.append(eng.train_step(b, opt))
            probs.append(eng.read_probs(b.B).copy())
        fam = eng.profile_get()
        assert ("lstm_fused_fwd_dual" in fam) == (dual == "1"), sorted(fam)
        eng.forward_async(b, 1)                      # ... and a pass with no training forward behind it
        probs.append(eng.read_probs(b.B).copy())
        res.append((probs, losses, eng.get_flat_params().astype(np.float64)))
        eng.close()
    # the default ("2"): the one launch below the 16-row-tile threshold only
    eng = _ffi.Engine(6, 30000, 9, 16, 32, 16, 64, 2)
    eng.set_option("score_overlap", "1")
    b = eng.batch(idx, labels)
    opt = _ffi.make_opt(method=1, lr=1e-3)
    eng.profile(True)
    for _ in range(2):   # (a new engine's first step zeroes the pad rows first -- a parameter change: the queued pass runs ahead of it, the usual way)
        eng.forward_async(b, 1)
        eng.train_step(b, opt)
    assert ("lstm_fused_fwd_dual" in eng.profile_get()) == (pairs * P < 8192)
    eng.close()
    assert np.array_equal(res[0][0][0], res[1][0][0])          # same parameters, same arithmetic: the first pass bit for bit
    for a, c in zip(res[0][0][1:], res[1][0][1:]):
        np.testing.assert_allclose(a, c, rtol=2e-5)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-5)
    assert float(np.max(np.abs(res[0][2] - res[1][2]))) < 2e-6


@pytest.mark.parametrize("pairs,P", [(60, 4), (500, 4), (8, 2)])
def test_both_layers_bptt_in_one_launch_equals_one_launch_per_layer(pairs, P):
    """option "bwd_pipe" (default): at batches of 16-row tiles whose two layers' workgroups fit the chip together, BPTT through both FastLSTM layers
    (model/OneModel.lua:236,268-274 backward) is ONE launch -- the bottom layer's workgroup of a tile waits, step by step, for the dx the top layer's
    workgroup of the same tile has just published -- instead of one launch per layer.  The same arithmetic in the same order per tile: every gradient
    equal to what the atomics of the embedding backward re-associate, and the pipeline really is one launch."""
    idx, labels = synth.make_paths(pairs, P, T, Ve=30000, seed=33 + pairs)
    res = []
    for pipe in ("1", "0"):
        eng = _ffi.Engine(6, 30000, 9, 16, 32, 16, 64, 2)
        eng.set_option("small_tiles", "1")
        eng.set_option("bwd_pipe", pipe)
        rng = np.random.default_rng(5)
        eng.set_flat_params((rng.random(eng.n_params) * 0.2 - 0.1).astype(np.float32))
        b = eng.batch(idx, labels)
        eng.forward(b, 1, want=("probs",))
        eng.profile(True)
        losses = [eng.backward(b, 1) for _ in range(3)]          # (three times: the epoch of the flags moves on)
        fam = eng.profile_get()
        res.append((eng.get_flat_grads().astype(np.float64), eng.layout(), losses, fam["lstm_fused_bwd"]))
        eng.close()
    (g1, lay, l1, f1), (g0, _, l0, f0) = res
    assert l1 == l0
    for nm, (off, shp) in lay.items():
        n = int(np.prod(shp))
        a, c = g1[off:off + n], g0[off:off + n]
        assert float(np.max(np.abs(a - c))) <= 2e-6 * max(1e-30, float(np.max(np.abs(c)))), nm
    if pairs * P > 16:     # (one tile: a single workgroup per layer, launched the usual way)
        assert f1[1] < f0[1], (f1, f0)   # launches recorded for the family: 3 x 1 against 3 x 2
