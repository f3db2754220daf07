"""-m gpu: parity of the BENCHMARKED code path at the benchmarked size.

bench.py's workload is 65 536 paths per step (1 024 tiles of 64 paths on 256 persistent workgroups = 4 tiles per workgroup:
id-tile double buffering by tile parity, cell-state reset between tiles, next-tile prefetch, weight-gradient slab flush after
several tiles), T = 6, D = H = 64, L = 2, the KKBox-size entity table (Ve = 2 851 220).  The small-shape parity tests never
put more than one tile on a workgroup, so this file compares exactly that configuration with two independent implementations:
  (a) the float64 CPU oracle (oracle/kprn_oracle.c; OpenMP over pairs: seconds at this size),
  (b) the engine's own generic pipeline (impl=generic: plain GEMM + element-wise kernels, no persistent kernels, no plan),
for compute_dtype 0 (fp32 MFMA), 2 (f32x6) and 3 (f32x3), identical-prefix plan on and off, plus five Adam steps, a ragged last
tile and the scoring kernel restricted to half the CUs (reserve_cus: 8 tiles per workgroup).
Reference semantics: module/MapReduce.lua:24-47, model/OneModel.lua:223-275, optimizer/MyOptimizer.lua:177-221.
"""
import numpy as np
import pytest

from kprn_amd import _ffi, synth
from oracle.oracle import Oracle, make_cfg, make_opt

pytestmark = pytest.mark.gpu

VE = 2851220            # run_scripts/config.sh:25
SHAPE = dict(Vt=6, Ve=VE, Vr=9, dt=16, de=32, dr=16, H=64, L=2)
PAIRS, P, T = 16384, 4, 6   # 65 536 paths = 1 024 tiles
SCORE_RTOL = 1e-4
GRAD_RTOL = 2e-4


def rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(1e-30, np.max(np.abs(b))))


class Case:
    """parameters, paths and the oracle's answers, computed once per module"""

    def __init__(self, pairs, seed):
        self.ocfg = make_cfg(**SHAPE)
        self.o64 = Oracle(self.ocfg, np.float64)
        th32 = self.o64.init_params(seed, 0.1).astype(np.float32)
        self.theta32 = th32
        self.theta = th32.astype(np.float64)
        self.idx, self.labels = synth.make_paths(pairs, P, T, Ve=VE, seed=seed + 1)
        self.rows = np.unique(self.idx[..., 1]) - 1   # entity rows the batch touches (0-based)
        self.ps, self.pooled, self.probs = self.o64.forward(self.theta, self.idx)
        self.loss, self.grad, _ = self.o64.forward_backward(self.theta, self.idx, self.labels)
        self.lay = self.o64.layout()

    def engine(self, compute_dtype=0, impl="auto", plan=True):
        eng = _ffi.Engine(SHAPE["Vt"], VE, SHAPE["Vr"], SHAPE["dt"], SHAPE["de"], SHAPE["dr"], SHAPE["H"], SHAPE["L"], compute_dtype=compute_dtype)
        eng.set_option("impl", impl)
        eng.set_option("prefix_plan", "1" if plan else "0")
        eng.set_flat_params(self.theta32)
        return eng

    def check_forward(self, out):
        # measured (scripts/gpu_parity_probe.py, round 3): max |error| = 8e-7 of the largest score for fp32 MFMA / f32x6 -- fp32 rounding of
        # sums of O(1) terms -- so a score that happens to sit at 1e-6 of the largest carries that ABSOLUTE error too (22 % of itself): the
        # per-element bar is 1e-4 relative above an absolute floor of 3e-6 of the largest score (round 2: 2e-5), probabilities 1e-5 relative
        smax = float(np.max(np.abs(self.ps)))
        assert rel_inf(out["path_scores"], self.ps) < 3e-6
        np.testing.assert_allclose(out["path_scores"], self.ps, rtol=SCORE_RTOL, atol=3e-6 * smax)
        np.testing.assert_allclose(out["pooled"], self.pooled, rtol=SCORE_RTOL, atol=2e-6)
        np.testing.assert_allclose(out["all_probs"], self.probs, rtol=SCORE_RTOL)
        np.testing.assert_allclose(out["probs"], self.probs[:, 0], rtol=1e-5)

    def check_grads(self, eng, loss):
        assert abs(loss - self.loss) < 1e-5 * max(1.0, abs(self.loss)), (loss, self.loss)
        g = eng.get_flat_grads()
        for nm, (off, shp) in self.lay.items():
            n = int(np.prod(shp))
            got, want = g[off:off + n], self.grad[off:off + n]
            if nm == "entity_emb":
                got, want = got.reshape(shp), want.reshape(shp)
                mask = np.ones(shp[0], bool)
                mask[self.rows] = False
                assert not np.any(got[mask]), "gradient on an entity row the batch does not reference"
                got, want = got[self.rows], want[self.rows]
            r = rel_inf(got, want)
            assert r < GRAD_RTOL, (nm, r)


@pytest.fixture(scope="module")
def full():
    return Case(PAIRS, 11)


@pytest.mark.parametrize("plan", [True, False])
@pytest.mark.parametrize("compute_dtype", [0, 2, 3])
def test_benchmarked_size_forward_backward_match_the_f64_oracle(full, compute_dtype, plan):
    eng = full.engine(compute_dtype, "auto", plan)
    b = eng.batch(full.idx, full.labels)
    n_exec = b.executed_steps
    assert (n_exec < 0.8 * PAIRS * P * T) if plan else (n_exec == PAIRS * P * T)
    out = eng.forward(b, 1, want=("probs", "all_probs", "pooled", "path_scores"))
    full.check_forward(out)
    loss = eng.backward(b, 1)
    full.check_grads(eng, loss)
    # the scoring kernel on half the CUs: 8 tiles per persistent workgroup
    eng.set_option("reserve_cus", "128")
    out2 = eng.forward(b, 1, want=("probs", "all_probs", "pooled", "path_scores"))
    full.check_forward(out2)
    eng.close()


def test_benchmarked_size_fused_matches_the_generic_pipeline(full):
    """two independent GPU implementations of the same step: persistent fused kernels with the plan vs GEMM + element-wise"""
    res = {}
    for impl in ("auto", "generic"):
        eng = full.engine(0, impl, impl == "auto")
        b = eng.batch(full.idx, full.labels)
        out = eng.forward(b, 1, want=("probs", "path_scores"))
        # a second pass over the same batch reuses the cached identical-prefix table: bit-identical scores (this caught the table being
        # zeroed by a null-stream hipMemset that ran after the prefix kernel of a new engine's first batch had filled it)
        again = eng.forward(b, 1, want=("path_scores",))["path_scores"]
        assert np.array_equal(again, out["path_scores"]), (impl, rel_inf(again, full.ps), rel_inf(out["path_scores"], full.ps))
        loss = eng.backward(b, 1)
        g = eng.get_flat_grads()
        res[impl] = (out, loss, g)
        if impl == "generic":
            full.check_grads(eng, loss)
        eng.close()
    (oa, la, ga), (og, lg, gg) = res["auto"], res["generic"]
    d_ag = rel_inf(oa["path_scores"], og["path_scores"].astype(np.float64))
    bad = np.nonzero(np.abs(oa["path_scores"] - full.ps).max(axis=1) > 1e-4 * np.abs(full.ps).max())[0]
    assert d_ag < 2e-5, (d_ag, rel_inf(oa["path_scores"], full.ps), rel_inf(og["path_scores"], full.ps), len(bad), bad[:40].tolist(), bad[-10:].tolist())
    np.testing.assert_allclose(oa["probs"], og["probs"], rtol=SCORE_RTOL)
    assert abs(la - lg) < 2e-5 * max(1.0, abs(lg))   # (each is within 1e-5 of the oracle's)
    for nm, (off, shp) in full.lay.items():
        n = int(np.prod(shp))
        assert rel_inf(ga[off:off + n], gg[off:off + n].astype(np.float64)) < GRAD_RTOL, nm


@pytest.mark.parametrize("compute_dtype,plan", [(0, True), (0, False), (2, True)])
def test_benchmarked_size_twenty_adam_steps_match_the_f64_oracle(full, compute_dtype, plan):
    """MyOptimizer:trainBatch x 20 (optim.adam at the reference's default -learningRate 1e-3, OneModel.lua:342; lazy-exact entity rows) on
    two alternating 65 536-path batches.  Measured: max |d theta| 4.4e-7 (f32x6: 5.3e-7), loss 1e-6, probabilities of the trained
    model 1.5e-7 relative; the bars sit an order of magnitude above that (round 2: 5 steps at lr 1e-2, 2e-4 / 2e-4 / 5e-4)."""
    eng = full.engine(compute_dtype, "auto", plan)
    idx2, lab2 = synth.make_paths(PAIRS, P, T, Ve=VE, seed=501)
    batches = [(full.idx, full.labels), (idx2, lab2)]
    gb = [eng.batch(i, l) for i, l in batches]
    th = full.theta.copy()
    st = full.o64.new_state()
    oopt = make_opt(method=1, lr=1e-3)
    gopt = _ffi.make_opt(method=1, lr=1e-3)
    for s in range(20):
        i, l = batches[s & 1]
        ol, _ = full.o64.train_step(th, st, oopt, i, l)
        gl = eng.train_step(gb[s & 1], gopt)
        assert abs(gl - ol) < 1e-5 * max(1.0, abs(ol)), (s, gl, ol)
    got = eng.get_flat_params()
    d = float(np.max(np.abs(got - th)))
    assert d < 5e-6, d
    # and the scores of the trained model
    out = eng.forward(gb[0], 1, want=("probs",))
    _, _, probs = full.o64.forward(th, full.idx)
    np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=1e-5)
    eng.close()


def test_host_buffer_entry_point_equals_the_batch_entry_point():
    """kprn_train_step / kprn_forward (host buffers: what bindings/kprn.lua calls, one upload + batch derivation per call) against
    kprn_batch_create + kprn_train_step_batch / kprn_forward_batch on the same minibatches, ragged sizes included (37 pairs x 3 = 111 paths:
    one full tile + 47 rows; 1 pair).  The FORWARD of both entry points is bit-identical (same kernels, same plan, same tiles).  A training
    step is reproducible to rounding only -- the weight-gradient slab reduce and the entity runs that straddle segments add fp32 partials
    with atomics, whose order differs from launch to launch (measured: 1 ulp differences, 1.5e-8 after four steps) -- so the parameters are
    held to 1e-7 and, given the SAME parameters, the two entry points must again score bit-identically."""
    shape = (6, 5000, 9, 16, 32, 16, 64, 2)
    a, b = _ffi.Engine(*shape, seed=8), _ffi.Engine(*shape, seed=8)
    opt = _ffi.make_opt(method=1, lr=1e-3)
    for k, (pairs, Pk) in enumerate([(37, 3), (128, 2), (1, 5), (300, 1), (37, 3)]):
        idx, labels = synth.make_paths(pairs, Pk, 6, Ve=5000, seed=70 + k)
        bb = b.batch(idx, labels)
        ph, _ = b.forward_host(idx, 1)
        pb = b.forward(bb, 1)["probs"]
        assert np.array_equal(ph, pb), k                       # same engine, same parameters: bit-identical
        pa, _ = a.forward_host(idx, 1)
        np.testing.assert_allclose(pa, pb, rtol=1e-6)          # the other engine's parameters differ by accumulated rounding
        la = a.train_step_host(idx, labels, opt)
        lb = b.train_step(bb, opt)
        assert abs(la - lb) < 1e-6 * max(1.0, abs(lb)), (k, la, lb)
    assert np.max(np.abs(a.get_flat_params() - b.get_flat_params())) < 1e-7
    a.close(); b.close()


@pytest.mark.parametrize("compute_dtype", [0, 2])
def test_ragged_last_tile_many_tiles_per_workgroup(compute_dtype):
    """16 389 pairs x 4 paths = 65 556 paths: 1 024 full tiles + one of 20 rows; scoring with reserve_cus = 128"""
    case = Case(16389, 23)
    eng = case.engine(compute_dtype, "auto", True)
    eng.set_option("reserve_cus", "128")
    b = eng.batch(case.idx, case.labels)
    out = eng.forward(b, 1, want=("probs", "all_probs", "pooled", "path_scores"))
    case.check_forward(out)
    loss = eng.backward(b, 1)
    case.check_grads(eng, loss)
    eng.close()


def test_reference_minibatch_regime_through_the_host_buffer_entry_points():
    """What bindings/kprn.lua calls at the reference's own sizes: kprn_train_step from host buffers with 128 pairs (run_scripts/config.sh:38),
    P per minibatch from the fixture distribution (one bucket file per minibatch: movie_data_format.py:311-314), the loss returned every
    step (MyOptimizer.lua:177-221), then kprn_forward with 512 pairs (test_from_checkpoint.lua:49,109) -- every step's loss, the final
    parameters and the scores against the float64 oracle at the fp32 bars of this file."""
    Ve = 20000
    shape = dict(SHAPE, Ve=Ve)
    o64 = Oracle(make_cfg(**shape), np.float64)
    th = o64.init_params(31, 0.1).astype(np.float32).astype(np.float64)
    eng = _ffi.Engine(6, Ve, 9, 16, 32, 16, 64, 2)
    eng.set_flat_params(th.astype(np.float32))
    rng = np.random.default_rng(99)
    opt, oopt, st = _ffi.make_opt(method=1, lr=1e-3), make_opt(method=1, lr=1e-3), o64.new_state()
    for k in range(12):
        Pk = int(min(rng.geometric(0.57), 28))
        idx, labels = synth.make_paths(128, Pk, T, Ve=Ve, seed=900 + k)
        ol, _ = o64.train_step(th, st, oopt, idx, labels)
        gl = eng.train_step_host(idx, labels, opt)
        assert abs(gl - ol) < 1e-5 * max(1.0, abs(ol)), (k, Pk, gl, ol)
    assert float(np.max(np.abs(eng.get_flat_params() - th))) < 5e-6
    for k, Pk in enumerate((1, 3, 28)):
        idx, _ = synth.make_paths(512, Pk, T, Ve=Ve, seed=950 + k)
        probs, allp = eng.forward_host(idx, 1)
        _, _, want = o64.forward(th, idx)
        np.testing.assert_allclose(probs, want[:, 0], rtol=1e-5)
        np.testing.assert_allclose(allp, want, rtol=1e-5)
        only, none = eng.forward_host(idx, 1, want_all=False)   # the selected class alone (k_pool_sel + the page-locked mirror): the same bits
        assert none is None and np.array_equal(only, probs)
    eng.close()


def test_train_step_returns_the_loss_before_the_step_has_drained_with_identical_results():
    """kprn_train_step / kprn_train_step_batch hand the loss back once the loss stage has run (option "train_step_return" = "loss", the default: the partials
    mirrored into page-locked memory, summed by the host in k_sum_partials' order) while backward and update run on; "drain" waits for the whole step as
    rounds 1-4 did.  Every loss the call returns must be the same BITS as kprn_read_loss right after it (the device-side sum of the same partials: the host
    adds them in the device kernel's order).  The same minibatches through both modes: losses, parameters and Adam moments agree to the last bits (two
    runs of the same steps differ in the order of the fp32 atomics that join a hub entity's segment sums: 1 ulp in a few elements, which a later loss
    inherits).  B = 300 pairs gives 19 partials, B = 7 one."""
    Ve = 20000
    engs = []
    for mode in ("loss", "drain"):
        e = _ffi.Engine(6, Ve, 9, 16, 32, 16, 64, 2, param_init=0.1, seed=5)
        e.set_option("train_step_return", mode)
        e.set_option("inline_upload", "side" if mode == "loss" else "main")   # (the round-4 call: drained, minibatch uploaded in stream order)
        engs.append(e)
    engs[1].set_flat_params(engs[0].get_flat_params())
    opt = _ffi.make_opt(method=1, lr=1e-3)
    rng = np.random.default_rng(4)
    got = [[], []]
    for k in range(14):
        B = (128, 300, 7, 64)[k % 4]
        Pk = int(min(rng.geometric(0.57), 28))
        idx, labels = synth.make_paths(B, Pk, T, Ve=Ve, seed=700 + k)
        for j, e in enumerate(engs):
            if k % 3 == 2:   # the resident-batch entry point
                b = e.batch(idx, labels)
                got[j].append(e.train_step(b, opt, 1, want_loss=True))
                assert e.read_loss() == got[j][-1]
                b.free()   # (kprn_batch_destroy waits for the step that still reads the batch)
            else:
                got[j].append(e.train_step_host(idx, labels, opt))
                assert e.read_loss() == got[j][-1]
    assert np.all(np.isfinite(got[0]))
    np.testing.assert_allclose(got[0], got[1], rtol=2e-6)
    np.testing.assert_allclose(engs[0].get_flat_params(), engs[1].get_flat_params(), rtol=0, atol=2e-7)
    for which, atol in ((0, 1e-10), (1, 1e-15)):   # (first moments ~ 1e-5, second ~ 1e-10: the atomics' reordering noise is ~ 1e-12 / 1e-17)
        np.testing.assert_allclose(engs[0].get_flat_opt_state(which), engs[1].get_flat_opt_state(which), rtol=1e-3, atol=atol)
    for e in engs:
        e.close()


@pytest.mark.parametrize("plan", [True, False])
def test_split_scoring_pass_is_bit_identical(plan):
    """kprn_set_option("score_split", f): the scoring pass as two launches over disjoint tile ranges (the second one placed by the caller -- a
    data-parallel step puts it under its collective -- or, when forgotten, by whoever waits for the pass): the same bits as the single launch,
    with and without the identical-prefix plan, ragged last tile included."""
    case = Case(16389, 29)
    eng = case.engine(0, "auto", plan)
    eng.set_option("score_overlap", "1")
    b = eng.batch(case.idx, case.labels)
    eng.forward_async(b, 1)
    np.testing.assert_allclose(eng.read_probs(b.B), case.probs[:, 0], rtol=1e-5)
    eng.zero_pad_tokens()                            # (the zero-step-size optimiser step below does this too: MyOptimizer.lua:74-93)
    eng.forward_async(b, 1)
    ref = eng.read_probs(b.B).copy()
    for f in ("0.4", "0.75"):
        eng.set_option("score_split", f)
        eng.forward_async(b, 1)
        eng.forward_async_rest()
        assert np.array_equal(eng.read_probs(b.B), ref), f
        eng.forward_async(b, 1)                      # the second part never placed: read_probs places it
        assert np.array_equal(eng.read_probs(b.B), ref), f
        eng.forward_async(b, 1)                      # ... or the optimiser step, which waits for the pass
        eng.train_step(b, _ffi.make_opt(method=1, lr=0.0), 1, want_loss=False)
        assert np.array_equal(eng.read_probs(b.B), ref), f
        eng.set_option("score_rest_in_backward", "1")   # ... or the fused backward, right behind its last BPTT launch (beside the step's serial tail)
        eng.forward_async(b, 1)
        eng.train_step(b, _ffi.make_opt(method=1, lr=0.0), 1, want_loss=False)
        assert np.array_equal(eng.read_probs(b.B), ref), f
        eng.set_option("score_rest_in_backward", "0")
        eng.set_option("score_rest_before_bptt", "1")   # ... or behind the loss stage on the lowest-priority stream (into the first BPTT launch's idle tail)
        eng.forward_async(b, 1)
        eng.train_step(b, _ffi.make_opt(method=1, lr=0.0), 1, want_loss=False)
        assert np.array_equal(eng.read_probs(b.B), ref), f
        eng.forward_async(b, 1)                          # (and a pass nobody trains on: read_probs places the rest)
        assert np.array_equal(eng.read_probs(b.B), ref), f
        eng.set_option("score_rest_before_bptt", "0")
    eng.set_option("score_split", "0")
    eng.close()
