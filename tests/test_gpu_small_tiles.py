"""-m gpu: small batches on tiles of ONE 16-row m-tile (kprn_amd/csrc/lstm_fused_fwd.hip fused::small_tiles, the NMT = 1 instantiations of
k_lstm_fwd / k_lstm_bwd): batches of <= 8 192 paths (D = H = 64, L = 2, fp32) put four times as many workgroups on the chip at a quarter of the
latency each -- the reference's own minibatch regime (run_scripts/config.sh:38: 128 pairs, test_from_checkpoint.lua:49: 512).  Same saves, same
dx layout, same oracle bars as the 64-path tiles (tests/test_gpu_parity.py pins those on small shapes); no identical-prefix plan in this mode.
Reference semantics: model/OneModel.lua:223-275, optimizer/MyOptimizer.lua:177-221."""
import numpy as np
import pytest

from kprn_amd import _ffi, synth
from oracle.oracle import Oracle, make_cfg, make_opt

pytestmark = pytest.mark.gpu
SHAPE = dict(Vt=6, Ve=3000, Vr=9, dt=16, de=32, dr=16, H=64, L=2)
T = 6


def rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(1e-30, np.max(np.abs(b))))


def mk(seed=3, small=True):
    eng = _ffi.Engine(SHAPE["Vt"], SHAPE["Ve"], SHAPE["Vr"], SHAPE["dt"], SHAPE["de"], SHAPE["dr"], SHAPE["H"], SHAPE["L"])
    eng.set_option("small_tiles", "1" if small else "0")
    o64 = Oracle(make_cfg(**SHAPE), np.float64)
    theta = o64.init_params(seed, 0.1).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    return eng, o64, theta


@pytest.mark.parametrize("pairs,P", [(1, 1), (5, 3), (16, 1), (17, 1), (50, 2), (333, 3), (2048, 4), (292, 28)])
def test_scores_and_gradients_against_the_f64_oracle(pairs, P):
    """1 .. 8 192 paths: one ragged tile, exactly one tile, many tiles per workgroup (8 192 paths = 512 tiles on 256 workgroups), P up to 28;
    every path's 46 scores, the pooled probabilities, the loss and every gradient."""
    eng, o64, theta = mk()
    idx, labels = synth.make_paths(pairs, P, T, Ve=SHAPE["Ve"], seed=pairs + P)
    b = eng.batch(idx, labels)
    assert b.executed_steps == pairs * P * T          # no identical-prefix plan in this mode
    out = eng.forward(b, 1, want=("probs", "all_probs", "pooled", "path_scores"))
    ps, pooled, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 3e-6
    np.testing.assert_allclose(out["pooled"], pooled, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(out["probs"], probs[:, 0], rtol=1e-5)
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1.0, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < 2e-4, nm
    eng.close()


def test_same_results_as_the_sixty_four_path_tiles():
    """The two tilings run the same per-row arithmetic (same MFMA sequence, same k order): scores to fp32 rounding of identical operations,
    gradients to the accumulation order of their per-workgroup partial sums."""
    idx, labels = synth.make_paths(700, 3, T, Ve=SHAPE["Ve"], seed=41)
    res = []
    for small in (True, False):
        eng, _, _ = mk(small=small)
        if not small:
            eng.set_option("prefix_plan", "0")
        b = eng.batch(idx, labels)
        out = eng.forward(b, 1, want=("path_scores", "probs"))
        loss = eng.backward(b, 1)
        res.append((out["path_scores"].astype(np.float64), out["probs"].astype(np.float64), loss, eng.get_flat_grads().astype(np.float64)))
        eng.close()
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=0, atol=1e-6 * np.max(np.abs(res[1][0])))
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=2e-6)
    assert abs(res[0][2] - res[1][2]) < 1e-6 * max(1.0, abs(res[1][2]))
    assert rel_inf(res[0][3], res[1][3]) < 2e-5


def test_twenty_adam_steps_of_reference_minibatches():
    """MyOptimizer.lua:177-221 at config.sh:38's 128 pairs, P per minibatch from the fixture distribution, lazy-exact Adam: loss of every step and
    the final parameters against the oracle; then scoring 512-pair minibatches with the trained model."""
    eng, o64, th = mk(seed=8)
    rng = np.random.default_rng(5)
    opt, oopt, st = _ffi.make_opt(method=1, lr=1e-3), make_opt(method=1, lr=1e-3), o64.new_state()
    for k in range(20):
        P = int(min(rng.geometric(0.57), 28))
        idx, labels = synth.make_paths(128, P, T, Ve=SHAPE["Ve"], seed=600 + k)
        ol, _ = o64.train_step(th, st, oopt, idx, labels)
        gl = eng.train_step_host(idx, labels, opt)
        assert abs(gl - ol) < 1e-5 * max(1.0, abs(ol)), (k, P, gl, ol)
    assert float(np.max(np.abs(eng.get_flat_params() - th))) < 5e-6
    for k, P in enumerate((1, 4)):
        idx, _ = synth.make_paths(512, P, T, Ve=SHAPE["Ve"], seed=650 + k)
        probs, _ = eng.forward_host(idx, 1)
        np.testing.assert_allclose(probs, o64.forward(th, idx)[2][:, 0], rtol=1e-5)
    eng.close()


def test_mode_boundary_and_option():
    """<= 8 192 paths: 16-row tiles, no plan; above: the 64-path tiles with their identical-prefix plan; "small_tiles" = "0": the plan everywhere"""
    eng, _, _ = mk()
    idx, labels = synth.make_paths(2048, 4, T, Ve=SHAPE["Ve"], seed=2)
    assert eng.batch(idx, labels).executed_steps == 2048 * 4 * T
    idx2, labels2 = synth.make_paths(2049, 4, T, Ve=SHAPE["Ve"], seed=2)
    assert eng.batch(idx2, labels2).executed_steps < 2049 * 4 * T      # (padded paths: the plan skips their identical leading steps)
    eng.set_option("small_tiles", "0")
    assert eng.batch(idx, labels).executed_steps < 2048 * 4 * T
    eng.close()
