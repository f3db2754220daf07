"""-m gpu: one recurrent layer of the wide fp32 shapes as ONE persistent launch, forward and BPTT (kprn_amd/csrc/layer_f32_persist.hip k_layer / k_bptt) -- "d = 64" reading B
(D = H = 192, L = 2), run_scripts/config.sh as shipped (rnn + MaskZero, D = 200, H = 250), FastLSTM at config.sh's sizes, H not a multiple of 64 --
against the float64 oracle (scores, every class probability, every gradient through the unchanged generic backward, Adam steps) and against the
per-step launches it replaces (kprn_set_option "persist_layers" = 0: the same arithmetic in another accumulation order).  "persist_layers" = 2
takes the launch at any batch size (the default waits for a tile per CU), so ragged last tiles, single-tile workgroups and several tiles per
workgroup are all reached at sizes the oracle handles."""
import numpy as np
import pytest

from kprn_amd import _ffi, synth
from oracle.oracle import Oracle, make_cfg, make_opt

pytestmark = pytest.mark.gpu


def rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(1e-30, np.max(np.abs(b))))


def _mk(kind, dims, L, Vr=9, use_relu=1, init=0.07, seed=5):
    dt, de, dr, H = dims
    rt = {"lstm": 0, "rnn": 1, "gru": 2}[kind]
    eng = _ffi.Engine(6, 800, Vr, dt, de, dr, H, L, rnn_type=rt, use_relu=use_relu, param_init=init)
    eng.set_option("impl", "generic")
    eng.set_option("persist_layers", "2")
    o64 = Oracle(make_cfg(Vt=6, Ve=800, Vr=Vr, dt=dt, de=de, dr=dr, H=H, L=L, rnn_type=rt, use_relu=use_relu), np.float64)
    theta = o64.init_params(seed, init).astype(np.float32).astype(np.float64)
    if rt == 1:
        o64.zero_pad(theta)   # zero pad embeddings -> MaskZero masks the pad steps
    eng.set_flat_params(theta.astype(np.float32))
    return eng, o64, theta


CASES = [("lstm", (64, 64, 64, 192), 2, 300, 3, 6),     # reading B: 900 paths = 14 tiles + 4 rows, both layers through the launch
         ("lstm", (64, 64, 64, 192), 2, 129, 1, 4),     # T = 4, 129 paths: two full tiles + one row
         ("rnn", (50, 100, 50, 250), 1, 280, 3, 6),     # config.sh as shipped: K padded 200 -> 224, H 250 -> 256, MaskZero
         ("rnn", (32, 32, 32, 96), 2, 150, 2, 6),       # two rnn layers: the upper layer's mask follows h of the lower one
         ("lstm", (50, 100, 50, 250), 1, 150, 2, 6),    # FastLSTM at config.sh's sizes: four chunks, the last one 58 units
         ("lstm", (16, 32, 16, 80), 1, 100, 2, 3),      # H = 80: second chunk a quarter full
         ("lstm", (64, 64, 64, 192), 1, 40, 1, 1)]      # T = 1: no recurrent half at all


@pytest.mark.parametrize("kind,dims,L,pairs,P,T", CASES)
def test_persistent_layer_against_the_f64_oracle(kind, dims, L, pairs, P, T):
    eng, o64, theta = _mk(kind, dims, L)
    idx, labels = synth.make_paths(pairs, P, T, Ve=800, seed=pairs + T)
    b = eng.batch(idx, labels)
    eng.profile(True)
    out = eng.forward(b, 1, want=("probs", "all_probs", "path_scores"))
    fam = eng.profile_get()
    assert ("rnn_layer_fwd" if kind == "rnn" else "lstm_layer_fwd") in fam and "lstm_step_fwd" not in fam and "rnn_step_fwd" not in fam, sorted(fam)
    ps, _, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5, rel_inf(out["path_scores"], ps)
    np.testing.assert_allclose(out["all_probs"], probs, rtol=1e-4)
    # training forward (saves in the generic backward's layouts) + the unchanged backward: every gradient
    loss = eng.backward(b, 1)
    fam = eng.profile_get()
    # ... BPTT through the layer is ONE launch too (cell backward + recurrent product of all T steps): no per-step gate-backward / dh launches
    assert ("rnn_layer_bwd" if kind == "rnn" else "lstm_layer_bwd") in fam and "lstm_gates_bwd" not in fam and "rnn_cell_bwd" not in fam and "gemm_o2g_bwd_dh" not in fam, sorted(fam)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < 2e-4, (nm, rel_inf(g[off:off + n], og[off:off + n]))
    # ... and against the per-step launches on the same engine
    eng.set_option("persist_layers", "0")
    out0 = eng.forward(b, 1, want=("probs", "path_scores"))
    assert rel_inf(out["path_scores"], out0["path_scores"].astype(np.float64)) < 2e-6
    eng.backward(b, 1)
    g0 = eng.get_flat_grads().astype(np.float64)
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], g0[off:off + n]) < 2e-5, (nm, rel_inf(g[off:off + n], g0[off:off + n]))
    eng.close()


GRU_CASES = [((50, 100, 50, 250), 1, 150, 2, 6),    # bench.py --dims gru: K padded 200 -> 224, four chunks (the last one 58 units), two [r; z] passes + the candidate pass
             ((64, 64, 64, 192), 2, 129, 1, 4),     # three chunks: the second [r; z] pass has one chunk only; both layers through the launch; a ragged last tile
             ((16, 32, 16, 80), 1, 100, 2, 3),      # H = 80: second chunk a quarter full
             ((16, 32, 16, 64), 2, 300, 3, 6),      # one chunk: half of every pass's n-tiles idle; 900 paths = 14 tiles + 4 rows
             ((64, 64, 64, 192), 1, 40, 1, 1)]      # T = 1: no recurrent half, no r * h' hand-over


@pytest.mark.parametrize("dims,L,pairs,P,T", GRU_CASES)
def test_persistent_gru_layer_against_the_f64_oracle(dims, L, pairs, P, T):
    """nn.GRU (OneModel.lua:237-238) through k_layer<2, NCH, SAVE>: both dependent products of a step inside the launch (r * h' takes h_{t-1}'s place in the
    LDS tile between them), and its BPTT through k_bptt<2, 1, UP> (three K chunks a step over one [c_h2h^T | o2g^T] stream).  Scores, every class probability
    and every gradient against the float64 oracle and against the per-step launches (measured: 5e-7 / 1.6e-6 / 1.2e-6)."""
    dt, de, dr, H = dims
    eng = _ffi.Engine(6, 800, 9, dt, de, dr, H, L, rnn_type=2, param_init=0.07)
    eng.set_option("impl", "generic")
    eng.set_option("persist_layers", "2")
    o64 = Oracle(make_cfg(Vt=6, Ve=800, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L, rnn_type=2), np.float64)
    theta = o64.init_params(5, 0.07).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(pairs, P, T, Ve=800, seed=pairs + T)
    b = eng.batch(idx, labels)
    eng.profile(True)
    out = eng.forward(b, 1, want=("probs", "all_probs", "path_scores"))
    fam = eng.profile_get()
    assert "gru_layer_fwd" in fam and "gru_cell_fwd" not in fam and "gemm_o2g_fwd" not in fam and "gemm_i2g_fwd" not in fam, sorted(fam)
    ps, _, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5, rel_inf(out["path_scores"], ps)
    np.testing.assert_allclose(out["all_probs"], probs, rtol=1e-4)
    loss = eng.backward(b, 1)
    fam = eng.profile_get()
    assert "gru_cell_fwd" not in fam and "gemm_o2g_fwd" not in fam, sorted(fam)   # (the training forward is the launch too)
    # ... and so is BPTT through the layer (k_bptt<2, 1, UP>: the cell backward of all T steps and both recurrent products of a step)
    assert "gru_layer_bwd" in fam and "gru_cell_bwd" not in fam and "gemm_o2g_bwd_dh" not in fam, sorted(fam)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < 2e-4, (nm, rel_inf(g[off:off + n], og[off:off + n]))
    eng.set_option("persist_layers", "0")
    out0 = eng.forward(b, 1, want=("probs", "path_scores"))
    assert rel_inf(out["path_scores"], out0["path_scores"].astype(np.float64)) < 2e-6
    eng.backward(b, 1)
    g0 = eng.get_flat_grads().astype(np.float64)
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], g0[off:off + n]) < 2e-5, (nm, rel_inf(g[off:off + n], g0[off:off + n]))
    eng.close()


@pytest.mark.parametrize("kind,dims,L", [("lstm", (64, 64, 64, 192), 2), ("rnn", (50, 100, 50, 250), 1), ("gru", (50, 100, 50, 250), 1), ("gru", (32, 64, 32, 128), 2)])
def test_adam_steps_through_the_persistent_layers(kind, dims, L):
    """Three Adam steps: the loss follows the float64 oracle (measured 1e-7), the parameters equal those of the per-step launches to fp32 reordering
    (measured 4e-7: scripts/gpu_probe_layer_persist.py), and both sit on the oracle's walk.  Against the oracle a max bar would be vacuous: Adam's first
    steps move an element by lr whatever its gradient, so the handful of elements whose gradient is at rounding level (here 24 of 80 000 entity
    elements, the same ones on both paths) may step the other way -- the bar is the rms and the share of elements further than 2e-4."""
    eng, o64, theta = _mk(kind, dims, L)
    ref, _, _ = _mk(kind, dims, L)
    ref.set_option("persist_layers", "0")
    idx, labels = synth.make_paths(200, 3, 6, Ve=800, seed=77)
    b, br = eng.batch(idx, labels), ref.batch(idx, labels)
    th, st = theta.copy(), o64.new_state()
    opt, oopt = _ffi.make_opt(method=1, lr=5e-3), make_opt(method=1, lr=5e-3)
    for s in range(3):
        ol, _ = o64.train_step(th, st, oopt, idx, labels)
        gl = eng.train_step(b, opt)
        ref.train_step(br, opt)
        assert abs(gl - ol) < 1e-5 * max(1, abs(ol)), (s, gl, ol)
    got = eng.get_flat_params().astype(np.float64)
    assert float(np.max(np.abs(got - ref.get_flat_params()))) < 2e-5
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        d = np.abs(got[off:off + n] - th[off:off + n])
        assert np.sqrt(np.mean(d * d)) < 5e-5 and np.mean(d > 2e-4) < 2e-3, (nm, float(np.sqrt(np.mean(d * d))), float(np.mean(d > 2e-4)))
    eng.close()
    ref.close()


@pytest.mark.parametrize("kind,dims,L,pairs,P", [("lstm", (64, 64, 64, 192), 2, 4400, 4), ("rnn", (50, 100, 50, 250), 1, 5700, 3), ("gru", (50, 100, 50, 250), 1, 5700, 3)])
def test_a_tile_per_cu_takes_the_launch_by_default(kind, dims, L, pairs, P):
    """>= 16 384 paths (a 64-path tile for every CU): the default configuration uses the persistent launch; a second pass is bit-identical; scores of a
    sample of pairs against the float64 oracle"""
    dt, de, dr, H = dims
    rt = {"lstm": 0, "rnn": 1, "gru": 2}[kind]
    eng = _ffi.Engine(6, 5000, 9, dt, de, dr, H, L, rnn_type=rt, use_relu=1, param_init=0.06)
    eng.set_option("impl", "generic")
    o64 = Oracle(make_cfg(Vt=6, Ve=5000, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L, rnn_type=rt, use_relu=1), np.float64)
    theta = eng.get_flat_params().astype(np.float64)
    idx, labels = synth.make_paths(pairs, P, 6, Ve=5000, seed=3)
    b = eng.batch(idx, labels)
    eng.profile(True)
    out = eng.forward(b, 1, want=("probs",))
    assert kind + "_layer_fwd" in eng.profile_get()
    again = eng.forward(b, 1, want=("probs",))
    assert np.array_equal(out["probs"], again["probs"])
    sel = np.arange(0, pairs, 37)
    _, _, probs = o64.forward(theta, idx[sel])
    np.testing.assert_allclose(out["probs"][sel], probs[:, 0], rtol=1e-4)
    eng.close()


@pytest.mark.parametrize("kind,dims,L", [("lstm", (64, 64, 64, 192), 2), ("rnn", (50, 100, 50, 250), 1), ("gru", (50, 100, 50, 250), 1), ("gru", (32, 64, 32, 128), 2)])
def test_full_batch_agrees_with_the_step_launches(kind, dims, L):
    """bench.py's own batch (65 536 paths: four tiles per workgroup on every CU, the DMA rings and the counted waits under full load -- the timing regime
    the small cases cannot reach): probabilities and every gradient of the persistent launches against the per-step launches on the same engine.  The
    counted `vmcnt` waits assume that loads retire in issue order whether they land in registers or (LDS-DMA) in LDS; a weight group used before it
    landed would show here as errors of order 1e-2."""
    dt, de, dr, H = dims
    rt = {"lstm": 0, "rnn": 1, "gru": 2}[kind]
    eng = _ffi.Engine(6, 200000, 9, dt, de, dr, H, L, rnn_type=rt, use_relu=1, param_init=0.06)
    eng.set_option("impl", "generic")
    idx, labels = synth.make_paths(16384, 4, 6, Ve=200000, seed=19)
    b = eng.batch(idx, labels)
    res = {}
    for mode in ("1", "0", "1"):
        eng.set_option("persist_layers", mode)
        eng.profile_reset()
        eng.profile(True)
        p = eng.forward(b, 1, want=("probs",))["probs"].astype(np.float64)
        eng.backward(b, 1)
        fam = eng.profile_get()
        assert (any(k + "_layer_bwd" in fam for k in ("lstm", "rnn", "gru")) and any(k + "_layer_fwd" in fam for k in ("lstm", "rnn", "gru"))) == (mode == "1"), sorted(fam)
        g = eng.get_flat_grads().astype(np.float64)
        if mode in res:   # a second pass through the persistent launches: the same answer again (to the atomics' reordering in the dW products)
            assert np.max(np.abs(p - res[mode][0])) < 1e-7
        res[mode] = (p, g)
    assert np.max(np.abs(res["1"][0] - res["0"][0])) < 2e-6
    for nm, (off, shp) in eng.layout().items():
        if nm == "entity_emb":
            continue   # (200 000 x d_e: compared through its column sums below)
        n = int(np.prod(shp))
        a, r = res["1"][1][off:off + n], res["0"][1][off:off + n]
        assert np.max(np.abs(a - r)) < 1e-4 * max(1e-30, np.max(np.abs(r))), (nm, float(np.max(np.abs(a - r))), float(np.max(np.abs(r))))
    off, shp = eng.layout()["entity_emb"]
    a = res["1"][1][off:off + int(np.prod(shp))].reshape(shp)
    r = res["0"][1][off:off + int(np.prod(shp))].reshape(shp)
    assert np.max(np.abs(a - r)) < 1e-4 * np.max(np.abs(r))
    eng.close()


def test_mixed_route_step_forward_with_persistent_bptt():
    """layer_f32_persist.hip decides the forward (lp32::supported: needs Din % 4 == 0, the LDS-DMA piece) and the BPTT (lp32::bptt_supported: no
    dependence on Din) independently.  D = 18 + 30 + 18 = 66 is not a multiple of 4: the forward of layer 0 runs on the per-step launches, its BPTT on
    the persistent launch over the saves those launches wrote -- the two share the generic layouts, so every gradient must still meet the oracle."""
    eng, o64, theta = _mk("lstm", (18, 30, 18, 96), 1)
    idx, labels = synth.make_paths(140, 2, 5, Ve=800, seed=21)
    b = eng.batch(idx, labels)
    eng.profile(True)
    out = eng.forward(b, 1, want=("path_scores",))
    loss = eng.backward(b, 1)
    fam = eng.profile_get()
    assert "lstm_step_fwd" in fam and "lstm_layer_fwd" not in fam, sorted(fam)      # forward: per step
    assert "lstm_layer_bwd" in fam and "lstm_gates_bwd" not in fam, sorted(fam)      # BPTT: one launch
    ps, _, _ = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < 2e-4, (nm, rel_inf(g[off:off + n], og[off:off + n]))
    eng.close()
