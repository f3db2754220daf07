"""-m gpu: the LDS-tiled GEMM + fused recurrent-step kernels (kprn_amd/csrc/gemm_tiled.hip) that carry the configurations the
D = H = 64 persistent kernels do not cover: "d = 64" reading B (D = H = 192, L = 2), run_scripts/config.sh's rnn, configs[3]'s
D = H = 384.  The tiled path takes a GEMM from 256 rows up, so these cases have >= 256 paths (the small-shape tests of
test_gpu_parity.py stay on the round-1 kernels).  Checked against the float64 oracle: scores 2e-5 of the largest, gradients 2e-4
of each tensor's largest, Adam steps 2e-4 absolute; and against the same library with the tiled kernels switched off."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from kprn_amd import _ffi, synth
from oracle.oracle import Oracle, make_cfg, make_opt

pytestmark = pytest.mark.gpu
SCORE_RTOL, GRAD_RTOL = 1e-4, 2e-4


def rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(1e-30, np.max(np.abs(b))))


def _check(eng, o64, theta, idx, labels, steps=0):
    b = eng.batch(idx, labels)
    out = eng.forward(b, 1, want=("probs", "all_probs", "path_scores"))
    ps, _, probs = o64.forward(theta, idx)
    assert rel_inf(out["path_scores"], ps) < 2e-5
    np.testing.assert_allclose(out["all_probs"], probs, rtol=SCORE_RTOL)
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        r = rel_inf(g[off:off + n], og[off:off + n])
        assert r < GRAD_RTOL, (nm, r)
    if steps:
        th, st = theta.copy(), o64.new_state()
        opt, oopt = _ffi.make_opt(method=1, lr=5e-3), make_opt(method=1, lr=5e-3)
        for s in range(steps):
            ol, _ = o64.train_step(th, st, oopt, idx, labels)
            gl = eng.train_step(b, opt)
            assert abs(gl - ol) < 2e-4 * max(1, abs(ol)), (s, gl, ol)
        assert float(np.max(np.abs(eng.get_flat_params() - th))) < 2e-4


def _lstm(dt, de, dr, H, L, Ve=700, Vr=9, seed=3, init=0.08):
    eng = _ffi.Engine(6, Ve, Vr, dt, de, dr, H, L)
    o64 = Oracle(make_cfg(Vt=6, Ve=Ve, Vr=Vr, dt=dt, de=de, dr=dr, H=H, L=L), np.float64)
    theta = o64.init_params(seed, init).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    return eng, o64, theta


@pytest.mark.parametrize("pairs,P,T", [(300, 3, 6), (257, 1, 4), (90, 7, 3)])
def test_reading_b_d192_h192_two_layers(pairs, P, T):
    """"d = 64" reading B: 64 / 64 / 64 -> D = H = 192, L = 2: fused step kernel on both layers, tiled dW / dx GEMMs"""
    eng, o64, theta = _lstm(64, 64, 64, 192, 2)
    idx, labels = synth.make_paths(pairs, P, T, Ve=700, seed=pairs)
    _check(eng, o64, theta, idx, labels, steps=3 if pairs == 300 else 0)


def test_hidden_size_not_a_multiple_of_the_unit_tile():
    """H = 100 (4 unit tiles of 32, the last one with 4 units), D = 200 (config.sh's 50 / 100 / 50), ragged row tiles"""
    eng, o64, theta = _lstm(50, 100, 50, 100, 1)
    idx, labels = synth.make_paths(173, 2, 6, Ve=700, seed=5)   # 346 paths: 2 full row tiles + 90 rows
    _check(eng, o64, theta, idx, labels, steps=3)


def test_lstm_with_config_sh_dimensions():
    """FastLSTM at config.sh's sizes (D = 200, H = 250): rows pitched at 8 bytes, fetched with 16-byte vectors at 4-byte granularity"""
    eng, o64, theta = _lstm(50, 100, 50, 250, 1, init=0.05)
    idx, labels = synth.make_paths(150, 2, 6, Ve=700, seed=12)
    _check(eng, o64, theta, idx, labels, steps=2)


@pytest.mark.parametrize("H,L", [(29, 1), (23, 2)])
def test_odd_dimensions_through_the_tiled_kernels(H, L):
    """D = 23 (5 / 11 / 7), H = 29 (one layer) or 23 (two): every row pitch is odd, so every 16-byte operand vector starts at a mere
    4-byte boundary and every K / N extent ends inside a vector (element-wise tail masks, gemm_tiled.hip)"""
    eng, o64, theta = _lstm(5, 11, 7, H, L)
    idx, labels = synth.make_paths(131, 3, 5, Ve=700, seed=21)   # 393 paths: 3 row tiles + 9 rows
    _check(eng, o64, theta, idx, labels, steps=3)


def test_configs3_shape_d384_h384_at_tiled_size():
    eng, o64, theta = _lstm(128, 128, 128, 384, 1, Vr=100, init=0.05)
    idx, labels = synth.make_paths(150, 2, 6, Ve=700, Vr=100, seed=6)
    _check(eng, o64, theta, idx, labels)


@pytest.mark.parametrize("use_relu,L,dims", [(1, 1, (50, 100, 50, 250)), (0, 1, (50, 100, 50, 250)), (1, 1, (50, 100, 50, 252)), (1, 2, (32, 32, 32, 96)),
                                             (0, 2, (32, 32, 32, 96)), (1, 1, (3, 5, 7, 33))])
def test_rnn_step_kernel(use_relu, L, dims):
    """nn.Recurrence + nn.MaskZero (OneModel.lua:240-266) through the fused step kernel: run_scripts/config.sh exactly (D = 200,
    H = 250: 8-byte row pitch), 16-byte shapes, two layers, odd dimensions (D = 15, H = 33: 4-byte row pitch)"""
    dt, de, dr, H = dims
    eng = _ffi.Engine(6, 700, 9, dt, de, dr, H, L, rnn_type=1, use_relu=use_relu, param_init=0.05)
    o64 = Oracle(make_cfg(Vt=6, Ve=700, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L, rnn_type=1, use_relu=use_relu), np.float64)
    theta = o64.init_params(7, 0.05).astype(np.float32).astype(np.float64)
    o64.zero_pad(theta)   # zero pad embeddings -> MaskZero masks the pad steps
    eng.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(140, 3, 6, Ve=700, seed=8)
    _check(eng, o64, theta, idx, labels, steps=3)


@pytest.mark.parametrize("kind,dims,L,pairs,P", [("rnn", (50, 100, 50, 250), 1, 1400, 3), ("lstm", (64, 64, 64, 192), 2, 1100, 4)])
def test_head_backward_wave_per_row_kernel_against_the_oracle(kind, dims, L, pairs, P):
    """From 4 096 paths up the generic pipelines' head backward is kk::k_head_bwd_w (a wave per row, a lane per column group, dH written
    in the same pass): run_scripts/config.sh's shape (H = 250: four column groups, the last one ragged) and reading B (H = 192: three),
    scores, loss and every gradient against the float64 oracle (tests/test_head_layout.py replays the kernel's index algebra on the CPU)."""
    dt, de, dr, H = dims
    if kind == "rnn":
        eng = _ffi.Engine(6, 700, 9, dt, de, dr, H, L, rnn_type=1, use_relu=1, param_init=0.05)
        o64 = Oracle(make_cfg(Vt=6, Ve=700, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L, rnn_type=1, use_relu=1), np.float64)
        theta = o64.init_params(7, 0.05).astype(np.float32).astype(np.float64)
        o64.zero_pad(theta)
        eng.set_flat_params(theta.astype(np.float32))
    else:
        eng, o64, theta = _lstm(dt, de, dr, H, L)
    idx, labels = synth.make_paths(pairs, P, 6, Ve=700, seed=pairs)
    assert pairs * P >= 4096
    _check(eng, o64, theta, idx, labels)


def test_tiled_kernels_agree_with_the_round1_kernels_at_the_bench_size():
    """16 384 paths, D = H = 192, L = 2: same library with KPRN_NO_TILED_GEMM / KPRN_NO_STEP_KERNEL (plain GEMM + element-wise
    kernels per step) -- two GPU implementations of every GEMM of the step"""
    code = textwrap.dedent("""
        import sys, json, numpy as np
        sys.path.insert(0, %r)
        from kprn_amd import _ffi, synth
        eng = _ffi.Engine(6, 100000, 9, 64, 64, 64, 192, 2, param_init=0.08)
        idx, labels = synth.make_paths(4096, 4, 6, Ve=100000, seed=3)
        b = eng.batch(idx, labels)
        out = eng.forward(b, 1, want=("probs",))
        loss = eng.backward(b, 1)
        g = eng.get_flat_grads()
        res = {"loss": float(loss), "probs": out["probs"][:3000].astype(float).tolist()}
        for nm, (off, shp) in eng.layout().items():
            v = g[off:off + int(np.prod(shp))].astype(np.float64)
            res[nm] = [float(np.abs(v).max()), float(v.sum()), float((v * np.cos(np.arange(v.size) * 0.37)).sum()), float(np.abs(v).sum())]
        print(json.dumps(res))
    """) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for tag, env in (("new", {}), ("old", {"KPRN_NO_TILED_GEMM": "1", "KPRN_NO_STEP_KERNEL": "1"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    assert abs(res["new"]["loss"] - res["old"]["loss"]) < 1e-6 * max(1.0, abs(res["old"]["loss"]))
    np.testing.assert_allclose(res["new"]["probs"], res["old"]["probs"], rtol=1e-5)
    for nm, ref in res["old"].items():
        if nm in ("loss", "probs"):
            continue
        got = res["new"][nm]
        assert abs(got[0] - ref[0]) < 1e-4 * max(1e-30, ref[0]), nm
        tol = 5e-5 * ref[3] + 1e-12
        assert abs(got[1] - ref[1]) < tol and abs(got[2] - ref[2]) < tol, (nm, got, ref)


# ---- BASELINE configs[3]: 20 M entities / 100 relations, d = 128, "bf16 MFMA LSTM" -------------------------------------------------
def _bf16_case(dims, H, L, pairs, P, Vr=100, seed=4, init=0.05):
    dt, de, dr = dims
    eng = _ffi.Engine(6, 700, Vr, dt, de, dr, H, L, compute_dtype=1)
    ref = _ffi.Engine(6, 700, Vr, dt, de, dr, H, L, compute_dtype=0)
    o64 = Oracle(make_cfg(Vt=6, Ve=700, Vr=Vr, dt=dt, de=de, dr=dr, H=H, L=L), np.float64)
    theta = o64.init_params(seed, init).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    ref.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(pairs, P, 6, Ve=700, Vr=Vr, seed=seed + 1)
    return eng, ref, o64, theta, idx, labels


@pytest.mark.parametrize("tile", ["small", "big"])
@pytest.mark.parametrize("dims,H,L,pairs,P", [((128, 128, 128), 384, 1, 150, 2), ((16, 32, 16), 64, 2, 101, 3), ((64, 64, 64), 192, 2, 129, 2)])
def test_bf16_storage_pipeline_is_tolerance_gated_against_the_f64_oracle(dims, H, L, pairs, P, tile, monkeypatch):
    """compute_dtype = 1 from 256 paths up: bf16 shadow tables / weights, bf16 activations and gate saves, v_mfma_f32_16x16x32_bf16 with
    fp32 accumulation, fp32 cell state and master parameters (kprn_amd/csrc/lstm_bf16.hip).  bf16 has 8 mantissa bits: scores within
    3e-2 of the largest, probabilities 2e-2 absolute, loss 3e-2, every gradient tensor within 6e-2 of its largest entry; and the
    result must differ from the fp32 path by far more than fp32 rounding (the bf16 path really ran).  (303 / 258 paths: not multiples of 8.)
    Both tile geometries of the bf16 GEMM (128 x 128 / 4 waves, 256 x 256 / 8 waves: the latter is chosen from 512 tiles up) run here."""
    monkeypatch.setenv("KPRN_BF16_TILE", tile)
    eng, ref, o64, theta, idx, labels = _bf16_case(dims, H, L, pairs, P)
    b, br = eng.batch(idx, labels), ref.batch(idx, labels)
    out = eng.forward(b, 1, want=("probs", "path_scores"))
    out32 = ref.forward(br, 1, want=("path_scores",))
    ps, _, probs = o64.forward(theta, idx)
    e16, e32 = rel_inf(out["path_scores"], ps), rel_inf(out32["path_scores"], ps)
    assert e16 < 3e-2 and e16 > 20 * e32, (e16, e32)
    np.testing.assert_allclose(out["probs"], probs[:, 0], atol=2e-2)
    loss = eng.backward(b, 1)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 3e-2 * max(1, abs(ol))
    g = eng.get_flat_grads()
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        r = rel_inf(g[off:off + n], og[off:off + n])
        assert r < 6e-2, (nm, r)
    # training: a few Adam steps stay close to the fp32 engine's (the master parameters are fp32; the shadows follow every update)
    opt = _ffi.make_opt(method=1, lr=2e-3)
    for _ in range(4):
        l16 = eng.train_step(b, opt)
        l32 = ref.train_step(br, opt)
        assert abs(l16 - l32) < 3e-2 * max(1.0, abs(l32))
    d = np.max(np.abs(eng.get_flat_params() - ref.get_flat_params()))
    assert d < 5e-3, d   # 4 steps of lr 2e-3 move a parameter by <= 8e-3
    # scores after training use the refreshed shadows (a stale shadow would score with the old parameters)
    s16 = eng.forward(b, 1, want=("probs",))["probs"]
    s32 = ref.forward(br, 1, want=("probs",))["probs"]
    np.testing.assert_allclose(s16, s32, atol=2e-2)


def test_ids_beyond_2_pow_24_gather_bit_exact():
    """configs[3] has 20 M entities: ids above 2^24 are not exact in float32 -- they stay int32 from the file to the kernel (SURVEY 7
    "Index dtype").  FeatureEmbedding output rows for such ids must be bit-identical to the table rows."""
    Ve = 20_000_000
    eng = _ffi.Engine(6, Ve, 100, 8, 8, 8, 24, 1)
    table = eng.get_param("entity_emb")
    ids = np.array([1, (1 << 24) - 1, 1 << 24, (1 << 24) + 1, (1 << 24) + 2, (1 << 24) + 3, 19_999_999, Ve], np.int32)
    idx = np.empty((len(ids), 1, 2, 3), np.int32)
    idx[..., 0] = 2
    idx[:, 0, 0, 1] = ids
    idx[:, 0, 1, 1] = ids[::-1]
    idx[..., 2] = 5
    x = eng.embed(idx)   # [N, T, D] = [type 8 | entity 8 | relation 8]
    assert np.array_equal(x[:, 0, 8:16], table[ids - 1]) and np.array_equal(x[:, 1, 8:16], table[ids[::-1] - 1])
    assert len({tuple(r) for r in table[ids - 1].round(7).tolist()}) == len(ids)   # distinct rows: a truncated id would alias
    # and the scoring path reads the same rows
    probs = eng.forward(eng.batch(idx), 1)["probs"]
    idx2 = idx.copy()
    idx2[3, 0, 0, 1] = 1 << 24          # the float32 image of 2^24 + 1
    assert eng.forward(eng.batch(idx2), 1)["probs"][3] != probs[3]
    eng.close()


@pytest.mark.parametrize("kind,dims,L,Vr", [("lstm", (64, 64, 64, 192), 2, 9), ("rnn", (50, 100, 50, 250), 1, 9), ("lstm", (128, 128, 128, 384), 1, 100), ("lstm", (16, 32, 16, 64), 2, 9)])
def test_generic_small_table_gradients_match_the_dx_route(kind, dims, L, Vr):
    """Round 5, generic fp32 pipelines: layer 0's type / relation gradients (tables and the matching column blocks of W_i2g) from G = dA^T [S_r | S_t]
    -- the one-hot selectors written over the last type columns of the saved step input, ONE dW product over [S | x_e], dx for the entity slice only
    (kprn_api.hip backward_layer0_small_tables).  Same engine, same batch: against the full dx product + table-gradient launches
    ("small_tables" = 0) every gradient agrees to fp32 reordering; against the float64 oracle inside the fp32 bar."""
    dt, de, dr, H = dims
    rt = 1 if kind == "rnn" else 0
    eng = _ffi.Engine(6, 900, Vr, dt, de, dr, H, L, rnn_type=rt, use_relu=1, param_init=0.06)
    eng.set_option("impl", "generic")
    o64 = Oracle(make_cfg(Vt=6, Ve=900, Vr=Vr, dt=dt, de=de, dr=dr, H=H, L=L, rnn_type=rt, use_relu=1), np.float64)
    theta = o64.init_params(17, 0.06).astype(np.float32).astype(np.float64)
    if rt:
        o64.zero_pad(theta)
    eng.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(170, 3, 6, Ve=900, Vr=Vr, seed=33)
    b = eng.batch(idx, labels)
    eng.profile(True)
    loss1 = eng.backward(b, 1)
    fam = eng.profile_get()
    assert "small_tables_finish" in fam and "gemm_bwd_dw_merged" in fam and "embed_scatter" not in fam, sorted(fam)
    g1 = eng.get_flat_grads().astype(np.float64)
    eng.set_option("small_tables", "0")
    eng.profile_reset()
    eng.profile(True)
    loss0 = eng.backward(b, 1)
    assert "embed_scatter" in eng.profile_get() and "small_tables_finish" not in eng.profile_get()
    g0 = eng.get_flat_grads().astype(np.float64)
    assert loss0 == loss1
    _, og, _ = o64.forward_backward(theta, idx, labels)
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        a, r = g1[off:off + n], g0[off:off + n]
        assert np.max(np.abs(a - r)) < 2e-5 * max(1e-30, np.max(np.abs(r))), (nm, float(np.max(np.abs(a - r))), float(np.max(np.abs(r))))
        assert rel_inf(a, og[off:off + n]) < GRAD_RTOL, (nm, rel_inf(a, og[off:off + n]))
    eng.close()
