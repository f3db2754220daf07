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
    """FastLSTM at config.sh's sizes (D = 200, H = 250): the 8-byte-vector variant of the step kernel and of the tiled GEMMs"""
    eng, o64, theta = _lstm(50, 100, 50, 250, 1, init=0.05)
    idx, labels = synth.make_paths(150, 2, 6, Ve=700, seed=12)
    _check(eng, o64, theta, idx, labels, steps=2)


def test_configs3_shape_d384_h384_at_tiled_size():
    eng, o64, theta = _lstm(128, 128, 128, 384, 1, Vr=100, init=0.05)
    idx, labels = synth.make_paths(150, 2, 6, Ve=700, Vr=100, seed=6)
    _check(eng, o64, theta, idx, labels)


@pytest.mark.parametrize("use_relu,L,dims", [(1, 1, (50, 100, 50, 250)), (0, 1, (50, 100, 50, 250)), (1, 1, (50, 100, 50, 252)), (1, 2, (32, 32, 32, 96)),
                                             (0, 2, (32, 32, 32, 96)), (1, 1, (3, 5, 7, 33))])
def test_rnn_step_kernel(use_relu, L, dims):
    """nn.Recurrence + nn.MaskZero (OneModel.lua:240-266) through the fused step kernel: run_scripts/config.sh exactly (D = 200,
    H = 250: even leading dimensions -> 8-byte vectors), 16-byte shapes, two layers; odd dimensions (15 / 33) stay on the
    unfused kernels and must still be right"""
    dt, de, dr, H = dims
    eng = _ffi.Engine(6, 700, 9, dt, de, dr, H, L, rnn_type=1, use_relu=use_relu, param_init=0.05)
    o64 = Oracle(make_cfg(Vt=6, Ve=700, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L, rnn_type=1, use_relu=use_relu), np.float64)
    theta = o64.init_params(7, 0.05).astype(np.float32).astype(np.float64)
    o64.zero_pad(theta)   # zero pad embeddings -> MaskZero masks the pad steps
    eng.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(140, 3, 6, Ve=700, seed=8)
    _check(eng, o64, theta, idx, labels, steps=3)


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
