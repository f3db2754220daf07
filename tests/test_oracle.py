"""Pins oracle/kprn_oracle.c (parity unpinned vs the Lua reference, see its header) against
an independent PyTorch-CPU autograd implementation, finite differences and hand-computed
known answers."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle.oracle import Oracle, make_cfg, make_opt
from kprn_amd import synth
from tests import torch_ref


def small(L=2, reducer=2, numTypes=1, F=3, H=12, dt=4, de=5, dr=3):
    if L > 1:
        dt, de, dr = 4, 5, 3
        H = dt + de + dr  # multi-layer needs D == H (OneModel.lua:236,270-273)
    cfg = make_cfg(Vt=6, Ve=50, Vr=9, dt=dt, de=de, dr=dr, F=F, numTypes=numTypes, H=H, L=L, C_=46, reducer=reducer, K=2)
    orc = Oracle(cfg, np.float64)
    theta = orc.init_params(7, 0.5)
    idx, labels = synth.make_paths(5, 3, 4, F=F, Vt=6, Ve=50, Vr=9, num_types=numTypes, seed=3)
    return cfg, orc, theta, idx, labels


@pytest.mark.parametrize("L,reducer,numTypes,F", [(1, 2, 1, 3), (2, 2, 1, 3), (1, 0, 1, 3), (1, 1, 1, 3), (1, 2, 2, 4), (2, 2, 2, 5)])
def test_forward_backward_matches_torch_autograd(L, reducer, numTypes, F):
    cfg, orc, theta, idx, labels = small(L, reducer, numTypes, F)
    ps, pooled, probs = orc.forward(theta, idx)
    loss, grad, p = orc.forward_backward(theta, idx, labels, class_id=1, bce_literal=True)
    tl, tg, ts, tp = torch_ref.loss_and_grads(orc, theta, idx, labels, 1, reducer, 2)
    np.testing.assert_allclose(ps, ts, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(probs, tp, rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(p, tp[:, 0], rtol=1e-11)
    assert abs(loss - tl) < 1e-12 * max(1, abs(tl))
    np.testing.assert_allclose(grad, tg, rtol=1e-8, atol=1e-12)
    # the fused (p-t)/B backward equals the literal one away from saturation
    _, grad2, _ = orc.forward_backward(theta, idx, labels, class_id=1, bce_literal=False)
    np.testing.assert_allclose(grad2, grad, rtol=1e-8, atol=1e-12)


def test_class_id_selects_the_column():
    cfg, orc, theta, idx, labels = small(1)
    for cid in (1, 7, 46):
        loss, grad, p = orc.forward_backward(theta, idx, labels, class_id=cid)
        tl, tg, _, tp = torch_ref.loss_and_grads(orc, theta, idx, labels, cid)
        np.testing.assert_allclose(p, tp[:, cid - 1], rtol=1e-11)
        np.testing.assert_allclose(grad, tg, rtol=1e-8, atol=1e-12)
        lay = orc.layout()
        off, shp = lay["out.bias"]
        gb = grad[off:off + 46]
        assert np.count_nonzero(gb) == 1 and gb[cid - 1] != 0  # only column classId gets gradient


def test_finite_differences():
    cfg, orc, theta, idx, labels = small(2)
    loss, grad, _ = orc.forward_backward(theta, idx, labels)
    rng = np.random.default_rng(0)
    nz = np.flatnonzero(grad)
    for i in rng.choice(nz, 40, replace=False):
        h = 1e-6
        tp, tm = theta.copy(), theta.copy()
        tp[i] += h
        tm[i] -= h
        lp, _, _ = orc.forward_backward(tp, idx, labels)
        lm, _, _ = orc.forward_backward(tm, idx, labels)
        fd = (lp - lm) / (2 * h)
        assert abs(fd - grad[i]) <= 1e-6 * max(1.0, abs(grad[i])) + 1e-9, (i, fd, grad[i])


def test_embedding_is_a_pure_gather():
    cfg, orc, theta, idx, labels = small(1, numTypes=2, F=4)
    x = orc.embed(theta, idx)
    lay = orc.layout()
    Wt = theta[lay["type_emb"][0]:][:6 * 4].reshape(6, 4)
    We = theta[lay["entity_emb"][0]:][:50 * 5].reshape(50, 5)
    Wr = theta[lay["relation_emb"][0]:][:9 * 3].reshape(9, 3)
    flat = idx.reshape(-1, idx.shape[2], 4)
    want = np.concatenate([Wt[flat[..., 0] - 1] + Wt[flat[..., 1] - 1], We[flat[..., 2] - 1], Wr[flat[..., 3] - 1]], axis=2)
    assert np.array_equal(x, want)  # bit-exact, order type | entity | relation (FeatureEmbedding.lua:118)


def test_lse_known_answer():
    # LogSumExp.lua:13-36 on a hand-made score matrix: y = m + log sum exp(s - m)
    cfg = make_cfg(Vt=6, Ve=50, Vr=9, dt=4, de=5, dr=3, H=12, L=1)
    orc = Oracle(cfg)
    theta = orc.init_params(1, 0.3)
    idx, _ = synth.make_paths(2, 4, 3, Ve=50, seed=5)
    ps, pooled, probs = orc.forward(theta, idx)
    s3 = ps.reshape(2, 4, 46)
    m = s3.max(axis=1)
    want = m + np.log(np.exp(s3 - m[:, None, :]).sum(axis=1))
    np.testing.assert_allclose(pooled, want, rtol=1e-14)
    np.testing.assert_allclose(probs, 1 / (1 + np.exp(-want)), rtol=1e-14)
    # single path: LSE is the identity
    idx1, _ = synth.make_paths(3, 1, 3, Ve=50, seed=6)
    ps1, pooled1, _ = orc.forward(theta, idx1)
    np.testing.assert_allclose(pooled1, ps1, rtol=1e-15)


def _adam_numpy(x, g, m, v, t, lr, b1, b2, eps):
    m[:] = b1 * m + (1 - b1) * g
    v[:] = b2 * v + (1 - b2) * g * g
    step = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    x -= step * m / (np.sqrt(v) + eps)


@pytest.mark.parametrize("method", [1, 0])
@pytest.mark.parametrize("regularize", [0, 1])
def test_train_step_semantics(method, regularize):
    """MyOptimizer.lua:177-221 step by step in numpy around the oracle's own forward/backward."""
    cfg, orc, theta, idx, labels = small(2)
    opt = make_opt(method=method, lr=1e-2, lr_decay=0.0167, regularize=regularize, use_grad_clip=1,
                   grad_clip_norm=0.05, l2=1e-3, bce_literal=0)
    th = theta.copy()
    st = orc.new_state()
    ref = theta.copy()
    m = np.zeros_like(ref)
    v = np.zeros_like(ref)
    for it in range(1, 4):
        loss, _ = orc.train_step(th, st, opt, idx, labels)
        orc.zero_pad(ref)
        l2, g, _ = orc.forward_backward(ref, idx, labels)
        assert abs(loss - l2) < 1e-13
        if regularize:
            nrm = np.linalg.norm(g)
            if nrm > 0.05:
                g = g * (0.05 / nrm)
            g = g + 1e-3 * ref
        if method == 1:
            _adam_numpy(ref, g, m, v, it, 1e-2, 0.9, 0.999, 1e-8)
        else:
            clr = 1e-2 / (1 + (it - 1) * 0.0167)
            m += g * g
            ref -= clr * g / (np.sqrt(m) + 1e-10)
        orc.zero_pad(ref)
        np.testing.assert_allclose(th, ref, rtol=1e-12, atol=1e-15)
    assert st["step"].value == 3
    lay = orc.layout()
    for nm, V, d in (("type_emb", 6, 4), ("entity_emb", 50, 5), ("relation_emb", 9, 3)):
        off = lay[nm][0]
        assert np.all(th[off + (V - 1) * d: off + V * d] == 0)  # zeroPadTokens rows (1-based row V)


def test_f32_build_tracks_f64():
    cfg, orc, theta, idx, labels = small(2)
    o32 = Oracle(cfg, np.float32)
    ps, _, probs = orc.forward(theta, idx)
    ps32, _, probs32 = o32.forward(theta.astype(np.float32), idx)
    np.testing.assert_allclose(ps32, ps, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(probs32, probs, rtol=2e-5)
    l, g, _ = orc.forward_backward(theta, idx, labels)
    l32, g32, _ = o32.forward_backward(theta.astype(np.float32), idx, labels)
    assert abs(l - l32) < 1e-5
    np.testing.assert_allclose(g32, g, rtol=1e-3, atol=1e-6)


def test_threads_do_not_change_the_answer():
    import os
    import subprocess
    import sys
    code = ("import numpy as np;from tests.test_oracle import small;"
            "cfg,orc,theta,idx,labels=small(2);l,g,p=orc.forward_backward(theta,idx,labels);"
            "print(repr(l), float(np.abs(g).sum()))")
    outs = []
    for nt in ("1", "3"):
        env = dict(os.environ, OMP_NUM_THREADS=nt)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=env, cwd=os.path.dirname(os.path.dirname(__file__))).decode().split())
    assert abs(float(outs[0][0]) - float(outs[1][0])) < 1e-13
    assert abs(float(outs[0][1]) - float(outs[1][1])) < 1e-10


# ---- rnnType "rnn": nn.Recurrence + nn.MaskZero [A7] --------------------------------------------------------
def _torch_rnn_loss(o, theta, idx, labels, use_relu):
    """independent restatement with torch autograd: the mask follows the actual step input rows of each layer"""
    import torch
    c = o.cfg
    lay = o.layout()
    th = torch.tensor(theta, dtype=torch.float64, requires_grad=True)

    def P(name):
        off, shp = lay[name]
        return th[off:off + int(np.prod(shp))].reshape(shp)
    ids = torch.tensor(idx.astype(np.int64)) - 1
    B, Pn, T, F = idx.shape
    x = torch.cat([P("type_emb")[ids[..., 0]], P("entity_emb")[ids[..., 1]], P("relation_emb")[ids[..., 2]]], dim=-1).reshape(B * Pn, T, -1)
    seq = [x[:, t] for t in range(T)]
    for l in range(c.L):
        Wi, bi, Wh, bh = (P(f"rnn{l + 1}.{n}") for n in ("i2h.weight", "i2h.bias", "h2h.weight", "h2h.bias"))
        h = torch.zeros(B * Pn, c.H, dtype=torch.float64)
        out = []
        for t in range(T):
            a = seq[t] @ Wi.T + bi + h @ Wh.T + bh
            v = torch.relu(a) if use_relu else torch.tanh(a)
            m = (seq[t] != 0).any(dim=1, keepdim=True).to(torch.float64)
            h = v * m
            out.append(h)
        seq = out
    s = seq[-1] @ P("out.weight").T + P("out.bias")
    y = torch.logsumexp(s.reshape(B, Pn, -1), dim=1)
    p = torch.sigmoid(y)[:, 0]
    t = torch.tensor(labels, dtype=torch.float64)
    eps = 1e-12
    loss = -(t * torch.log(p + eps) + (1 - t) * torch.log(1 - p + eps)).mean()
    loss.backward()
    return float(loss), th.grad.numpy()


@pytest.mark.parametrize("use_relu,L", [(1, 1), (0, 2), (1, 2)])
def test_rnn_cell_matches_torch_autograd(use_relu, L):
    cfg = make_cfg(Vt=6, Ve=60, Vr=9, dt=4, de=4, dr=4, H=12, L=L, rnn_type=1, use_relu=use_relu)
    o = Oracle(cfg, np.float64)
    theta = o.init_params(2, 0.4)
    o.zero_pad(theta)
    idx, labels = synth.make_paths(9, 3, 5, Ve=60, seed=4)
    loss, g, _ = o.forward_backward(theta, idx, labels, class_id=1, bce_literal=True)
    tl, tg = _torch_rnn_loss(o, theta, idx, labels, use_relu)
    assert abs(loss - tl) < 1e-12
    assert np.max(np.abs(g - tg)) < 1e-11


def test_rnn_golden_vector():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_rnn_small.npz"))
    c = [int(x) for x in z["cfg"]]
    cfg = make_cfg(Vt=c[0], Ve=c[1], Vr=c[2], dt=c[3], de=c[4], dr=c[5], F=c[6], numTypes=c[7], H=c[8], L=c[9], C_=c[10], reducer=c[11], K=c[12],
                   rnn_type=c[13], use_relu=c[14])
    o = Oracle(cfg, np.float64)
    ps, pooled, probs = o.forward(z["theta"].astype(np.float64), z["idx"])
    np.testing.assert_allclose(ps, z["path_scores"], rtol=0, atol=1e-12)
    loss, g, _ = o.forward_backward(z["theta"].astype(np.float64), z["idx"], z["labels"])
    assert abs(loss - float(z["loss"])) < 1e-12
    np.testing.assert_allclose(g, z["grad"], rtol=0, atol=1e-12)


# ---- rnnType "gru": nn.GRU [A8] -----------------------------------------------------------------------------
@pytest.mark.parametrize("L", [1, 2])
def test_gru_cell_matches_torch_autograd(L):
    import torch
    cfg = make_cfg(Vt=6, Ve=60, Vr=9, dt=4, de=4, dr=4, H=12, L=L, rnn_type=2)
    o = Oracle(cfg, np.float64)
    theta = o.init_params(2, 0.4)
    idx, labels = synth.make_paths(9, 3, 5, Ve=60, seed=4)
    loss, g, _ = o.forward_backward(theta, idx, labels, class_id=1, bce_literal=True)
    lay = o.layout()
    th = torch.tensor(theta, dtype=torch.float64, requires_grad=True)

    def P(name):
        off, shp = lay[name]
        return th[off:off + int(np.prod(shp))].reshape(shp)
    ids = torch.tensor(idx.astype(np.int64)) - 1
    B, Pn, T, F = idx.shape
    x = torch.cat([P("type_emb")[ids[..., 0]], P("entity_emb")[ids[..., 1]], P("relation_emb")[ids[..., 2]]], dim=-1).reshape(B * Pn, T, -1)
    seq = [x[:, t] for t in range(T)]
    H = cfg.H
    for l in range(L):
        Wi, bi, Wo, Wc, bc, Uc = (P(f"gru{l + 1}.{n}") for n in ("i2g.weight", "i2g.bias", "o2g.weight", "c_i2h.weight", "c_i2h.bias", "c_h2h.weight"))
        h = torch.zeros(B * Pn, H, dtype=torch.float64)
        out = []
        for t in range(T):
            gz = torch.sigmoid(seq[t] @ Wi.T + bi + h @ Wo.T)
            r, z = gz[:, :H], gz[:, H:]
            n_ = torch.tanh(seq[t] @ Wc.T + bc + (r * h) @ Uc.T)  # reset applied BEFORE the recurrent product (Element-Research nn.GRU)
            h = (1 - z) * n_ + z * h
            out.append(h)
        seq = out
    s = seq[-1] @ P("out.weight").T + P("out.bias")
    p = torch.sigmoid(torch.logsumexp(s.reshape(B, Pn, -1), dim=1))[:, 0]
    t = torch.tensor(labels, dtype=torch.float64)
    tl = -(t * torch.log(p + 1e-12) + (1 - t) * torch.log(1 - p + 1e-12)).mean()
    tl.backward()
    assert abs(loss - float(tl.detach())) < 1e-12
    assert np.max(np.abs(g - th.grad.numpy())) < 1e-11
