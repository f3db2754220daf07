"""ctypes front-end for oracle/libkprn_oracle.so (see kprn_oracle.c).

TEST INFRASTRUCTURE ONLY -- parity unpinned (kprn_oracle.c header).  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by kprn_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Cfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("Vt", "Ve", "Vr", "dt", "de", "dr", "F", "numTypes", "H", "L", "C", "reducer", "K", "rnn_type", "use_relu")]


class Opt(C.Structure):
    _fields_ = [("method", C.c_int32), ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double),
                ("eps", C.c_double), ("lr_decay", C.c_double), ("regularize", C.c_int32),
                ("use_grad_clip", C.c_int32), ("grad_clip_norm", C.c_double), ("l2", C.c_double),
                ("bce_literal", C.c_int32)]


def build(force=False):
    so = os.path.join(_HERE, "libkprn_oracle.so")
    src = os.path.join(_HERE, "kprn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def make_cfg(Vt=6, Ve=100, Vr=9, dt=4, de=8, dr=4, F=3, numTypes=1, H=16, L=1, C_=46, reducer=2, K=5, rnn_type=0, use_relu=1):
    return Cfg(Vt, Ve, Vr, dt, de, dr, F, numTypes, H, L, C_, reducer, K, rnn_type, use_relu)


def make_opt(method=1, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, lr_decay=0.0, regularize=0,
             use_grad_clip=1, grad_clip_norm=5.0, l2=1e-3, bce_literal=0):
    return Opt(method, lr, beta1, beta2, eps, lr_decay, regularize, use_grad_clip, grad_clip_norm, l2, bce_literal)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """dtype: np.float64 (the reference's CPU arithmetic) or np.float32."""

    def __init__(self, cfg, dtype=np.float64):
        self.cfg = cfg
        self.dtype = np.dtype(dtype)
        self.sfx = "_f64" if self.dtype == np.float64 else "_f32"
        self.l = lib()
        f = getattr(self.l, "okprn_num_params" + self.sfx)
        f.restype = C.c_size_t
        self.n = int(f(C.byref(cfg)))
        getattr(self.l, "okprn_forward_backward" + self.sfx).restype = C.c_double
        getattr(self.l, "okprn_train_step" + self.sfx).restype = C.c_double

    @property
    def D(self):
        return self.cfg.dt + self.cfg.de + self.cfg.dr

    def layout(self):
        """dict name -> (offset, shape) in the flat vector, reference getParameters() order."""
        c = self.cfg
        rnn, gru = c.rnn_type == 1, c.rnn_type == 2
        out = np.zeros(3 + (4 if rnn else (6 if gru else 3)) * c.L + 3, dtype=np.int64)
        getattr(self.l, "okprn_layout" + self.sfx)(C.byref(c), _p(out))
        names = [("type_emb", (c.Vt, c.dt)), ("entity_emb", (c.Ve, c.de)), ("relation_emb", (c.Vr, c.dr))]
        for i in range(c.L):
            din = self.D if i == 0 else c.H
            if rnn:
                names += [(f"rnn{i + 1}.i2h.weight", (c.H, din)), (f"rnn{i + 1}.i2h.bias", (c.H,)),
                          (f"rnn{i + 1}.h2h.weight", (c.H, c.H)), (f"rnn{i + 1}.h2h.bias", (c.H,))]
            elif gru:
                names += [(f"gru{i + 1}.i2g.weight", (2 * c.H, din)), (f"gru{i + 1}.i2g.bias", (2 * c.H,)),
                          (f"gru{i + 1}.o2g.weight", (2 * c.H, c.H)), (f"gru{i + 1}.c_i2h.weight", (c.H, din)),
                          (f"gru{i + 1}.c_i2h.bias", (c.H,)), (f"gru{i + 1}.c_h2h.weight", (c.H, c.H))]
            else:
                names += [(f"lstm{i + 1}.i2g.weight", (4 * c.H, din)), (f"lstm{i + 1}.i2g.bias", (4 * c.H,)),
                          (f"lstm{i + 1}.o2g.weight", (4 * c.H, c.H))]
        names += [("out.weight", (c.C, c.H)), ("out.bias", (c.C,))]
        assert out[-1] == self.n
        return {nm: (int(out[k]), shp) for k, (nm, shp) in enumerate(names)}

    def init_params(self, seed, param_init=0.1, rnn_init=False):
        """uniform(-paramInit, paramInit) over every parameter (OneModel.lua:306-309); rnn_init (-rnnInitialization 1,
        rnnType rnn only): i2h.weight <- torch.eye(D, H) copied in STORAGE order into the [H, D] weight, h2h.weight <-
        eye(H), both biases <- 0 (OneModel.lua:310-322)."""
        rng = np.random.default_rng(seed)
        th = rng.uniform(-param_init, param_init, self.n).astype(self.dtype)
        if rnn_init and self.cfg.rnn_type == 1:
            lay = self.layout()
            for i in range(self.cfg.L):
                off, (H, din) = lay[f"rnn{i + 1}.i2h.weight"]
                th[off:off + H * din] = np.eye(din, H, dtype=self.dtype).ravel()
                off, (H, _) = lay[f"rnn{i + 1}.h2h.weight"]
                th[off:off + H * H] = np.eye(H, dtype=self.dtype).ravel()
                for nm in ("i2h.bias", "h2h.bias"):
                    off, shp = lay[f"rnn{i + 1}.{nm}"]
                    th[off:off + shp[0]] = 0
        return th

    def _idx(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        assert idx.ndim == 4 and idx.shape[3] == self.cfg.F
        return idx

    def embed(self, theta, idx):
        idx = self._idx(idx)
        B, P, T, _ = idx.shape
        x = np.empty((B * P, T, self.D), self.dtype)
        getattr(self.l, "okprn_embed" + self.sfx)(C.byref(self.cfg), _p(theta), _p(idx), C.c_int64(B * P), T, _p(x))
        return x

    def forward(self, theta, idx):
        """returns path_scores [B*P,C], pooled [B,C], probs [B,C]."""
        idx = self._idx(idx)
        B, P, T, _ = idx.shape
        c = self.cfg
        ps = np.empty((B * P, c.C), self.dtype)
        pooled = np.empty((B, c.C), self.dtype)
        probs = np.empty((B, c.C), self.dtype)
        theta = np.ascontiguousarray(theta, self.dtype)
        getattr(self.l, "okprn_forward" + self.sfx)(C.byref(c), _p(theta), _p(idx), B, P, T, _p(ps), _p(pooled), _p(probs))
        return ps, pooled, probs

    def forward_backward(self, theta, idx, labels, class_id=1, bce_literal=False, inv_batch=0.0):
        """returns loss, grad (flat), probs[B]."""
        idx = self._idx(idx)
        B, P, T, _ = idx.shape
        theta = np.ascontiguousarray(theta, self.dtype)
        labels = np.ascontiguousarray(labels, self.dtype)
        grad = np.zeros(self.n, self.dtype)
        probs = np.empty(B, self.dtype)
        loss = getattr(self.l, "okprn_forward_backward" + self.sfx)(
            C.byref(self.cfg), _p(theta), _p(idx), B, P, T, _p(labels), int(class_id), int(bool(bce_literal)),
            C.c_double(inv_batch), _p(grad), _p(probs))
        return float(loss), grad, probs

    def zero_pad(self, theta):
        getattr(self.l, "okprn_zero_pad" + self.sfx)(C.byref(self.cfg), _p(theta))

    def new_state(self):
        return {"s1": np.zeros(self.n, self.dtype), "s2": np.zeros(self.n, self.dtype),
                "g": np.zeros(self.n, self.dtype), "step": C.c_int64(0)}

    def train_step(self, theta, state, opt, idx, labels, class_id=1):
        """in-place MyOptimizer:trainBatch on theta/state; returns loss, probs[B]."""
        idx = self._idx(idx)
        B, P, T, _ = idx.shape
        labels = np.ascontiguousarray(labels, self.dtype)
        probs = np.empty(B, self.dtype)
        assert theta.dtype == self.dtype and theta.flags.c_contiguous
        loss = getattr(self.l, "okprn_train_step" + self.sfx)(
            C.byref(self.cfg), _p(theta), _p(state["g"]), _p(state["s1"]), _p(state["s2"]), C.byref(state["step"]),
            C.byref(opt), _p(idx), B, P, T, _p(labels), int(class_id), _p(probs))
        return float(loss), probs
