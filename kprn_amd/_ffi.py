"""ctypes binding of libkprn.so (include/kprn.h).  There is NO CPU path: if the HIP library
is missing or no GPU is visible, everything here fails loudly.

The LuaJIT twin of this file is bindings/kprn.lua.
"""
import ctypes as C
import os
import re
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KPRN_LIB") or os.path.join(_HERE, "libkprn.so")   # (KPRN_LIB: a measurement build, scripts/build_variants.py)
HEADER = os.path.join(_HERE, "..", "include", "kprn.h")


class KprnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"kprn error {code}: {msg}")
        self.code = code
        self.msg = msg


E_ARG, E_INDEX, E_DEVICE, E_UNSUPPORTED, E_IO, E_NOMEM = -1, -2, -3, -4, -5, -6


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("Vt", "Ve", "Vr", "dt", "de", "dr", "F", "num_types", "H", "L", "C",
                                          "rnn_type", "use_relu", "rnn_init", "compute_dtype", "reducer", "K", "device_id", "rank", "world")] + \
               [("param_init", C.c_float), ("seed", C.c_uint64), ("stream", C.c_void_p)]


class Opt(C.Structure):
    _fields_ = [("method", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("lr_decay", C.c_float), ("regularize", C.c_int32), ("use_grad_clip", C.c_int32),
                ("grad_clip_norm", C.c_float), ("l2", C.c_float), ("bce_literal", C.c_int32), ("entity_update", C.c_int32)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("total_ms", C.c_double), ("launches", C.c_int64)]


def declared_symbols():
    """every function include/kprn.h declares (used by the CPU test that the library exports them all)."""
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"\b(kprn_[a-z_0-9]+)\s*\(", txt)))


def dp_unique_id(rccl_path=None):
    """128 bytes from ncclGetUniqueId (rank 0 draws it; the caller broadcasts it to every rank's Engine.dp_init)"""
    L = lib()
    buf = (C.c_char * 128)()
    rc = L.kprn_dp_unique_id(rccl_path.encode() if rccl_path else None, buf)
    if rc != 0:
        raise KprnError(rc, L.kprn_last_error(None).decode())
    return bytes(buf.raw)


def dp_available(rccl_path=None):
    return lib().kprn_dp_available(rccl_path.encode() if rccl_path else None) == 0


def torch_rccl_path():
    """the librccl.so the running torch build ships (the copy torch.distributed's "nccl" backend has loaded), or None"""
    try:
        import torch
        p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        return p if os.path.exists(p) else None
    except Exception:
        return None


def format_score_lines(counter0, probs, labels):
    """bytes of the scoring writer's lines for pairs counter0 .. (kprn_format_score_lines; host-only)"""
    L = lib()
    probs = np.ascontiguousarray(probs, np.float32)
    labels = np.ascontiguousarray(labels, np.float32)
    n = int(probs.shape[0])
    cap = 32 * n + 64
    w = C.c_int64()
    while True:
        buf = C.create_string_buffer(cap)
        rc = L.kprn_format_score_lines(C.c_int64(int(counter0)), _fp(probs), _fp(labels), C.c_int64(n), buf, C.c_int64(cap), C.byref(w))
        if rc == 0:
            return buf.raw[:w.value]
        if w.value >= 0:
            raise KprnError(rc, "kprn_format_score_lines failed")
        cap = -w.value + 64


_lib = None


def lib():
    """Loads libkprn.so.  torch (if importable) is imported first so that both share ONE HIP runtime
    (torch bundles libamdhip64.so.7; the loader de-duplicates by SONAME)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m kprn_amd.build` "
                          "(hipcc --offload-arch=gfx950). kprn_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    L = C.CDLL(LIB_PATH)
    L.kprn_last_error.restype = C.c_char_p
    L.kprn_last_error.argtypes = [C.c_void_p]
    L.kprn_version.restype = C.c_char_p
    L.kprn_destroy.restype = None
    L.kprn_destroy.argtypes = [C.c_void_p]
    L.kprn_batch_destroy.restype = None
    L.kprn_batch_destroy.argtypes = [C.c_void_p, C.c_void_p]
    L.kprn_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.kprn_host_free.argtypes = [C.c_void_p, C.c_void_p]
    _lib = L
    return L


_SLOW_FP = os.environ.get("KPRN_FFI_NUMPY_POINTERS") == "1"   # (measurement: ndarray.ctypes for every pointer, as before round 6)


def _fp(a):
    """the array's address as a ctypes pointer.  `ndarray.ctypes` builds a helper object per access (2.8 us a call: two of them were 4 % of a 128-pair
    kprn_train_step); the buffer protocol gives the same address in 0.9 us.  Empty, read-only and non-contiguous arrays take numpy's route."""
    if a is None:
        return None
    if _SLOW_FP:
        return a.ctypes.data_as(C.c_void_p)
    try:
        return C.c_void_p(C.addressof(C.c_char.from_buffer(a)))
    except (TypeError, ValueError, BufferError):
        return a.ctypes.data_as(C.c_void_p)


def make_opt(method=1, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, lr_decay=0.0, regularize=0, use_grad_clip=1,
             grad_clip_norm=5.0, l2=1e-3, bce_literal=0, entity_update=0):
    return Opt(method, lr, beta1, beta2, eps, lr_decay, regularize, use_grad_clip, grad_clip_norm, l2, bce_literal, entity_update)


class Batch:
    """A minibatch resident in HBM (BatcherFileList:populateGPUTensor, BatcherFileList.lua:78-96)."""

    def __init__(self, engine, idx, labels=None, feed=False):
        """feed=False: kprn_batch_create (ready on return).  feed=True: a slot for Batch.refill -- the upload and the index build
        run on the engine's feed stream, under whatever is queued next (kprn_batch_feed_async)."""
        self._attach(engine)
        if feed:
            self.refill(idx, labels)
            return
        idx, lab = self._check(idx, labels)
        engine._ck(engine.L.kprn_batch_create(engine.h, _fp(idx), _fp(lab), self.B, self.P, self.T, self.F, C.byref(self.ptr)))

    def _attach(self, engine):
        self.engine = engine
        self.ptr = C.c_void_p()
        self._src = None
        engine._batches.add(self)   # Engine.close() releases the batches still alive before the handle goes

    def _check(self, idx, labels):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        if idx.ndim != 4:
            raise KprnError(E_ARG, "idx must be [B,P,T,F]")
        self.B, self.P, self.T, self.F = (int(x) for x in idx.shape)
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.float32)
        if lab is not None and lab.shape != (self.B,):
            raise KprnError(E_ARG, "labels must be [B]")
        self.has_labels = lab is not None
        return idx, lab

    @classmethod
    def reserve(cls, engine, max_pairs, max_paths, T, F, with_labels=True):
        """an empty feed slot sized for the largest minibatch it will hold (kprn_batch_slot_reserve)"""
        self = cls.__new__(cls)
        self._attach(engine)
        self.B = self.P = 0
        self.T, self.F, self.has_labels = T, F, with_labels
        engine._ck(engine.L.kprn_batch_slot_reserve(engine.h, C.byref(self.ptr), int(max_pairs), C.c_int64(int(max_paths)), int(T), int(F), int(bool(with_labels))))
        return self

    def refill(self, idx, labels=None):
        """streaming feed: new contents for this slot, asynchronously.  idx / labels should live in page-locked memory
        (Engine.host_array) and must stay unchanged until the slot is first used."""
        idx, lab = self._check(idx, labels)
        self._src = (idx, lab)  # keep the host buffers alive until the copy has run
        self.engine._ck(self.engine.L.kprn_batch_feed_async(self.engine.h, C.byref(self.ptr), _fp(idx), _fp(lab), self.B, self.P, self.T, self.F))
        return self

    def refill_rows(self, data, labels, rows):
        """shuffled streaming feed (kprn_batch_feed_rows_async): pair i of the minibatch = row rows[i] of the file's arrays
        data [n,P,T,F] int32 / labels [n] float32, gathered by the engine's worker threads -- no host-side copy here."""
        if data.dtype != np.int32 or not data.flags.c_contiguous or data.ndim != 4:
            raise KprnError(E_ARG, "data must be a C-contiguous int32 [n,P,T,F] array")
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.float32)
        self.B, (_, self.P, self.T, self.F) = int(rows.shape[0]), (int(x) for x in data.shape)
        self.has_labels = lab is not None
        self._src = (data, lab, rows)   # alive until the worker threads have read them
        self.engine._ck(self.engine.L.kprn_batch_feed_rows_async(self.engine.h, C.byref(self.ptr), _fp(data), _fp(lab), C.c_int64(int(data.shape[0])), _fp(rows),
                                                                   self.B, self.P, self.T, self.F))
        return self

    @property
    def n_paths(self):
        return self.B * self.P

    @property
    def n_uniq(self):
        """distinct entity rows of the batch (the rows one training step touches)"""
        n = C.c_int32()
        self.engine._ck(self.engine.L.kprn_batch_distinct_rows(self.engine.h, self.ptr, C.byref(n)))
        return int(n.value)

    @property
    def executed_steps(self):
        """(path, step) positions a pass executes: B*P*T less the identical leading steps that are run once per batch"""
        n = C.c_int64()
        self.engine._ck(self.engine.L.kprn_batch_executed_steps(self.engine.h, self.ptr, C.byref(n)))
        return int(n.value)

    @property
    def handover_stats(self):
        """time-split tile hand-over on this batch: (pairs, steps moved, longest workgroup in half steps without, with)"""
        out = (C.c_int64 * 4)()
        self.engine._ck(self.engine.L.kprn_batch_handover_stats(self.engine.h, self.ptr, out))
        return tuple(int(v) for v in out)

    def free(self):
        if self.ptr:
            self.engine.L.kprn_batch_destroy(self.engine.h, self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            if self.engine.h:
                self.free()
        except Exception:
            pass


STREAM_LEGACY_DEFAULT = C.c_void_p(-1).value   # kprn_config.stream: queue on the legacy default (null) stream (KPRN_STREAM_LEGACY_DEFAULT)


class Engine:
    """Thin object wrapper over one kprn_handle.  stream: None = the engine creates its own; a hipStream_t handle; or
    STREAM_LEGACY_DEFAULT for the null stream (whose handle, 0, would otherwise read as "create your own")."""

    def __init__(self, Vt, Ve, Vr, dt, de, dr, H, L=1, F=3, num_types=1, C_=46, reducer=2, K=5, rnn_type=0, device_id=0,
                 rank=0, world=1, param_init=0.1, seed=12345, stream=None, use_relu=1, rnn_init=0, compute_dtype=0):
        self.L = lib()
        self._batches = weakref.WeakSet()
        self.cfg = Config(Vt, Ve, Vr, dt, de, dr, F, num_types, H, L, C_, rnn_type, use_relu, rnn_init, compute_dtype, reducer, K, device_id, rank, world,
                          param_init, seed, stream)
        self.h = C.c_void_p()
        rc = self.L.kprn_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            self.h = C.c_void_p()
            raise KprnError(rc, (self.L.kprn_last_error(None) or b"").decode())
        n = C.c_int64()
        self._ck(self.L.kprn_num_params(self.h, C.byref(n)))
        self.n_params = int(n.value)
        self.D = dt + de + dr
        if os.environ.get("KPRN_SMALL_TILES") is not None:   # (tests: the 64-path tiles + identical-prefix plan at small sizes too)
            self.set_option("small_tiles", os.environ["KPRN_SMALL_TILES"])

    # -- plumbing ------------------------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            raise KprnError(rc, (self.L.kprn_last_error(self.h) or b"").decode())

    def close(self):
        if self.h:
            for b in list(self._batches):   # (device blocks, page-locked images and events of batches the caller still holds)
                b.free()
            for p in getattr(self, "_pinned", []):
                self.L.kprn_host_free(self.h, p)
            self._pinned = []
            self.L.kprn_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def layout(self):
        """name -> (flat offset, shape) in nn.Module:getParameters() order."""
        c = self.cfg
        out, off = {}, 0
        items = [("type_emb", (c.Vt, c.dt)), ("entity_emb", (c.Ve, c.de)), ("relation_emb", (c.Vr, c.dr))]
        for i in range(c.L):
            din = self.D if i == 0 else c.H
            if c.rnn_type == 1:  # nn.Recurrence: i2h / h2h nn.Linear (OneModel.lua:231-232)
                items += [(f"rnn{i + 1}.i2h.weight", (c.H, din)), (f"rnn{i + 1}.i2h.bias", (c.H,)),
                          (f"rnn{i + 1}.h2h.weight", (c.H, c.H)), (f"rnn{i + 1}.h2h.bias", (c.H,))]
            elif c.rnn_type == 2:  # nn.GRU: gates (r, z) + candidate maps
                items += [(f"gru{i + 1}.i2g.weight", (2 * c.H, din)), (f"gru{i + 1}.i2g.bias", (2 * c.H,)),
                          (f"gru{i + 1}.o2g.weight", (2 * c.H, c.H)), (f"gru{i + 1}.c_i2h.weight", (c.H, din)),
                          (f"gru{i + 1}.c_i2h.bias", (c.H,)), (f"gru{i + 1}.c_h2h.weight", (c.H, c.H))]
            else:
                items += [(f"lstm{i + 1}.i2g.weight", (4 * c.H, din)), (f"lstm{i + 1}.i2g.bias", (4 * c.H,)),
                          (f"lstm{i + 1}.o2g.weight", (4 * c.H, c.H))]
        items += [("out.weight", (c.C, c.H)), ("out.bias", (c.C,))]
        for nm, shp in items:
            out[nm] = (off, shp)
            off += int(np.prod(shp))
        assert off == self.n_params
        return out

    # -- parameters ----------------------------------------------------------------------
    def _shape(self, name):
        lay = self.layout()
        if name not in lay:  # let the library produce its own error code / message
            self._ck(self.L.kprn_get_param(self.h, str(name).encode(), None, C.c_int64(0)))
            raise KprnError(E_ARG, f"unknown parameter name: {name}")
        return lay[name][1]

    def get_param(self, name):
        shp = self._shape(name)
        a = np.empty(shp, np.float32)
        self._ck(self.L.kprn_get_param(self.h, name.encode(), _fp(a), C.c_int64(a.size)))
        return a

    def set_param(self, name, value):
        a = np.ascontiguousarray(value, np.float32)
        self._ck(self.L.kprn_set_param(self.h, name.encode(), _fp(a), C.c_int64(a.size)))

    def get_param_rows(self, name, rows):
        """rows (0-based) of one parameter tensor: [len(rows), cols]"""
        rows = np.ascontiguousarray(rows, np.int64)
        out = np.empty((len(rows), self._shape(name)[1] if len(self._shape(name)) > 1 else 1), np.float32)
        self._ck(self.L.kprn_get_param_rows(self.h, name.encode(), _fp(rows), C.c_int64(len(rows)), _fp(out)))
        return out

    def set_param_rows(self, name, rows, values):
        rows = np.ascontiguousarray(rows, np.int64)
        values = np.ascontiguousarray(values, np.float32)
        self._ck(self.L.kprn_set_param_rows(self.h, name.encode(), _fp(rows), C.c_int64(len(rows)), _fp(values)))

    def get_grad(self, name):
        shp = self._shape(name)
        a = np.empty(shp, np.float32)
        self._ck(self.L.kprn_get_grad(self.h, name.encode(), _fp(a), C.c_int64(a.size)))
        return a

    def get_flat_params(self):
        a = np.empty(self.n_params, np.float32)
        self._ck(self.L.kprn_get_flat_params(self.h, _fp(a), C.c_int64(a.size)))
        return a

    def set_flat_params(self, theta):
        a = np.ascontiguousarray(theta, np.float32)
        self._ck(self.L.kprn_set_flat_params(self.h, _fp(a), C.c_int64(a.size)))

    def get_flat_grads(self):
        a = np.empty(self.n_params, np.float32)
        self._ck(self.L.kprn_get_flat_grads(self.h, _fp(a), C.c_int64(a.size)))
        return a

    def get_flat_opt_state(self, slot):
        a = np.empty(self.n_params, np.float32)
        self._ck(self.L.kprn_get_flat_opt_state(self.h, int(slot), _fp(a), C.c_int64(a.size)))
        return a

    def zero_pad_tokens(self):
        self._ck(self.L.kprn_zero_pad_tokens(self.h))

    # -- scoring -------------------------------------------------------------------------
    def batch(self, idx, labels=None):
        return Batch(self, idx, labels)

    def feed(self, idx, labels=None, slot=None):
        """streaming feed (BatcherFileList.lua:53-96): fills `slot` (or a new one) on the feed stream and returns at once"""
        if slot is None:
            return Batch(self, idx, labels, feed=True)
        return slot.refill(idx, labels)

    def feed_rows(self, data, labels, rows, slot=None):
        """the feed for a shuffled order: rows of the file's arrays, gathered inside the engine (Batch.refill_rows)"""
        if slot is None:
            slot = Batch.__new__(Batch)
            slot._attach(self)
        return slot.refill_rows(data, labels, rows)

    def host_array(self, shape, dtype=np.int32):
        """numpy array in page-locked host memory (kprn_host_alloc): the staging buffers of the streaming feed"""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._ck(self.L.kprn_host_alloc(self.h, C.c_size_t(max(n, 1)), C.byref(p)))
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p)
        return arr

    def forward(self, batch, class_id=1, want=("probs",)):
        """want: any of probs [B], all_probs [B,C], pooled [B,C], path_scores [B*P,C]."""
        if not isinstance(batch, Batch):
            batch = Batch(self, batch)
        c = self.cfg
        bufs = {"probs": np.empty(batch.B, np.float32) if "probs" in want else None,
                "all_probs": np.empty((batch.B, c.C), np.float32) if "all_probs" in want else None,
                "pooled": np.empty((batch.B, c.C), np.float32) if "pooled" in want else None,
                "path_scores": np.empty((batch.n_paths, c.C), np.float32) if "path_scores" in want else None}
        self._ck(self.L.kprn_forward_batch(self.h, batch.ptr, int(class_id), _fp(bufs["probs"]), _fp(bufs["all_probs"]),
                                           _fp(bufs["pooled"]), _fp(bufs["path_scores"])))
        return {k: v for k, v in bufs.items() if v is not None}

    def forward_host(self, idx, class_id=1, want_all=True):
        """the host-pointer entry point kprn_forward (one H2D copy per call).  want_all=False: only the selected class's probabilities come back --
        what test_from_checkpoint.lua:82,109 reads (its model ends in nn.Select(2, 1)); returns (probs, None)."""
        idx = np.ascontiguousarray(idx, np.int32)
        B, P, T, F = idx.shape
        probs = np.empty(B, np.float32)
        allp = np.empty((B, self.cfg.C), np.float32) if want_all else None
        self._ck(self.L.kprn_forward(self.h, _fp(idx), B, P, T, F, int(class_id), _fp(probs), _fp(allp)))
        return probs, allp

    def forward_async(self, batch, class_id=1):
        self._ck(self.L.kprn_forward_batch_async(self.h, batch.ptr, int(class_id)))

    def forward_async_rest(self):
        """second part of a split scoring pass (set_option("score_split", f))"""
        self._ck(self.L.kprn_forward_batch_async_rest(self.h))

    def read_probs(self, B):
        a = np.empty(B, np.float32)
        self._ck(self.L.kprn_read_probs(self.h, _fp(a), int(B)))
        return a

    def embed(self, idx):
        idx = np.ascontiguousarray(idx, np.int32)
        T, F = idx.shape[-2], idx.shape[-1]
        N = idx.size // (T * F)
        x = np.empty((N, T, self.D), np.float32)
        self._ck(self.L.kprn_embed(self.h, _fp(idx), C.c_int64(N), T, F, _fp(x)))
        return x

    # -- training ------------------------------------------------------------------------
    def backward(self, batch, class_id=1, bce_literal=False, inv_batch=0.0, want_loss=True):
        loss = C.c_float()
        self._ck(self.L.kprn_backward_batch(self.h, batch.ptr, int(class_id), int(bool(bce_literal)), C.c_float(inv_batch),
                                            C.byref(loss) if want_loss else None))
        return float(loss.value) if want_loss else None

    def apply_update(self, opt):
        self._ck(self.L.kprn_apply_update(self.h, C.byref(opt)))

    def train_step(self, batch, opt, class_id=1, want_loss=True):
        loss = C.c_float()
        self._ck(self.L.kprn_train_step_batch(self.h, batch.ptr, int(class_id), C.byref(opt), C.byref(loss) if want_loss else None))
        return float(loss.value) if want_loss else None

    def train_step_host(self, idx, labels, opt, class_id=1):
        idx = np.ascontiguousarray(idx, np.int32)
        labels = np.ascontiguousarray(labels, np.float32)
        B, P, T, F = idx.shape
        loss = C.c_float()
        self._ck(self.L.kprn_train_step(self.h, _fp(idx), B, P, T, F, _fp(labels), int(class_id), C.byref(opt), C.byref(loss)))
        return float(loss.value)

    def read_loss(self):
        loss = C.c_float()
        self._ck(self.L.kprn_read_loss(self.h, C.byref(loss)))
        return float(loss.value)

    def loss_sum(self, reset=True):
        """(sum of the losses, number of steps) accumulated on the device since the last reset (option loss_accumulate = 1)"""
        v, n = C.c_float(), C.c_int32()
        self._ck(self.L.kprn_read_loss_sum(self.h, C.byref(v), C.byref(n), int(bool(reset))))
        return float(v.value), int(n.value)

    def sync(self):
        self._ck(self.L.kprn_sync(self.h))

    # -- data parallel -------------------------------------------------------------------
    def dense_grad_buffer(self):
        p, n = C.c_void_p(), C.c_int64()
        self._ck(self.L.kprn_dense_grad_buffer(self.h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def sparse_grad_capacity(self):
        n = C.c_int32()
        self._ck(self.L.kprn_sparse_grad_capacity(self.h, C.byref(n)))
        return int(n.value)

    def sparse_grad_pack(self, capacity):
        """-> (device pointer of the packed buffer, its length in 32-bit words)"""
        buf, n = C.c_void_p(), C.c_int64()
        self._ck(self.L.kprn_sparse_grad_pack(self.h, int(capacity), C.byref(buf), C.byref(n)))
        return int(buf.value), int(n.value)

    def sparse_grad_merge(self, all_ptr, world, capacity):
        self._ck(self.L.kprn_sparse_grad_merge(self.h, C.c_void_p(all_ptr), int(world), int(capacity)))

    # ---- the exchange issued by the engine over RCCL (kprn_dp_*) ----
    def dp_init(self, id128, rank, world, rccl_path=None):
        buf = (C.c_char * 128).from_buffer_copy(bytes(id128))
        self._ck(self.L.kprn_dp_init(self.h, rccl_path.encode() if rccl_path else None, buf, int(rank), int(world)))

    def dp_exchange_begin(self, capacity):
        self._ck(self.L.kprn_dp_exchange_begin(self.h, int(capacity)))

    def dp_exchange_finish(self, opt):
        self._ck(self.L.kprn_dp_exchange_finish(self.h, C.byref(opt)))

    def dp_comm_size(self):
        n = C.c_int32()
        self._ck(self.L.kprn_dp_comm_size(self.h, C.byref(n)))
        return n.value

    def dp_shutdown(self):
        self._ck(self.L.kprn_dp_shutdown(self.h))

    def stream(self):
        p = C.c_void_p()
        self._ck(self.L.kprn_stream(self.h, C.byref(p)))
        return p.value or 0

    # -- checkpoints / measurement ---------------------------------------------------------
    def save(self, path):
        self._ck(self.L.kprn_save(self.h, os.fsencode(path)))

    def load(self, path):
        self._ck(self.L.kprn_load(self.h, os.fsencode(path)))

    def profile(self, on=True):
        self._ck(self.L.kprn_profile_enable(self.h, int(bool(on))))

    def profile_reset(self):
        self._ck(self.L.kprn_profile_reset(self.h))

    def profile_get(self):
        n = C.c_int32()
        arr = (ProfEntry * 128)()
        self._ck(self.L.kprn_profile_get(self.h, arr, 128, C.byref(n)))
        return {arr[i].name.decode(): (arr[i].total_ms, int(arr[i].launches)) for i in range(min(n.value, 128))}

    def set_option(self, key, value):
        self._ck(self.L.kprn_set_option(self.h, key.encode(), value.encode()))
