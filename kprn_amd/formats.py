"""On-disk formats either side of the hot path.

1. ``.int`` text files written by release/songPathRnn/data/movie_data_format.py and parsed by
   release/songPathRnn/data/int2torch.lua:27-56:
       <label> TAB <path>;<path>;...      path = <step> <step> ...      step = f1,f2,...,fF
   (0-based ids; int2torch -addOne 1 shifts the DATA, not the labels, to 1-based: lines 60-63.)

2. Torch7 binary serialisation of ``{labels=DoubleTensor[N], data=DoubleTensor[N,P,T,F],
   classId=number}`` (int2torch.lua:65-70, insertClassLabels.lua:12-17; consumed by
   model/batcher/Batcher.lua:11-28).  The reference tree ships NO .torch file, so this
   reader/writer follows the published Torch7 File format [restated from torch7/File.lua +
   generic/Tensor.c, unpinned -- SURVEY.md 8f N2] and is verified by round trips only:
     object   := int32 type ; type 0 nil | 1 number(f64) | 2 string(int32 len, bytes) |
                 3 table | 4 torch object | 5 boolean(int32)
     table    := int32 ref-index ; (first time) int32 n ; n x (key object, value object)
     torch    := int32 ref-index ; (first time) string "V 1" ; string class name ; payload
     Tensor   := int32 nDim ; int64 size[nDim] ; int64 stride[nDim] ; int64 storageOffset(1-based) ;
                 storage object (torch.<T>Storage: int64 n ; raw elements)

3. ``.npz`` -- the native container of this engine (same three fields, data already int32).
"""
import struct

import numpy as np

# ------------------------------------------------------------------------------------------
# .int text


def read_int_file(path, add_one=True):
    """-> (labels float64 [N], data int32 [N,P,T,F]); mirrors int2torch.lua -tokenFeatures 1."""
    labels, rows = [], []
    with open(path, "r") as f:
        for ln, line in enumerate(f, 1):
            line = line.rstrip("\n")
            if not line:
                continue
            fields = line.split("\t")
            if len(fields) < 2:
                raise ValueError(f"{path}:{ln}: expected '<label>\\t<paths>'")
            labels.append(float(fields[0]))
            paths = []
            for p in fields[1].split(";"):
                paths.append([[int(x) for x in tok.split(",")] for tok in p.split(" ") if tok != ""])
            rows.append(paths)
    if not rows:
        raise ValueError(f"{path}: empty file")
    P, T, F = len(rows[0]), len(rows[0][0]), len(rows[0][0][0])
    for i, r in enumerate(rows):
        # Util:table2tensor asserts regular sizes (util/Util.lua:121-140)
        if len(r) != P or any(len(p) != T for p in r) or any(len(s) != F for p in r for s in p):
            raise ValueError(f"{path}: row {i + 1} is not {P}x{T}x{F}: input tensor is expected to have the same "
                             "number of elements in each dim")
    data = np.asarray(rows, dtype=np.int64)
    if add_one:
        data = data + 1
    if data.min() < 0 or data.max() >= 2 ** 31:
        raise ValueError(f"{path}: id out of int32 range")
    return np.asarray(labels, dtype=np.float64), data.astype(np.int32)


def write_int_file(path, labels, data, sub_one=True):
    data = np.asarray(data)
    if sub_one:
        data = data - 1
    with open(path, "w") as f:
        for lab, pairs in zip(labels, data):
            lab_s = str(int(lab)) if float(lab).is_integer() else repr(float(lab))
            f.write(lab_s + "\t" + ";".join(" ".join(",".join(str(int(v)) for v in step) for step in p) for p in pairs) + "\n")


# ------------------------------------------------------------------------------------------
# Torch7 binary

_T_NIL, _T_NUMBER, _T_STRING, _T_TABLE, _T_TORCH, _T_BOOL = 0, 1, 2, 3, 4, 5
_STORAGE_DTYPES = {
    "torch.DoubleStorage": np.float64, "torch.FloatStorage": np.float32, "torch.LongStorage": np.int64,
    "torch.IntStorage": np.int32, "torch.ShortStorage": np.int16, "torch.ByteStorage": np.uint8, "torch.CharStorage": np.int8,
}
_TENSOR_OF = {np.dtype(np.float64): "Double", np.dtype(np.float32): "Float", np.dtype(np.int64): "Long", np.dtype(np.int32): "Int"}


class _T7Reader:
    def __init__(self, buf):
        self.b, self.o, self.objs, self.tensor_ids = buf, 0, {}, {}

    def _rd(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v[0] if len(v) == 1 else v

    def _string(self):
        n = self._rd("i")
        s = bytes(self.b[self.o:self.o + n]).decode("latin-1")
        self.o += n
        return s

    def obj(self):
        t = self._rd("i")
        if t == _T_NIL:
            return None
        if t == _T_NUMBER:
            return self._rd("d")
        if t == _T_STRING:
            return self._string()
        if t == _T_BOOL:
            return self._rd("i") != 0
        if t == _T_TABLE:
            ref = self._rd("i")
            if ref in self.objs:
                return self.objs[ref]
            out = {}
            self.objs[ref] = out
            for _ in range(self._rd("i")):
                k = self.obj()
                out[k] = self.obj()
            return out
        if t == _T_TORCH:
            ref = self._rd("i")
            if ref in self.objs:
                return self.objs[ref]
            version = self._string()
            cls = self._string() if version.startswith("V ") else version
            if cls in _STORAGE_DTYPES:
                n = self._rd("q")
                dt = np.dtype(_STORAGE_DTYPES[cls])
                arr = np.frombuffer(self.b, dtype=dt, count=n, offset=self.o).copy()
                self.o += n * dt.itemsize
                self.objs[ref] = arr
                return arr
            if cls.endswith("Tensor"):
                nd = self._rd("i")
                size = [self._rd("q") for _ in range(nd)]
                stride = [self._rd("q") for _ in range(nd)]
                off = self._rd("q") - 1
                storage = self.obj()
                if storage is None or nd == 0:
                    ten = np.zeros(size, dtype=np.float64)
                else:
                    ten = np.lib.stride_tricks.as_strided(storage[off:], shape=size, strides=[s * storage.itemsize for s in stride]).copy()
                self.objs[ref] = ten
                # identity of the view (storage object, offset, shape): shared-parameter clones of the rnn library
                # (AbstractRecurrent.sharedClones) point at the same storage and are recognised by it
                self.tensor_ids[id(ten)] = (id(storage) if storage is not None else 0, off, tuple(size))
                return ten
            # any other torch class (nn.* modules ...): torch.File writes the object's fields as one table
            mod = T7Object(cls)
            self.objs[ref] = mod
            fields = self.obj()
            if isinstance(fields, dict):
                mod.update(fields)
            return mod
        raise ValueError(f"unsupported Torch7 type tag {t}")


class T7Object(dict):
    """a non-tensor torch object (e.g. an nn module): its fields, plus the Lua class name"""

    def __init__(self, cls):
        super().__init__()
        self.torch_class = cls


def t7_load(path, with_ids=False):
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    r = _T7Reader(buf)
    o = r.obj()
    return (o, r.tensor_ids) if with_ids else o


class _T7Writer:
    def __init__(self):
        self.parts, self.n, self.seen = [], 0, {}

    def _w(self, fmt, *v):
        self.parts.append(struct.pack("<" + fmt, *v))

    def _string(self, s):
        b = s.encode("latin-1")
        self._w("i", len(b))
        self.parts.append(b)

    def obj(self, v):
        if v is None:
            self._w("i", _T_NIL)
        elif isinstance(v, bool):
            self._w("ii", _T_BOOL, int(v))
        elif isinstance(v, (int, float, np.integer, np.floating)):
            self._w("id", _T_NUMBER, float(v))
        elif isinstance(v, str):
            self._w("i", _T_STRING)
            self._string(v)
        elif isinstance(v, T7Object):
            if id(v) in self.seen:
                self._w("ii", _T_TORCH, self.seen[id(v)])
                return
            self.n += 1
            self.seen[id(v)] = self.n
            self._w("ii", _T_TORCH, self.n)
            self._string("V 1")
            self._string(v.torch_class)
            self.obj(dict(v))
        elif isinstance(v, dict):
            self.n += 1
            self._w("iii", _T_TABLE, self.n, len(v))
            for k, x in v.items():
                self.obj(k)
                self.obj(x)
        elif isinstance(v, np.ndarray):
            a = np.ascontiguousarray(v)
            name = _TENSOR_OF[a.dtype]
            self.n += 1
            self._w("ii", _T_TORCH, self.n)
            self._string("V 1")
            self._string(f"torch.{name}Tensor")
            self._w("i", a.ndim)
            for s in a.shape:
                self._w("q", s)
            st = [x // a.itemsize for x in a.strides]
            for s in st:
                self._w("q", s)
            self._w("q", 1)
            base = a.base if a.base is not None else a
            if id(base) in self.seen and a.base is not None:  # a second view of a storage already written: back-reference
                self._w("ii", _T_TORCH, self.seen[id(base)])
            else:
                self.n += 1
                self.seen[id(v)] = self.n
                self._w("ii", _T_TORCH, self.n)
                self._string("V 1")
                self._string(f"torch.{name}Storage")
                self._w("q", a.size)
                self.parts.append(a.tobytes())
        else:
            raise TypeError(type(v))


def t7_save(path, obj):
    w = _T7Writer()
    w.obj(obj)
    with open(path, "wb") as f:
        for p in w.parts:
            f.write(p)


# ------------------------------------------------------------------------------------------
# path-set files ({labels, data, classId}) in any of the three containers


def _npz_member_memmap(path, name):
    """read-only memory map of an uncompressed .npy member of a .npz archive (None when it is compressed or anything looks
    unusual): a multi-GB path file then costs no load time and no second copy in RAM -- the engine's feed threads read the rows
    they need straight from the page cache"""
    import zipfile
    try:
        with zipfile.ZipFile(path) as zf:
            info = zf.getinfo(name + ".npy")
            if info.compress_type != zipfile.ZIP_STORED:
                return None
            with open(path, "rb") as f:
                f.seek(info.header_offset)
                hdr = f.read(30)
                if hdr[:4] != b"PK\x03\x04":
                    return None
                n_name, n_extra = int.from_bytes(hdr[26:28], "little"), int.from_bytes(hdr[28:30], "little")
                start = info.header_offset + 30 + n_name + n_extra
                f.seek(start)
                version = np.lib.format.read_magic(f)
                shape, fortran, dtype = (np.lib.format.read_array_header_1_0(f) if version == (1, 0) else np.lib.format.read_array_header_2_0(f))
                if fortran or dtype.hasobject:
                    return None
                offset = f.tell()
            if offset + int(np.prod(shape)) * dtype.itemsize > start + info.file_size:
                return None
            return np.memmap(path, dtype=dtype, mode="r", offset=offset, shape=tuple(shape))
    except Exception:
        return None


def load_path_file(path, check_ids=True):
    """-> (labels float32 [N], data int32 [N,P,T,F] 1-based, classId int).
    check_ids=False: skip the id >= 1 pass over an int32 file (a memory-mapped .npz is otherwise never read on the host) -- only for
    callers that hand the ids to the engine, which checks every id against its vocabulary (KPRN_E_INDEX)."""
    if path.endswith(".npz"):
        z = np.load(path)
        labels, cid = z["labels"], int(z["classId"]) if "classId" in z else 1
        data = _npz_member_memmap(path, "data")   # (np.savez stores members uncompressed: map the ids instead of copying them)
        if data is None:
            data = z["data"]
    elif path.endswith(".int"):
        labels, data = read_int_file(path, add_one=True)
        cid = 1  # movie_data_format.sh:39  insertClassLabels -classLabel 1
    else:
        t = t7_load(path)
        if not isinstance(t, dict) or "labels" not in t or "data" not in t:
            raise ValueError(f"{path}: not a {{labels, data, classId}} table")
        labels, data = t["labels"], t["data"]
        cid = int(t.get("classId", 1))
    data = np.asarray(data)
    if data.ndim != 4:
        raise ValueError(f"{path}: data must be [N,P,T,F], got {data.shape}")
    if not np.issubdtype(data.dtype, np.integer):
        # .torch files hold float64 ids (util/Util.lua:129); 20M-entity vocabularies exceed 2^24, so go
        # through int64, never float32
        if np.any(data != np.rint(data)):
            raise ValueError(f"{path}: non-integer id")
        data = data.astype(np.int64)
    if data.dtype != np.int32 and data.size and (data.min() < 1 or data.max() >= 2 ** 31):
        raise ValueError(f"{path}: id outside 1..2^31-1")
    if data.dtype == np.int32 and data.size and check_ids and int(data.min()) < 1:   # (the upper bound is the engine's: it knows the vocabularies)
        raise ValueError(f"{path}: id < 1")
    return np.asarray(labels, dtype=np.float32).reshape(-1), np.ascontiguousarray(data, dtype=np.int32), cid


def save_path_file(path, labels, data, class_id=1):
    labels = np.asarray(labels)
    data = np.asarray(data)
    if path.endswith(".npz"):
        np.savez(path, labels=labels.astype(np.float32), data=data.astype(np.int32), classId=np.int32(class_id))
    elif path.endswith(".int"):
        write_int_file(path, labels, data, sub_one=True)
    else:
        t7_save(path, {"labels": labels.astype(np.float64), "data": data.astype(np.float64), "classId": float(class_id)})


# ------------------------------------------------------------------------------------------
# reference checkpoints: torch.save{embeddingLayer=..., predictor_net=...} (OneModel.lua:392-408), read back by
# test_from_checkpoint.lua:68-75 and -initModel (OneModel.lua:277-282).  [restated from torch7 File.lua / nn / rnn
# object layouts, unpinned: the tree ships no checkpoint -- SURVEY.md 8f N4d; verified on synthetic object graphs only]


def _walk_modules(o, seen, out):
    """depth-first over an nn object graph in module order (`modules[1..n]`, then named sub-modules), collecting
    every parameter-holding leaf once (shared-parameter step clones are skipped by object identity)."""
    if isinstance(o, T7Object):
        if id(o) in seen:
            return
        seen.add(id(o))
        cls = o.torch_class
        if cls in ("nn.LookupTable", "nn.Linear", "nn.LinearNoBias") and "weight" in o:
            out.append(o)
            return
        mods = o.get("modules")
        if isinstance(mods, dict):
            for k in sorted(k for k in mods if isinstance(k, (int, float))):
                _walk_modules(mods[k], seen, out)
        for k in ("module", "i2g", "o2g", "recurrentModule", "inputModule", "initialModule", "feedbackModule"):
            if k in o:
                _walk_modules(o[k], seen, out)
    elif isinstance(o, dict):
        for k in sorted(k for k in o if isinstance(k, (int, float))):
            _walk_modules(o[k], seen, out)


def checkpoint_params(path_or_obj, num_layers=None):
    """-> dict reference-parameter-name -> float32 array, in the engine's names (`type_emb`, `entity_emb`,
    `relation_emb`, `lstm{l}.i2g.weight|bias`, `lstm{l}.o2g.weight` or `rnn{l}.i2h.*|h2h.*`, `out.weight|bias`),
    extracted from a {embeddingLayer, predictor_net} checkpoint.  Feed them to Engine.set_param / kprn_set_param."""
    ck, ids = (t7_load(path_or_obj, with_ids=True) if isinstance(path_or_obj, str) else (path_or_obj, {}))
    if not isinstance(ck, dict) or "embeddingLayer" not in ck or "predictor_net" not in ck:
        raise ValueError("not a {embeddingLayer, predictor_net} checkpoint (OneModel.lua:396-400)")
    emb, pred = [], []
    _walk_modules(ck["embeddingLayer"], set(), emb)
    _walk_modules(ck["predictor_net"], set(), pred)
    # shared clones that were serialised as separate objects still share storages: keep the first of each weight view
    def dedupe(mods):
        keep, seen_w = [], set()
        for m in mods:
            key = ids.get(id(m["weight"]), id(m["weight"]))
            if key in seen_w:
                continue
            seen_w.add(key)
            keep.append(m)
        return keep
    emb, pred = dedupe(emb), dedupe(pred)
    tables = [m for m in emb if m.torch_class == "nn.LookupTable"]
    if len(tables) != 3:
        raise ValueError(f"expected 3 lookup tables (type, entity, relation: FeatureEmbedding.lua:118), found {len(tables)}")
    out = {"type_emb": tables[0]["weight"], "entity_emb": tables[1]["weight"], "relation_emb": tables[2]["weight"]}
    lin = [m for m in pred if m.torch_class in ("nn.Linear", "nn.LinearNoBias")]
    if not lin:
        raise ValueError("no nn.Linear in predictor_net")
    head, body = lin[-1], lin[:-1]
    out["out.weight"], out["out.bias"] = head["weight"], head["bias"]
    # nn.GRU (OneModel.lua:237-238) contributes FOUR maps per layer -- i2g (Linear, 2H rows), o2g (LinearNoBias, 2H rows) and the
    # candidate's Linear(D, H) + LinearNoBias(H, H) -- nn.FastLSTM and nn.Recurrence two.  Told apart by the shapes: a GRU layer's
    # first map has twice the rows of its third, an LSTM's four times its hidden size, a Recurrence's exactly its hidden size.
    H = head["weight"].shape[1]
    def rows(m):
        return m["weight"].shape[0]
    if body and rows(body[0]) == 2 * H:
        if len(body) % 4 != 0:
            raise ValueError("nn.GRU layers must contribute four linear maps each (i2g, o2g, candidate i2h, candidate h2h)")
        L = len(body) // 4
        for l in range(L):
            a, b, c, d = body[4 * l:4 * l + 4]
            if not (rows(a) == rows(b) == 2 * H and rows(c) == rows(d) == H and "bias" in a and "bias" in c):
                raise ValueError(f"layer {l + 1}: not an nn.GRU parameter set")
            out[f"gru{l + 1}.i2g.weight"], out[f"gru{l + 1}.i2g.bias"], out[f"gru{l + 1}.o2g.weight"] = a["weight"], a["bias"], b["weight"]
            out[f"gru{l + 1}.c_i2h.weight"], out[f"gru{l + 1}.c_i2h.bias"], out[f"gru{l + 1}.c_h2h.weight"] = c["weight"], c["bias"], d["weight"]
    else:
        if len(body) % 2 != 0:
            raise ValueError("recurrent layers must contribute two linear maps each (i2g/o2g or i2h/h2h)")
        L = len(body) // 2
        for l in range(L):
            a, b = body[2 * l], body[2 * l + 1]
            if rows(a) == 4 * H and (b.torch_class == "nn.LinearNoBias" or "bias" not in b):  # nn.FastLSTM: i2g (Linear) + o2g (LinearNoBias)
                out[f"lstm{l + 1}.i2g.weight"], out[f"lstm{l + 1}.i2g.bias"], out[f"lstm{l + 1}.o2g.weight"] = a["weight"], a["bias"], b["weight"]
            elif rows(a) == H and "bias" in a and "bias" in b:                                 # nn.Recurrence: input2hidden + hidden2hidden
                out[f"rnn{l + 1}.i2h.weight"], out[f"rnn{l + 1}.i2h.bias"] = a["weight"], a["bias"]
                out[f"rnn{l + 1}.h2h.weight"], out[f"rnn{l + 1}.h2h.bias"] = b["weight"], b["bias"]
            else:
                raise ValueError(f"layer {l + 1}: neither an nn.FastLSTM nor an nn.Recurrence parameter set (rows {rows(a)}, H {H})")
    if num_layers is not None and L != num_layers:
        raise ValueError(f"checkpoint has {L} recurrent layers, expected {num_layers}")
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in out.items()}


# ---- writer: an engine-trained model as a reference checkpoint (OneModel.lua:392-400 torch.save{embeddingLayer, predictor_net}) ----
# Everything Torch7-side here is [from memory, unpinned] -- the tree ships no checkpoint and no Torch7.  What each assumption governs:
#   * object framing (type tags, "V 1" + class name, reference indices): _T7Writer.obj above, byte for byte what _T7Reader expects;
#   * an nn module serialises as a torch object whose payload is ONE table of its fields (torch.File:writeObject for non-tensor
#     classes) -- T7Object; the fields written are the ones nn's updateOutput reads: weight / bias (+ empty gradWeight / gradBias /
#     output / gradInput tensors, `_type`);
#   * container children live in the `modules` table under 1-based number keys (nn.Container); the Element-Research classes keep
#     theirs in named fields: nn.Sequencer.module, nn.FastLSTM.i2g / .o2g (+ recurrentModule), nn.Recurrence.recurrentModule, nn.GRU.i2g /
#     .o2g (+ recurrentModule) -- the names _walk_modules follows;
#   * predictor_net = Sequential{SplitTable(3), embeddingLayer, SplitTable(2), Sequencer(cell) x L, SelectTable(-1), Linear(H, 46)}
#     (OneModel.lua:223-275) and embeddingLayer = Sequential{ConcatTableNoGrad{type net, entity net, relation net}, JoinTable(3)}
#     (FeatureEmbedding.lua:112-121) share the embeddingLayer OBJECT (written once, back-referenced);
#   * tensors are torch.DoubleTensor (the CPU reference computes in float64), contiguous, storageOffset 1.
# A reviewer with a Torch7 install can falsify each line by torch.load()ing a file written here.


def _t(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _mod(cls, **fields):
    m = T7Object(cls)
    m.update({"_type": "torch.DoubleTensor", "output": np.zeros(0), "gradInput": np.zeros(0), "train": False})
    m.update(fields)
    return m


def _linear(w, b=None):
    if b is None:
        return _mod("nn.LinearNoBias", weight=_t(w), gradWeight=np.zeros(0))
    return _mod("nn.Linear", weight=_t(w), bias=_t(b), gradWeight=np.zeros(0), gradBias=np.zeros(0))


def _seq(*mods):
    return _mod("nn.Sequential", modules={float(i + 1): m for i, m in enumerate(mods)})


def write_checkpoint(path, params, num_entity_types=1, use_relu=1):
    """params: dict engine parameter name -> array (Engine.get_param for every name of Engine.layout()).  Writes the
    {embeddingLayer, predictor_net} table eval/test_from_checkpoint.lua:68 loads; checkpoint_params() reads it back."""
    g = lambda n: params[n]
    lut = lambda w: _mod("nn.LookupTable", weight=_t(w), gradWeight=np.zeros(0), shouldScaleGradByFreq=False)
    type_lut = lut(g("type_emb"))
    par = _mod("nn.ParallelTable", modules={float(i + 1): type_lut for i in range(max(1, num_entity_types))})   # shared weight (FeatureEmbedding.lua:44-48)
    type_net = _seq(_mod("nn.NarrowTable"), par, _mod("nn.CAddTable"))
    ent_net = _seq(_mod("nn.SelectTable"), lut(g("entity_emb")))
    rel_net = _seq(_mod("nn.SelectTable"), lut(g("relation_emb")))
    emb = _seq(_mod("nn.ConcatTableNoGrad", modules={1.0: type_net, 2.0: ent_net, 3.0: rel_net}), _mod("nn.JoinTable", dimension=3.0))
    layers = []
    l = 1
    while True:
        if f"lstm{l}.i2g.weight" in params:
            cell = _mod("nn.FastLSTM", i2g=_linear(g(f"lstm{l}.i2g.weight"), g(f"lstm{l}.i2g.bias")), o2g=_linear(g(f"lstm{l}.o2g.weight")))
        elif f"rnn{l}.i2h.weight" in params:
            rm = _seq(_mod("nn.ParallelTable", modules={1.0: _linear(g(f"rnn{l}.i2h.weight"), g(f"rnn{l}.i2h.bias")),
                                                        2.0: _linear(g(f"rnn{l}.h2h.weight"), g(f"rnn{l}.h2h.bias"))}),
                      _mod("nn.CAddTable"), _mod("nn.ReLU" if use_relu else "nn.Tanh"))
            cell = _mod("nn.Recurrence", recurrentModule=_mod("nn.MaskZero", module=rm, nInputDim=1.0))
        elif f"gru{l}.i2g.weight" in params:
            cand = _seq(_mod("nn.ParallelTable", modules={1.0: _linear(g(f"gru{l}.c_i2h.weight"), g(f"gru{l}.c_i2h.bias")),
                                                          2.0: _linear(g(f"gru{l}.c_h2h.weight"))}), _mod("nn.CAddTable"), _mod("nn.Tanh"))
            cell = _mod("nn.GRU", i2g=_linear(g(f"gru{l}.i2g.weight"), g(f"gru{l}.i2g.bias")), o2g=_linear(g(f"gru{l}.o2g.weight")), recurrentModule=cand)
        else:
            break
        layers.append(_mod("nn.Sequencer", module=cell))
        l += 1
    if not layers:
        raise ValueError("no recurrent layer parameters (lstm1.* / rnn1.* / gru1.*) in params")
    pred = _seq(_mod("nn.SplitTable", dimension=3.0), emb, _mod("nn.SplitTable", dimension=2.0), *layers, _mod("nn.SelectTable", index=-1.0),
                _linear(g("out.weight"), g("out.bias")))
    t7_save(path, {"embeddingLayer": emb, "predictor_net": pred})
