#!/usr/bin/env python3
"""End-to-end epoch through the reference-shaped host loop: path files on disk -> BatcherFileList (shuffled, as OneModel.lua:326)
-> MyOptimizer.train -> engine, on the C2 shapes (T=6, D=H=64, L=2, Ve=2 851 220).  Prints one JSON line: paths/s of whole epochs
(wall clock of MyOptimizer.train, file loading excluded, the counting pass and per-epoch reshuffle included) next to the resident
bench figure's step time.  usage: gpu_train_epoch.py [pairs_per_batch] [epochs] [minibatches_per_file]"""
import io, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kprn_amd import _ffi, formats, synth
from kprn_amd.batcher import BatcherFileList
from kprn_amd.optimizer import MyOptimizer

minibatch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 12
Ve = 2851220
d = tempfile.mkdtemp(prefix="kprn_epoch_")
names, total_paths = [], 0
for i, P in enumerate([1, 2, 3, 4, 5, 8]):
    pairs = NB * minibatch                     # NB minibatches per bucket file -> 6 NB steps per epoch
    idx, labels = synth.make_paths(pairs, P, 6, Ve=Ve, seed=100 + i)
    nm = "train_%d.npz" % P
    formats.save_path_file(os.path.join(d, nm), labels, idx, 1)
    names.append(nm)
    total_paths += pairs * P
open(os.path.join(d, "train.list"), "w").write("\n".join(names) + "\n")
eng = _ffi.Engine(6, Ve, 9, 16, 32, 16, 64, 2, seed=1)
opt = _ffi.make_opt(method=1, lr=1e-3)
res = {}
for shuffle in (True, False):
    tb = BatcherFileList(d, minibatch, shuffle, 100, True, "train.list", seed=3)
    out = io.StringIO()
    mo = MyOptimizer(eng, {"numEpochs": epochs, "epochHooks": [], "minibatchsize": minibatch}, opt, out=out)
    t0 = time.time()
    hist = mo.train(tb)
    dt = time.time() - t0
    rates = [float(l.split("=")[1]) for l in out.getvalue().splitlines() if l.startswith("examples/sec")]
    per_epoch = [total_paths / (minibatch * 6 * NB / r) for r in rates]    # examples = pairs; 6 NB minibatches per epoch
    res["shuffled" if shuffle else "fixed_order"] = {"epochs": epochs, "wall_s": round(dt, 3), "paths_per_s_by_epoch": [round(x) for x in per_epoch],
                                                      "loss_by_epoch": [round(float(h), 5) for h in hist]}
print(json.dumps({"what": "MyOptimizer.train over BatcherFileList on .npz bucket files", "paths_per_epoch": total_paths, "steps_per_epoch": 6 * NB,
                  "pairs_per_minibatch": minibatch, "result": res}))
