#!/usr/bin/env python3
"""End-to-end scoring through the reference-shaped entry point: path files on disk -> scoring.test_from_checkpoint -> test.res
(eval/test_from_checkpoint.lua: counter \\t %.5f \\t label per pair), C2 shapes.  One JSON line: pairs/s and paths/s of the whole call
(file reading + scoring + formatting + writing).  usage: gpu_score_files.py [pairs_per_file]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from kprn_amd import _ffi, formats, synth, scoring

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
Ve = 2851220
d = tempfile.mkdtemp(prefix="kprn_score_")
names, total_paths, total_pairs = [], 0, 0
for i, P in enumerate([1, 2, 3, 4, 5, 8]):
    idx, labels = synth.make_paths(pairs, P, 6, Ve=Ve, seed=200 + i)
    nm = "test_%d.npz" % P
    formats.save_path_file(os.path.join(d, nm), labels, idx, 1)
    names.append(nm)
    total_paths += pairs * P
    total_pairs += pairs
open(os.path.join(d, "test.list"), "w").write("\n".join(names) + "\n")
eng = _ffi.Engine(6, Ve, 9, 16, 32, 16, 64, 2, seed=1)
out = os.path.join(d, "test.res")
from kprn_amd.batcher import BatcherFileList
res, load_s, score_s = [], [], []
for rep in range(3):
    t0 = time.time()
    n = scoring.test_from_checkpoint(eng, d, "test.list", out)
    dt = time.time() - t0
    assert n == total_pairs
    res.append(dt)
    # the same call in its two halves: reading the files (np.load + id checks), then score + format + write
    t0 = time.time()
    fl = BatcherFileList(d, 512, False, 1000, True, "test.list")
    t1 = time.time()
    with open(out + ".2", "wb") as f:
        assert scoring.write_scores(eng, fl, f, 1) == total_pairs
    t2 = time.time()
    load_s.append(t1 - t0); score_s.append(t2 - t1)
assert open(out + ".2").read() == open(out).read()
lines = open(out).read().splitlines()
assert len(lines) == total_pairs and lines[-1].split("\t")[0] == str(total_pairs - 1)
print(json.dumps({"what": "scoring.test_from_checkpoint over 6 bucket files (P = 1,2,3,4,5,8; T = 6), file load + score + format + write",
                  "pairs": total_pairs, "paths": total_paths, "wall_s": [round(x, 3) for x in res],
                  "pairs_per_s": round(total_pairs / min(res)), "paths_per_s": round(total_paths / min(res)),
                  "file_load_s": round(min(load_s), 3), "score_format_write_s": round(min(score_s), 3),
                  "paths_per_s_after_load": round(total_paths / min(score_s))}))
