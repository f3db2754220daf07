#!/usr/bin/env python3
"""Generates tests/golden/pathformat/: outputs of the REFERENCE's own movie_data_format.py (run in THIS container from
/root/reference, under Python 3 with the two Python-2-isms patched in memory: xrange -> range, `path_len / 2` ->
`path_len // 2`) on slices of its shipped sample inputs.  The reference source is read at generation time and never
copied into the repo; only its inputs (slices of the shipped samples), a synthesised covering vocabulary and its
OUTPUTS are committed.  tests/test_pathformat.py compares kprn_amd/pathformat.py with these byte for byte.

  python tests/golden/make_pathformat_golden.py            # needs /root/reference
"""
import io
import os
import shutil
import sys
import tempfile
from contextlib import redirect_stdout

REF = "/root/reference/release/songPathRnn/data"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "pathformat")
N_LINES = {"positive": 250, "negative": 250, "test": 300}
CASES = {  # name -> flags of movie_data_format.sh:2 and variations
    "m6_t1": dict(o="0", g="0", m="6", t="1"),     # the shipped invocation
    "m4_t1": dict(o="0", g="0", m="4", t="1"),     # shorter cap: length-6 paths dropped, some pairs missed
    "m6_t2": dict(o="0", g="0", m="6", t="2"),     # two type slots (pad slot + type)
    "m6_g1": dict(o="0", g="1", m="6", t="1"),     # relations only, extracted from entity paths
}


def build_inputs(work):
    os.makedirs(os.path.join(work, "input"))
    os.makedirs(os.path.join(work, "vocab"))
    ents = set()
    for kind, n in N_LINES.items():
        src = os.path.join(REF, "input", f"{kind}_matrix_sample.tsv.translated")
        with open(src) as f, open(os.path.join(work, "input", f"{kind}_matrix.tsv.translated"), "w") as w:
            for i, line in enumerate(f):
                if i >= n:
                    break
                w.write(line)
                parts = line.split("\t")
                ents.update([parts[0].strip(), parts[1].strip()])
                for p in parts[2].strip().split("###"):
                    ents.update(p.split("-")[1::2])
    for name in ("entity_type_id.txt", "all_relation_id.txt", "domain-label"):
        shutil.copy(os.path.join(REF, "vocab", name), os.path.join(work, "vocab", name))
    # covering vocabulary: every 7th entity is left out (exercises #UNK_ENTITY), every 5th has no type (#UNK_ENTITY_TYPE);
    # ids follow the KKBox convention (#UNK_ENTITY, #PAD_TOKEN last; format_entity_pair.py:13)
    ents = sorted(ents)
    kind_of = {"u": "user", "s": "song", "p": "person", "t": "type"}
    with open(os.path.join(work, "vocab", "all_entity_id.txt"), "w") as w, open(os.path.join(work, "vocab", "entity_to_type.txt"), "w") as tw:
        k = 0
        for i, e in enumerate(ents):
            if i % 7 == 3:
                continue
            w.write(f"{e}\t{k}\n")
            k += 1
            if i % 5 != 1:
                tw.write(f"{e}\t{kind_of.get(e[0], 'thing')}\n")
        w.write(f"#UNK_ENTITY\t{k}\n#PAD_TOKEN\t{k + 1}\n")


def run_reference(work, flags, out_name):
    src = open(os.path.join(REF, "movie_data_format.py")).read()
    assert src.count("xrange(") >= 3 and src.count("path_len / 2 + 2") == 2, "reference changed: re-check the patches"
    src = src.replace("xrange(", "range(").replace("path_len / 2 + 2", "path_len // 2 + 2")
    argv = ["movie_data_format.py", "-i", "input", "-d", out_name, "-o", flags["o"], "-g", flags["g"], "-e", "0", "-m", flags["m"], "-t", flags["t"]]
    cwd, old_argv = os.getcwd(), sys.argv
    os.chdir(work)
    sys.argv = argv
    buf = io.StringIO()
    try:
        with redirect_stdout(buf):
            exec(compile(src, "movie_data_format.py", "exec"), {"__name__": "__main__"})
    finally:
        os.chdir(cwd)
        sys.argv = old_argv
    return buf.getvalue()


def main():
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    work = tempfile.mkdtemp()
    build_inputs(work)
    shutil.copytree(os.path.join(work, "input"), os.path.join(OUT, "input"))
    shutil.copytree(os.path.join(work, "vocab"), os.path.join(OUT, "vocab"))
    for name, flags in CASES.items():
        log = run_reference(work, flags, "out_" + name)
        dst = os.path.join(OUT, "expected_" + name)
        shutil.copytree(os.path.join(work, "out_" + name), dst)
        with open(os.path.join(dst, "stdout_tail.txt"), "w") as w:
            w.write("\n".join(l for l in log.splitlines() if l.startswith(("Max length", "pad features", "Missed entity"))) + "\n")
        n = sum(len(files) for _, _, files in os.walk(dst))
        print(name, "->", n, "files")
    shutil.rmtree(work)


if __name__ == "__main__":
    main()
