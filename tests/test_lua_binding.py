"""not gpu: bindings/kprn.lua against include/kprn.h WITHOUT a LuaJIT (the image has none; the stub has never executed).

The LuaJIT FFI trusts its cdef: a prototype whose parameter TYPES or ORDER differ from the library's is a silent corruption at call time, not
an error.  So the cdef block is parsed here as C and compared with the header prototype by prototype -- return type, parameter count and every
parameter's type (names dropped, qualifiers kept) -- and the two structs field by field, in order.  Every C.kprn_* the Lua code calls must be
declared in the cdef, and every declared symbol must be exported by the built library (tests/test_abi.py checks the header's list the same way).
Reference call sites the stub stands for: model/optimizer/MyOptimizer.lua:177-221, eval/test_from_checkpoint.lua:68-118."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPES = r"(?:const\s+)?(?:unsigned\s+)?(?:kprn_\w+|int32_t|int64_t|uint64_t|size_t|float|double|char|void|int)(?:\s+const)?(?:\s*\*+)?"


def _strip_comments(txt):
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    return re.sub(r"//[^\n]*", " ", txt)


def _norm_type(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\s*\*", "*", t)
    return t


def _param_type(p):
    """'const kprn_batch* b' / 'float*' / 'int32_t class_id' -> the type alone"""
    p = re.sub(r"\s+", " ", p.strip())
    if p in ("void", ""):
        return None
    m = re.match(r"^(" + TYPES + r")\s*(\w+)?(\s*\[\s*\d*\s*\])?$", p)
    assert m, "unparsed parameter: %r" % p
    t = _norm_type(m.group(1))
    if m.group(3):
        t += "*"
    return t


def prototypes(txt):
    """{name: (return type, [parameter types])} of every kprn_* function declared in C text"""
    txt = _strip_comments(txt)
    out = {}
    for m in re.finditer(r"(?:^|[;}\n])\s*(" + TYPES + r")\s*(kprn_\w+)\s*\(([^()]*)\)\s*;", txt, re.S):
        params = [q for q in (_param_type(p) for p in m.group(3).split(",")) if q is not None]
        out[m.group(2)] = (_norm_type(m.group(1)), params)
    return out


def struct_fields(txt, name):
    """[(type, field)] of `typedef struct { ... } name;` (or `struct name { ... }`), in declaration order"""
    txt = _strip_comments(txt)
    m = re.search(r"typedef\s+struct\s*(?:\w+\s*)?\{([^{}]*)\}\s*" + name + r"\s*;", txt, re.S)
    assert m, name
    fields = []
    for decl in m.group(1).split(";"):
        decl = decl.strip()
        if not decl:
            continue
        tm = re.match(r"^(" + TYPES + r")\s*(.*)$", decl, re.S)
        assert tm, decl
        base = _norm_type(tm.group(1))
        for f in tm.group(2).split(","):
            f = f.strip()
            stars = f.count("*")
            fields.append((base + "*" * stars, f.replace("*", "").strip()))
    return fields


def _sources():
    hdr = open(os.path.join(ROOT, "include", "kprn.h")).read()
    lua = open(os.path.join(ROOT, "bindings", "kprn.lua")).read()
    cdef = lua[lua.index("ffi.cdef[[") + len("ffi.cdef[["):]
    cdef = cdef[:cdef.index("]]")]
    return hdr, lua, cdef


def test_every_cdef_prototype_has_the_headers_types():
    hdr, lua, cdef = _sources()
    want, got = prototypes(hdr), prototypes(cdef)
    assert len(got) >= 35 and len(want) >= len(got)
    for name, (ret, params) in sorted(got.items()):
        assert name in want, "declared in bindings/kprn.lua but not in include/kprn.h: " + name
        wret, wparams = want[name]
        assert ret == wret, (name, ret, wret)
        assert len(params) == len(wparams), (name, params, wparams)
        for i, (p, w) in enumerate(zip(params, wparams)):
            assert p == w, "%s: parameter %d is %r in the cdef and %r in the header" % (name, i + 1, p, w)


def test_cdef_structs_match_the_header_field_by_field():
    hdr, lua, cdef = _sources()
    for name in ("kprn_config", "kprn_opt"):
        assert struct_fields(cdef, name) == struct_fields(hdr, name), name


def test_every_call_the_stub_makes_is_declared_and_exported():
    import subprocess
    from kprn_amd import build as kbuild
    hdr, lua, cdef = _sources()
    code = "\n".join(l.split("--")[0] for l in lua[lua.index("]]", lua.index("ffi.cdef[[")):].splitlines())
    called = set(re.findall(r"\bC\.(kprn_\w+)", code))
    declared = set(prototypes(cdef))
    assert called and called <= declared, sorted(called - declared)
    syms = subprocess.check_output(["nm", "-D", kbuild.build()]).decode()
    missing = [s for s in sorted(declared) if " T %s" % s not in syms]
    assert not missing, missing


def test_lua_source_is_lexically_well_formed():
    """no interpreter here: at least the block structure must balance -- every function / if / for / while / do opens what an `end` closes, long
    brackets and parentheses pair up (comments and strings removed first)"""
    hdr, lua, cdef = _sources()
    src = re.sub(r"--\[\[.*?\]\]", " ", lua, flags=re.S)
    src = re.sub(r"\[\[.*?\]\]", '""', src, flags=re.S)
    src = "\n".join(l.split("--")[0] for l in src.splitlines())
    src = re.sub(r"'(?:\\.|[^'\\])*'|\"(?:\\.|[^\"\\])*\"", '""', src)
    for a, b in ("()", "{}", "[]"):
        assert src.count(a) == src.count(b), (a, src.count(a), src.count(b))
    toks = re.findall(r"\b(function|if|for|while|repeat|until|do|end|then|elseif)\b", src)
    depth = 0
    for i, t in enumerate(toks):
        if t in ("function", "if", "repeat"):
            depth += 1
        elif t == "do":
            # `for ... do` / `while ... do` open ONE block between them: count the block at its `do` only when it stands alone
            prev = [q for q in toks[:i] if q in ("for", "while", "do", "end")]
            if not prev or prev[-1] not in ("for", "while"):
                depth += 1
        elif t in ("for", "while"):
            depth += 1
        elif t in ("end", "until"):
            depth -= 1
            assert depth >= 0, "unbalanced end near token %d" % i
    assert depth == 0, depth
