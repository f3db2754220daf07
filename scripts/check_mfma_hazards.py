#!/usr/bin/env python3
"""Static check of the hand-placed (inline-asm) MFMAs: hipcc's hazard recogniser cannot see inside an asm statement, so
nothing inserts the wait states gfx950 needs between a VALU write of a VGPR and an MFMA that reads it as SrcA / SrcB / SrcC
(2 wait states), or between an MFMA result and a non-MFMA reader.  This script compiles the kernels to ISA and reports
  (a) every v_mfma whose source registers are written by a VALU instruction in the 2 issue slots in front of it (s_nop N counts N+1);
  (b) every non-MFMA instruction (VALU, v_accvgpr_read, LDS / global / scratch store) that reads a register an MFMA wrote fewer
      than 10 issue cycles earlier (another MFMA in between counts 8 -- it occupies the pipe for at least that --, s_nop N counts
      N + 1, anything else 1; 10 is what hipcc's own hazard recogniser leaves behind its builtin f32 16x16x4 MFMAs).  This is the class that produced
      wrong scores in round 1 (a VALU read hoisted above an operand-less drain); the drains are now followed by pins, and this
      check looks at what the compiler finally emitted.  Loop bodies are covered by scanning every kernel twice back to back.
tests/test_hazards.py runs both on the CPU (hipcc -S needs no GPU).

usage: scripts/check_mfma_hazards.py [file.hip ...]      (default: every fused kernel file)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["lstm_fused_fwd.hip", "lstm_fused_bwd.hip", "lstm_fused_fwd_mc.hip", "layer_f32_persist.hip"]   # (the last one: pinned sched_barrier groups around builtin MFMAs)


def regs(tok):
    """'v[12:15]' -> ('v', {12..15}); 'v7' -> ('v', {7}); 'a[0:3]' -> ('a', {...}); else None"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None


_ISA = {}


def compile_isa(path):
    """gfx950 ISA text of one .hip file (hipcc -S --cuda-device-only; cached per process)"""
    if path not in _ISA:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            # (with the flags the source asks for in its `// hipcc-flags:` line, as kprn_amd/build.py compiles it: the register allocation the tests pin
            #  is the shipped one)
            extra = []
            with open(path) as f:
                for line in f.readlines()[:5]:
                    if line.startswith("// hipcc-flags:"):
                        extra = [w for w in line[len("// hipcc-flags:"):].split() if w.startswith("-")]
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include")] + extra +
                                  ["-S", "--cuda-device-only", "-o", out, path])
            _ISA[path] = open(out).read()
    return _ISA[path]


def kernel_resources(text):
    """{kernel symbol: {vgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size}} from the .amdhsa metadata"""
    out = {}
    for m in re.finditer(r"\.name:\s+(\S+)\s*\n(.*?)(?=\n\s+- \.|\namdhsa\.target|\Z)", text, re.S):
        blk = m.group(0)
        rec = {}
        for key in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
            mm = re.search(r"\.%s:\s+(\d+)" % key, blk)
            if mm:
                rec[key] = int(mm.group(1))
        if "vgpr_count" in rec:
            out[m.group(1)] = rec
    return out



def serialized_loads(text, kernel_regex):
    """{kernel symbol: (vector memory loads, loads that are waited for with vmcnt(0) before the next load is issued)} for the kernels whose symbol
    matches.  hipcc turns `in_range ? p[i] : 0` (or a select right behind an unconditional load) into a branch with the load AND its wait inside:
    a loop of such loads is a chain of dependent round trips.  A load counts as serialized when, walking forward over at most
    8 instructions, an `s_waitcnt vmcnt(0)` comes before any other load."""
    out = {}
    for km in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S):   # (to the function's end: a kernel with early returns has several s_endpgm)
        if not re.search(kernel_regex, km.group(1)):
            continue
        ins = [l.split(";")[0].strip() for l in km.group(2).split("\n")]
        ins = [l for l in ins if l and not l.startswith(".") and not l.endswith(":")]
        is_load = lambda l: re.match(r"(global_load|buffer_load)", l) and "lds" not in l
        loads = serial = 0
        for i, l in enumerate(ins):
            if not is_load(l):
                continue
            loads += 1
            for nxt in ins[i + 1:i + 9]:
                if is_load(nxt):
                    break
                if nxt.startswith("s_waitcnt") and "vmcnt(0)" in nxt:
                    serial += 1
                    break
        out[km.group(1)] = (loads, serial)
    return out

def check(path):
    text = compile_isa(path)
    return check_text(text, os.path.basename(path))


def check_text(text, fname):
    path = fname
    bad = 0
    total = 0
    for km in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)(?:\n\.Lfunc_end|s_endpgm(?![\s\S]*?\n\.Lfunc_end))", text, re.S):
        name, body = km.group(1), km.group(2)
        ins = []
        for l in body.split("\n"):
            l = l.split(";")[0].strip()
            if l and not l.startswith(".") and not l.endswith(":"):
                ins.append(l)
        for i, l in enumerate(ins):
            if not l.startswith("v_mfma"):
                continue
            total += 1
            ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
            srcs = [regs(o) for o in ops[1:4]]
            need = 2
            jx = i - 1
            while jx >= 0 and need > 0:
                p = ins[jx]
                if p.startswith("s_nop"):
                    need -= int(p.split()[1]) + 1
                    jx -= 1
                    continue
                if p.startswith("v_") and not p.startswith("v_mfma") and not p.startswith("v_cmp"):
                    dst = regs(p.split(None, 1)[1].split(",")[0].strip())
                    if dst and dst[0] == "v":
                        for sreg in srcs:
                            if sreg and sreg[0] == "v" and (sreg[1] & dst[1]):
                                bad += 1
                                print(f"{os.path.basename(path)} {name[:48]}: '{p}' feeds '{l}' {i - jx} slot(s) later")
                                break
                need -= 1
                jx -= 1
    print(f"{os.path.basename(path)}: {total} MFMAs checked, {bad} unprotected VALU -> MFMA operand hazards")
    bad_b = check_results(text, os.path.basename(path))
    return bad + bad_b


STORE_PREFIXES = ("global_store", "ds_write", "ds_add", "scratch_store", "buffer_store", "flat_store", "global_atomic", "flat_atomic", "ds_bpermute", "ds_permute")
NEED = 10   # what hipcc itself leaves between its own (builtin) f32 16x16x4 MFMAs and the first read of their result


def check_results(text, fname):
    """(b): MFMA result -> non-MFMA reader closer than NEED issue cycles"""
    bad = 0
    checked = 0
    for km in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)(?:\n\.Lfunc_end|s_endpgm(?![\s\S]*?\n\.Lfunc_end))", text, re.S):
        name, body = km.group(1), km.group(2)
        ins = []
        for l in body.split("\n"):
            l = l.split(";")[0].strip()
            if l and not l.startswith(".") and not l.endswith(":"):
                ins.append(l)
        if not any(l.startswith("v_mfma") for l in ins):
            continue
        seq = ins + ins   # a second pass sees what the first iteration's tail does to the next iteration's head
        pending = {}      # (file, reg) -> cycles since the MFMA that wrote it issued
        seen = set()      # positions already reported (the second pass repeats the first pass's findings)
        for pos_, l in enumerate(seq):
            parts = l.split(None, 1)
            op = parts[0]
            ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
            if op.startswith("v_mfma"):
                for k in list(pending):
                    pending[k] += 8
                    if pending[k] >= NEED:
                        del pending[k]
                d = regs(ops[0]) if ops else None
                if d:
                    for r in d[1]:
                        pending[(d[0], r)] = 0
                continue
            if op.startswith("s_nop"):
                step = int(ops[0]) + 1
            else:
                step = 1
                if pending and (op.startswith("v_") or op.startswith(STORE_PREFIXES)):
                    srcs = ops if op.startswith(STORE_PREFIXES) else ops[1:]
                    if op.startswith(("v_fmac", "v_mac", "v_pk_fmac", "v_dot2c")):
                        srcs = ops
                    for o in srcs:
                        rg = regs(o.split(" ")[0])
                        if not rg:
                            continue
                        hit = [r for r in rg[1] if (rg[0], r) in pending]
                        if hit:
                            checked += 1
                            if pos_ % len(ins) not in seen:
                                seen.add(pos_ % len(ins))
                                bad += 1
                                print(f"{fname} {name[:48]}: '{l}' reads {rg[0]}{hit[0]} {pending[(rg[0], hit[0])]} cycle(s) after the MFMA that writes it")
                            for r in hit:
                                pending.pop((rg[0], r), None)
                # a non-MFMA WRITE of the register ends the window too
                if ops:
                    d = regs(ops[0].split(" ")[0])
                    if d and not op.startswith(STORE_PREFIXES):
                        for r in d[1]:
                            pending.pop((d[0], r), None)
            for k in list(pending):
                pending[k] += step
                if pending[k] >= NEED:
                    del pending[k]
    print(f"{fname}: {bad} MFMA result -> non-MFMA reader hazards")
    return bad


if __name__ == "__main__":
    files = sys.argv[1:] or [os.path.join(ROOT, "kprn_amd", "csrc", f) for f in DEFAULT]
    sys.exit(1 if sum(check(f) for f in files) else 0)
