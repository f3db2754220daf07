#!/usr/bin/env python3
"""Static check of the hand-placed (inline-asm) MFMAs: hipcc's hazard recogniser cannot see inside an asm statement, so
nothing inserts the wait states gfx950 needs between a VALU write of a VGPR and an MFMA that reads it as SrcA / SrcB / SrcC
(2 wait states), or between an MFMA result and a non-MFMA reader.  This script compiles the kernels to ISA and reports every
v_mfma whose source registers are written by a VALU instruction in the 2 issue slots in front of it (s_nop N counts N+1).

usage: scripts/check_mfma_hazards.py [file.hip ...]      (default: every fused kernel file)
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["lstm_fused_fwd.hip", "lstm_fused_bwd.hip", "lstm_fused_fwd_mc.hip"]


def regs(tok):
    """'v[12:15]' -> ('v', {12..15}); 'v7' -> ('v', {7}); 'a[0:3]' -> ('a', {...}); else None"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None


def check(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include"),
                               "-S", "--cuda-device-only", "-o", out, path])
        text = open(out).read()
    bad = 0
    total = 0
    for km in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)s_endpgm", text, re.S):
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
    return bad


if __name__ == "__main__":
    files = sys.argv[1:] or [os.path.join(ROOT, "kprn_amd", "csrc", f) for f in DEFAULT]
    sys.exit(1 if sum(check(f) for f in files) else 0)
