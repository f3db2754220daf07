"""CPU: static checks of the hand-scheduled kernels' ISA (hipcc -S needs no GPU).

The fused kernels issue their MFMAs as inline asm, which hipcc's hazard recogniser cannot see into; round 1 shipped a fix for wrong
scores caused by exactly that (a VALU read scheduled above an operand-less drain).  scripts/check_mfma_hazards.py looks at the
instructions the compiler finally emitted: (a) VALU write -> MFMA operand, (b) MFMA result -> non-MFMA reader.  Register spills of
the persistent kernels are pinned here too: none in the forward / top-layer kernels, and in the bottom-layer backward only the
per-tile address registers parked in the prologue -- nothing is spilled inside the step loop."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import check_mfma_hazards as chk  # noqa: E402

CSRC = os.path.join(ROOT, "kprn_amd", "csrc")


def _kernel(body):
    return "\n_Zfake:\n" + body + "\n\ts_endpgm\n"


def test_checker_sees_a_valu_result_fed_to_an_mfma_too_early():
    bad = _kernel("\tv_add_f32_e32 v5, v1, v2\n\tv_mfma_f32_16x16x4_f32 v[8:11], v5, a4, v[8:11]")
    ok = _kernel("\tv_add_f32_e32 v5, v1, v2\n\ts_nop 1\n\tv_mfma_f32_16x16x4_f32 v[8:11], v5, a4, v[8:11]")
    assert chk.check_text(bad, "fake") == 1
    assert chk.check_text(ok, "fake") == 0


def test_checker_sees_an_mfma_result_read_before_it_has_landed():
    """the round-1 bug class: the drain (s_nop) is BELOW the read"""
    hoisted = _kernel("\tv_mfma_f32_16x16x4_f32 v[8:11], v5, a4, v[8:11]\n\tv_mul_f32_e32 v20, v8, v8\n\ts_nop 15\n\ts_nop 15")
    drained = _kernel("\tv_mfma_f32_16x16x4_f32 v[8:11], v5, a4, v[8:11]\n\ts_nop 15\n\ts_nop 15\n\tv_mul_f32_e32 v20, v8, v8")
    stored = _kernel("\tv_mfma_f32_16x16x4_f32 a[0:3], v5, v6, a[0:3]\n\ts_nop 3\n\tglobal_store_dwordx4 v[2:3], a[0:3], off")
    behind_two_mfmas = _kernel("\tv_mfma_f32_16x16x4_f32 v[8:11], v5, a4, v[8:11]\n\tv_mfma_f32_16x16x4_f32 v[12:15], v5, a5, v[12:15]\n"
                               "\tv_mfma_f32_16x16x4_f32 v[16:19], v5, a6, v[16:19]\n\tv_mul_f32_e32 v20, v8, v8")
    assert chk.check_results(hoisted, "fake") == 1
    assert chk.check_results(drained, "fake") == 0
    assert chk.check_results(stored, "fake") == 1
    assert chk.check_results(behind_two_mfmas, "fake") == 0
    # loop back-edge: the read sits at the top of the body, the MFMA at its bottom
    loop = _kernel("\tv_mul_f32_e32 v20, v8, v8\n\tv_mfma_f32_16x16x4_f32 v[8:11], v5, a4, v[8:11]")
    assert chk.check_results(loop, "fake") == 1


@pytest.mark.parametrize("fname", chk.DEFAULT)
def test_fused_kernels_have_no_unprotected_mfma_hazards(fname):
    assert chk.check(os.path.join(CSRC, fname)) == 0


def test_register_spills_of_the_persistent_kernels():
    res = {}
    for f in ("lstm_fused_fwd.hip", "lstm_fused_bwd.hip", "lstm_fused_fwd_mc.hip"):
        res.update(chk.kernel_resources(chk.compile_isa(os.path.join(CSRC, f))))
    fused = {k: v for k, v in res.items() if re.search(r"k_lstm_(fwd|bwd|fwd_mc)I", k)}
    assert len(fused) >= 12
    for k, v in fused.items():
        assert v["vgpr_count"] <= 512
        bottom_bwd = "k_lstm_bwdILb1E" in k
        if "k_lstm_bwdILb0ELb0E" in k:
            assert v["vgpr_spill_count"] <= 16, (k, v)   # middle-layer backward (no launch site: the fused path stops at two layers): 11, all at tile starts
        elif not bottom_bwd:
            assert v["vgpr_spill_count"] <= 8, (k, v)    # forward, matrix-core forward: none; top backward: 6 (tile starts; round 6: the next tile's first step requested a tile ahead)
        elif "ILb1ELb0E" in k:
            assert v["vgpr_spill_count"] <= 64, (k, v)   # bottom-layer backward of two layers (gather + three output layouts): 49 with the hand-over (round 5: 19)
        else:
            assert v["vgpr_spill_count"] <= 80, (k, v)   # one layer, bottom and top at once: 71 (59)
    # ... and what is spilled stays out of the step bodies: the blocks that hold a step's MFMAs (the recurrent step of the backward: 416 + 3 x 32 + 512, its
    # last step 2 x 256; the forward's slots: 1024 / 8 x 68) write nothing to scratch and reload at most one register (the early / late hand-over pieces are
    # further instantiations of the same bodies: every copy is checked)
    # (round 6: k_lstm_fwd_dual holds BOTH forward bodies as two branches -- each must stay as clean as the kernel of its own)
    for f, pat, need in (("lstm_fused_bwd.hip", r"k_lstm_bwdIL", 2), ("lstm_fused_fwd.hip", r"k_lstm_fwdIL", 2), ("lstm_fused_fwd.hip", r"k_lstm_fwd_dualIL", 1)):
        text = chk.compile_isa(os.path.join(CSRC, f))
        seen = 0
        for km in re.finditer(r"\n(_ZN5fused\d+" + pat + r"\w+):[^\n]*\n(.*?)\n\.Lfunc_end", text, re.S):
            if "Li4EEEv" not in km.group(1):
                continue   # (the 16-row small-batch instantiations have registers to spare)
            blocks = re.split(r"\n\.LBB\d+_\d+:", km.group(2))
            hot = [b for b in blocks if len(re.findall(r"\bv_mfma", b)) >= 60]
            assert len(hot) >= 4, (km.group(1), len(hot))
            for b in hot:
                assert not re.findall(r"scratch_store", b), km.group(1)
                assert len(re.findall(r"scratch_load", b)) <= (4 if "bwdILb1E" in km.group(1) else 1), km.group(1)
            seen += 1
        assert seen >= need, (f, pat)


def test_persistent_bptt_kernel_keeps_its_state_in_registers():
    """lstm_bf16_bwd_persist.hip: the accumulators of dh_{t-1} and dc (2 x 96 registers), the weight ring and two save sets live in the 512-entry file of one
    wave per SIMD; the first builds spilled 309 / 146 registers into the step loop (dh in registers too; the dA^T row offsets precomputed per chunk).  The product
    instantiations must not spill at all, and their prefetch ring must survive hipcc's scheduling: counted vmcnt waits in front of the MFMAs, not vmcnt(0)."""
    text = chk.compile_isa(os.path.join(CSRC, "lstm_bf16_bwd_persist.hip"))
    res = {k: v for k, v in chk.kernel_resources(text).items() if "k_lstm16_bwd_persist" in k}
    assert len(res) >= 4
    for k, v in res.items():
        if "ILi2E" in k and k.endswith("Li0EEEvNS0_5BArgsE"):      # 64-row tiles, one workgroup per CU, no dx_e tile
            assert v["vgpr_spill_count"] == 0 and v["vgpr_count"] <= 512, (k, v)
        elif "ILi2ELi2ELi8E" in k:   # round 5, the product default: + the entity slice of dx as a fourth result tile per wave (32 more accumulators), ring of 8
            assert v["vgpr_spill_count"] <= 16 and v["vgpr_count"] <= 512, (k, v)
        elif "ILi2ELi2ELi16E" in k:  # ... ring of 16: opt-in, measured slower (reloads inside the step loop)
            assert v["vgpr_spill_count"] <= 96 and v["vgpr_count"] <= 512, (k, v)
        else:                 # 32-row tiles, two workgroups per CU (opt-in, measured slower): 256 registers per wave
            assert v["vgpr_spill_count"] <= 40 and v["vgpr_count"] <= 256, (k, v)
    for sym, n_mfma in (("ILi2ELi2ELi0E", 12 * 8 * 3 * 2), ("ILi2ELi2ELi8E", 12 * 8 * 4 * 2)):   # chunks x k-steps x result tiles x path tiles: ONE copy of the step body
        km = re.search(r"\n(_ZN5bf16p2pb20k_lstm16_bwd_persist" + sym + r"\w+):[^\n]*\n(.*?)s_endpgm", text, re.S)
        ins = [l.split(";")[0].strip() for l in km.group(2).split("\n")]
        ins = [l for l in ins if l and not l.startswith(".") and not l.endswith(":")]
        mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
        assert len(mf) == n_mfma, (sym, len(mf))
        waits = [l for l in ins[mf[0]:mf[-1]] if l.startswith("s_waitcnt") and "vmcnt" in l]
        drained = [l for l in waits if "vmcnt(0)" in l]
        assert len(drained) <= 4, (sym, drained[:5])   # (the first builds: a vmcnt(0) in front of most of the MFMAs)
        assert not [l for l in ins[mf[0]:mf[-1]] if l.startswith("scratch_")], sym   # whatever is spilled stays outside the step loop


def test_wide_fp32_layer_kernels_keep_state_in_registers_and_their_weight_ring_counted():
    """layer_f32_persist.hip (round 5): one persistent launch per wide fp32 recurrent layer.  The cell state / dh, dc accumulators live in registers (one wave per
    SIMD, 512 entries) and the weights arrive through an LDS-DMA ring with COUNTED vmcnt waits.  Pinned here, per instantiation <CELL, NCH, flag>: nothing is spilled
    (the H = 256 BPTT, NCH = 4, parks a few dozen registers: bounded), no scratch traffic between the MFMAs, the ring is still LDS-DMA, and most of its waits are
    counted ones -- a build in which hipcc sinks the waits to vmcnt(0) (the first builds did) would pass every parity test and lose the overlap.
    (These kernels leave early for workgroups without a tile, so their bodies end at .Lfunc_end, not at the first s_endpgm.)
    Round 6, CELL 2 (nn.GRU): the forward instantiations spill nothing; the BPTT launch parks 35 (122 with a layer above) registers, all of it in the cell's three parts --
    a schedule that kept save planes alive across the step loop spilled ~260 and reloaded them between the MFMAs.  Their forward's waits between the
    first and the last MFMA of a step include hipcc's own for the cell's bias quads and the x rows (most of them vmcnt(0)): only the counted ring waits are pinned there."""
    text = chk.compile_isa(os.path.join(CSRC, "layer_f32_persist.hip"))
    res = {k: v for k, v in chk.kernel_resources(text).items() if "lp32" in k and ("k_layer" in k or "k_bptt" in k)}
    assert len(res) == 30, sorted(res)
    seen = 0
    for km in re.finditer(r"\n(_ZN4lp32\d+k_(layer|bptt)\w+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", text, re.S):
        name, kind = km.group(1), km.group(2)
        nch4 = "ELi4EL" in name
        gru = "ILi2E" in name
        v = res[name]
        assert v["vgpr_count"] <= 512, (name, v)
        if kind == "bptt" and gru:
            assert v["vgpr_spill_count"] <= (136 if name.endswith("Lb1EEEvNS_6BPArgsE") else 44), (name, v)     # 122 / 35 (with the bias sums formed in the launch; 80 / 13 without)
        elif kind == "bptt" and nch4:
            assert v["vgpr_spill_count"] <= 64, (name, v)     # 27 / 51
        else:   # (the H = 256 FastLSTM scoring forward sits at 511 registers: two values parked in accumulator registers, no scratch memory)
            assert v["vgpr_spill_count"] == 0 or (nch4 and v["private_segment_fixed_size"] == 0 and v["vgpr_spill_count"] <= 4), (name, v)
        ins = [l.split(";")[0].strip() for l in km.group(3).split("\n")]
        ins = [l for l in ins if l and not l.startswith(".") and not l.endswith(":")]
        mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
        assert len(mf) >= (96 if gru else 128), (name, len(mf))
        inside = ins[mf[0]:mf[-1]]
        scratch = [l for l in inside if l.startswith("scratch_")]
        if kind == "bptt" and gru:   # (42 / 3: between a step's products, in the cell's parts)
            assert len(scratch) <= (96 if name.endswith("Lb1EEEvNS_6BPArgsE") else 24), (name, len(scratch))
        else:
            assert len(scratch) <= (16 if (kind == "bptt" and nch4) else 0), (name, len(scratch))
        assert [l for l in inside if l.startswith("global_load_lds")], name       # the weight ring
        waits = [l for l in inside if l.startswith("s_waitcnt") and "vmcnt" in l]
        drained = [l for l in waits if "vmcnt(0)" in l]
        if gru:
            counted = [l for l in waits if "vmcnt(0)" not in l]
            assert len(counted) >= 8, (name, len(counted), len(waits))      # the ring's own waits (vmcnt(8) / (4), with the save requests' allowance in the BPTT)
        else:
            assert waits and len(drained) <= (0.45 if kind == "layer" else 0.32) * len(waits), (name, len(drained), len(waits))
        seen += 1
    assert seen == 30


def test_operand_loads_are_not_a_chain_of_round_trips():
    """The conditional-load rule: in the kernels it was found in, a load is no longer followed by its own vmcnt(0).
    (The last load of a batch always is: the bars are counts, not zero.)"""
    isa = chk.compile_isa(os.path.join(CSRC, "gemm_f32.hip"))
    res = chk.serialized_loads(isa, r"gemm_kernel")
    assert len(res) == 16
    for k, (loads, serial) in res.items():
        # prologue + main loop: 16 operand loads each, in flight together; the accumulate epilogue's old values: 8 / 16 per row block, together
        assert loads >= 50 and serial <= 4, (k, loads, serial)
    isa = chk.compile_isa(os.path.join(CSRC, "batch_index.hip"))
    res = chk.serialized_loads(isa, r"k_entity_grad")
    assert len(res) == 3
    for k, (loads, serial) in res.items():
        assert loads >= 64 and serial <= 8, (k, loads, serial)   # 2 x 32 gathers in flight; the index reads and the passenger jobs' tails remain
    isa = chk.compile_isa(os.path.join(CSRC, "kernels_basic.hip"))
    res = chk.serialized_loads(isa, r"k_table_grad_mfma")
    assert res
    for k, (loads, serial) in res.items():
        assert serial <= 4, (k, loads, serial)
