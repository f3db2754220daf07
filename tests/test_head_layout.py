"""Index algebra of the bf16 pipeline's head launches (kprn_amd/csrc/lstm_bf16.hip k_head_fwd16, kernels_basic.hip k_head_bwd_w), replayed lane by
lane in numpy.

k_head_fwd16 computes nn.Linear(H, C) (OneModel.lua:275) on v_mfma_f32_16x16x32_bf16 with a PERMUTED k index: a lane reads 64 contiguous bytes of
its path's row per 64-k block, so MFMA (j, i) of lane (row, kg) contracts over k = 64 j + 16 kg + 8 i + q, q = 0..7, and the weight fragments are
laid out in LDS in that same order.  This model states the instruction's operand / result layouts (A: lane (m, kg) holds A[m][8 kg .. 8 kg + 7];
B: lane (n, kg) holds B[8 kg .. + 7][n]; D: lane (n, rg), register r holds D[4 rg + r][n]), fills the LDS image by the kernel's formula, walks the
kernel's blocks and compares with the plain product (float64: only the index algebra is under test; the GPU parity tests check the kernel itself).
k_head_bwd_w's row / column-group split and LDS reduction are replayed the same way."""
import numpy as np
import pytest


def mfma_16x16x32(a_frag, b_frag, acc):
    """a_frag, b_frag: [64 lanes][8]; acc: [64 lanes][4] -> acc + A B in the hardware layouts"""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for lane in range(64):
        m, kg = lane & 15, lane >> 4
        A[m, 8 * kg:8 * kg + 8] = a_frag[lane]
        B[8 * kg:8 * kg + 8, m] = b_frag[lane]
    Dm = A @ B
    out = acc.copy()
    for lane in range(64):
        n, rg = lane & 15, lane >> 4
        for r in range(4):
            out[lane, r] += Dm[4 * rg + r, n]
    return out


def head_fwd_model(hT, W, bias, grid):
    """the launch: `grid` workgroups of 4 waves; returns S [N][C]"""
    N, H = hT.shape
    C = W.shape[0]
    NT, NJ = (C + 15) // 16, H // 64
    E = NJ * 2 * NT * 64
    # the LDS image, filled by the kernel's formula (entry e <- 8 consecutive k of one class row; rows past C zero)
    wf = np.zeros((E, 8))
    for e in range(E):
        l, nt, ji = e & 63, (e >> 6) % NT, (e >> 6) // NT
        cls = nt * 16 + (l & 15)
        k0 = 64 * (ji >> 1) + 16 * (l >> 4) + 8 * (ji & 1)
        if cls < C:
            wf[e] = W[cls, k0:k0 + 8]
    S = np.full((N, C), np.nan)
    for wg in range(grid):
        for wave in range(4):
            blk = wg * 4 + wave
            while blk * 16 < N:
                acc = np.zeros((NT, 64, 4))
                for j in range(NJ):
                    a0 = np.zeros((64, 8))
                    a1 = np.zeros((64, 8))
                    for lane in range(64):
                        row = min(blk * 16 + (lane & 15), N - 1)
                        x = hT[row, 64 * j + 16 * (lane >> 4):64 * j + 16 * (lane >> 4) + 16]   # the lane's 64 contiguous bytes
                        a0[lane], a1[lane] = x[:8], x[8:]
                    for nt in range(NT):
                        w0 = wf[((j * 2) * NT + nt) * 64:((j * 2) * NT + nt) * 64 + 64]
                        w1 = wf[((j * 2 + 1) * NT + nt) * 64:((j * 2 + 1) * NT + nt) * 64 + 64]   # = w0[(NT + nt) * 64] of the kernel
                        acc[nt] = mfma_16x16x32(a0, w0, acc[nt])
                        acc[nt] = mfma_16x16x32(a1, w1, acc[nt])
                for nt in range(NT):
                    for lane in range(64):
                        col = nt * 16 + (lane & 15)
                        if col >= C:
                            continue
                        for r in range(4):
                            ro = blk * 16 + 4 * (lane >> 4) + r
                            if ro < N:
                                assert np.isnan(S[ro, col])      # every score is written exactly once
                                S[ro, col] = acc[nt, lane, r] + bias[col]
                blk += grid * 4
    return S


@pytest.mark.parametrize("N,H,C,grid", [(100, 128, 46, 1), (37, 384, 46, 2), (64, 192, 16, 3), (50, 256, 61, 1), (16, 128, 3, 4)])
def test_head_forward_fragment_order_and_k_permutation(N, H, C, grid):
    rng = np.random.default_rng(N + H + C)
    hT, W, b = rng.normal(size=(N, H)), rng.normal(size=(C, H)), rng.normal(size=C)
    S = head_fwd_model(hT, W, b, grid)
    np.testing.assert_allclose(S, hT @ W.T + b, rtol=1e-12, atol=1e-12)


def head_bwd_model(dS, hT, Wout, cid, rows_per_block, want_dH):
    """k_head_bwd_w: a workgroup takes rows_per_block rows, wave w rows r0 + w, r0 + w + 4, ..; lane l the columns l + 64 g; sums meet in LDS"""
    N, H = hT.shape
    NG = (H + 63) // 64
    gW = np.zeros_like(Wout)
    gb = np.zeros(Wout.shape[0])
    dH = np.full((N, H), np.nan) if want_dH else None
    for blk in range((N + rows_per_block - 1) // rows_per_block):
        r0, r1 = blk * rows_per_block, min(N, (blk + 1) * rows_per_block)
        red = np.zeros((4, NG * 64 + 1))
        for wave in range(4):
            acc = np.zeros((NG, 64))
            bsum = 0.0
            for n in range(r0 + wave, r1, 4):
                d = dS[n]
                bsum += d
                for g in range(NG):
                    for lane in range(64):
                        j = g * 64 + lane
                        jc = min(j, H - 1)                   # idle lanes re-read column H - 1 and drop the sum
                        acc[g, lane] += d * hT[n, jc]
                        if want_dH and j < H:
                            assert np.isnan(dH[n, j])
                            dH[n, j] = d * Wout[cid, j]
            red[wave, :NG * 64] = acc.reshape(-1)
            red[wave, NG * 64] = bsum
        for j in range(H):
            gW[cid, j] += red[:, j].sum()
        gb[cid] += red[:, NG * 64].sum()
    return gW, gb, dH


@pytest.mark.parametrize("N,H,rpb,want_dH", [(300, 250, 64, True), (513, 384, 256, False), (70, 64, 32, True), (129, 100, 16, False)])
def test_head_backward_row_and_column_split(N, H, rpb, want_dH):
    rng = np.random.default_rng(N + H)
    C, cid = 5, 3
    dS, hT, Wout = rng.normal(size=N), rng.normal(size=(N, H)), rng.normal(size=(C, H))
    gW, gb, dH = head_bwd_model(dS, hT, Wout, cid, rpb, want_dH)
    want = np.zeros_like(Wout)
    want[cid] = dS @ hT
    np.testing.assert_allclose(gW, want, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(gb[cid], dS.sum(), rtol=1e-12)
    assert gb[np.arange(C) != cid].max() == 0
    if want_dH:
        np.testing.assert_allclose(dH, np.outer(dS, Wout[cid]), rtol=1e-12)
