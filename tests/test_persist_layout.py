"""Index algebra of the persistent bf16 layer kernel (kprn_amd/csrc/lstm_bf16_persist.hip), replayed lane by lane in numpy.

The kernel's correctness rests on five layout formulas that no CPU build can execute: the packed weight fragments (k_pack_w), the packed
bias image, the fragment-major operand tiles in LDS, the lane -> (gate, hidden unit, path) map of the transposed product's accumulators,
and the position the cell writes h_t to so that the next step's linear copy is again a B fragment.  This model states
v_mfma_f32_32x32x16_bf16's operand / result layouts (A: lane (m, kg) holds A[m][8 kg .. 8 kg + 7]; B: lane (n, kg) holds
B[8 kg .. + 7][n]; D: lane (n, half), register r holds D[(r & 3) + 8 (r >> 2) + 4 half][n]) and pushes a small FastLSTM through the
same formulas; the result must equal the plain recurrence (float64: only the index algebra is under test).  The GPU parity tests
check the kernel itself against the oracle."""
import numpy as np


def pack_w(Wi, Wo, bi, D, H):
    KS = (D + H) // 16
    NCH = H // 32
    Wp = np.zeros((NCH, 4, KS, 64, 8))
    for c in range(NCH):
        for w in range(4):
            for s in range(KS):
                for lane in range(64):
                    m, kg = lane & 31, lane >> 5
                    row = (m >> 3) * H + 32 * c + 8 * w + (m & 7)
                    for j in range(8):
                        k = 16 * s + 8 * kg + j
                        Wp[c, w, s, lane, j] = Wi[row, k] if k < D else Wo[row, k - D]
    Bp = np.zeros((NCH, 4, 2, 16))
    for c in range(NCH):
        for w in range(4):
            for half in range(2):
                for r in range(16):
                    Bp[c, w, half, r] = bi[(r >> 2) * H + 32 * c + 8 * w + 4 * half + (r & 3)]
    return Wp, Bp


def mfma_32x32x16(a_frag, b_frag, acc):
    """a_frag, b_frag: [64 lanes][8]; acc: [64 lanes][16] -> acc + A B in the hardware layouts"""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for lane in range(64):
        i, kg = lane & 31, lane >> 5
        A[i, 8 * kg:8 * kg + 8] = a_frag[lane]
        B[8 * kg:8 * kg + 8, i] = b_frag[lane]
    Dm = A @ B
    out = acc.copy()
    for lane in range(64):
        n, half = lane & 31, lane >> 5
        for r in range(16):
            out[lane, r] += Dm[(r & 3) + 8 * (r >> 2) + 4 * half, n]
    return out


def sigm(x):
    return 1.0 / (1.0 + np.exp(-x))


def run_kernel_model(x, Wi, Wo, bi, D, H, NPT):
    """x: [T][rows][D] step inputs of one work tile (rows = 32 NPT); returns h_T [rows][H], c_T [rows][H]"""
    T = x.shape[0]
    KX, KH = D // 16, H // 16
    NCH = H // 32
    Wp, Bp = pack_w(Wi, Wo, bi, D, H)
    hscr = np.zeros((2, NPT * KH * 64 * 8))   # fragment-order slabs
    cscr = np.zeros((NCH, NPT, 4, 64, 4))
    hT = np.zeros((32 * NPT, H))
    cT = np.zeros((32 * NPT, H))
    HB = None
    for t in range(T):
        # the gather: LDS piece (pt, s), lane (n, kg) <- x[32 pt + n][16 s + 8 kg ..]
        XB = np.zeros((NPT, KX, 64, 8))
        for pt in range(NPT):
            for s in range(KX):
                for lane in range(64):
                    n, kg = lane & 31, lane >> 5
                    XB[pt, s, lane] = x[t, 32 * pt + n, 16 * s + 8 * kg:16 * s + 8 * kg + 8]
        if t > 0:   # fetch_h: a linear copy of the slab
            HB = hscr[(t - 1) & 1].reshape(NPT, KH, 64, 8).copy()
        for c in range(NCH):
            for w in range(4):
                acc = np.zeros((NPT, 64, 16))
                for pt in range(NPT):
                    for lane in range(64):
                        acc[pt, lane] = Bp[c, w, lane >> 5]
                for s in range(KX):
                    for pt in range(NPT):
                        acc[pt] = mfma_32x32x16(Wp[c, w, s], XB[pt, s], acc[pt])
                if t > 0:
                    for s in range(KH):
                        for pt in range(NPT):
                            acc[pt] = mfma_32x32x16(Wp[c, w, KX + s], HB[pt, s], acc[pt])
                # the cell, per lane
                for pt in range(NPT):
                    for lane in range(64):
                        ln, half = lane & 31, lane >> 5
                        u0 = 32 * c + 8 * w + 4 * half
                        cp = cscr[c, pt, w, lane] if t > 0 else np.zeros(4)
                        pre = acc[pt, lane]
                        ig, gg, fg, og = sigm(pre[0:4]), np.tanh(pre[4:8]), sigm(pre[8:12]), sigm(pre[12:16])
                        cc = fg * cp + ig * gg
                        hh = og * np.tanh(cc)
                        cscr[c, pt, w, lane] = cc
                        sh, kg = 2 * c + (w >> 1), w & 1
                        pos = ((pt * KH + sh) * 64 + kg * 32 + ln) * 8 + 4 * half
                        hscr[t & 1][pos:pos + 4] = hh
                        if t == T - 1:
                            hT[32 * pt + ln, u0:u0 + 4] = hh
                            cT[32 * pt + ln, u0:u0 + 4] = cc
    return hT, cT


def reference(x, Wi, Wo, bi, H):
    T, R, _ = x.shape
    h = np.zeros((R, H))
    c = np.zeros((R, H))
    for t in range(T):
        a = x[t] @ Wi.T + bi + h @ Wo.T
        i, g, f, o = sigm(a[:, :H]), np.tanh(a[:, H:2 * H]), sigm(a[:, 2 * H:3 * H]), sigm(a[:, 3 * H:])   # FastLSTM chunk order [A2]
        c = f * c + i * g
        h = o * np.tanh(c)
    return h, c


def test_persistent_kernel_index_algebra():
    rng = np.random.default_rng(5)
    D, H, T, NPT = 32, 64, 3, 2   # KX = 2, KH = 4, two chunks: every formula is exercised, D != H on purpose
    Wi = rng.normal(size=(4 * H, D)) * 0.3
    Wo = rng.normal(size=(4 * H, H)) * 0.3
    bi = rng.normal(size=4 * H) * 0.3
    x = rng.normal(size=(T, 32 * NPT, D))
    hT, cT = run_kernel_model(x, Wi, Wo, bi, D, H, NPT)
    hr, cr = reference(x, Wi, Wo, bi, H)
    assert np.max(np.abs(hT - hr)) < 1e-12
    assert np.max(np.abs(cT - cr)) < 1e-12
