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


# ---- the persistent BPTT kernel (kprn_amd/csrc/lstm_bf16_bwd_persist.hip) ------------------------------------------------------------
# Formulas under test: k_pack_wb's fragment order, the forward-record addressing (4 c + w = forward chunk * 8 + forward wave), the
# B-fragment placement of the cell backward's [di dg] / [df do] pieces, the (tile j, register 4 q + r) <-> (chunk 4 j + q, unit r)
# composition of the result tiles, the dA^T gather from the LDS tile and the bias-gradient rows.

def pack_wb(Wo, H, MJ):
    NW, KSC, NCH = 4, 8, H // 32
    FR = NCH * KSC * MJ
    WpB = np.zeros((NW, FR, 64, 8))
    for w in range(NW):
        for f in range(FR):
            j, ks, c = f % MJ, (f // MJ) % KSC, f // (MJ * KSC)
            for lane in range(64):
                m, kg = lane & 31, lane >> 5
                um = 32 * (4 * j + (m >> 3)) + 8 * w + (m & 7)
                for e in range(8):
                    gate, uk = 2 * kg + (e >> 2), 32 * c + 8 * (ks >> 1) + 4 * (ks & 1) + (e & 3)
                    WpB[w, f, lane, e] = Wo[gate * H + uk, um]
    return WpB


def forward_records(i, g, f, o, c, H):
    """[T][N][H] planes -> the forward's fragment-order saves (Cell::store): rec = (((t NU + unit) NCHF + cf) 8 + wf) 64 + lane"""
    T, N, _ = i.shape
    NU, NCHF = N // 32, H // 64
    A0 = np.zeros((T * NU * NCHF * 8 * 64, 8)); A1 = np.zeros_like(A0); CF = np.zeros((A0.shape[0], 4))
    for t in range(T):
        for u in range(NU):
            for cf in range(NCHF):
                for wf in range(8):
                    for lane in range(64):
                        ln, half = lane & 31, lane >> 5
                        rec = (((t * NU + u) * NCHF + cf) * 8 + wf) * 64 + lane
                        n, u0 = 32 * u + ln, 64 * cf + 8 * wf + 4 * half
                        A0[rec, :4], A0[rec, 4:] = i[t, n, u0:u0 + 4], g[t, n, u0:u0 + 4]
                        A1[rec, :4], A1[rec, 4:] = f[t, n, u0:u0 + 4], o[t, n, u0:u0 + 4]
                        CF[rec] = c[t, n, u0:u0 + 4]
    return A0, A1, CF


def run_bwd_model(A0, A1, CF, dS, Wc, Wo, T, N, H, MJ):
    NW, NPT, KSC, NCH, NCHF = 4, 2, 8, H // 32, H // 64
    NU, UREC = N // 32, H // 4 * 32
    FR = NCH * KSC * MJ
    WpB = pack_wb(Wo, H, MJ)
    dA = np.zeros((T, N, 4 * H)); dAT = np.zeros((4 * H, T, N)); db = np.zeros(4 * H)
    for tile in range(N // 64):
        dh = np.zeros((NW, MJ, NPT, 64, 16)); dc = np.zeros_like(dh)
        for w in range(NW):
            for c in range(NCH):
                for pt in range(NPT):
                    for lane in range(64):
                        ln, half = lane & 31, lane >> 5
                        wq = Wc[32 * c + 8 * w + 4 * half:32 * c + 8 * w + 4 * half + 4]
                        dh[w, c >> 2, pt, lane, 4 * (c & 3):4 * (c & 3) + 4] = wq * dS[tile * 64 + 32 * pt + ln]
        for t in range(T - 1, -1, -1):
            acc = np.zeros((NW, MJ, NPT, 64, 16))
            for c in range(NCH):
                buf = np.zeros((NPT, KSC, 64, 8))
                j, q = c >> 2, c & 3
                for w in range(NW):
                    for pt in range(NPT):
                        for lane in range(64):
                            ln, half = lane & 31, lane >> 5
                            rec = (t * NU + tile * NPT + pt) * UREC + (4 * c + w) * 64 + lane
                            ig, gg, fg, og = A0[rec, :4], A0[rec, 4:], A1[rec, :4], A1[rec, 4:]
                            cc = CF[rec]
                            cp = CF[rec - NU * UREC] if t > 0 else np.zeros(4)
                            tc = np.tanh(cc)
                            d_h = dh[w, j, pt, lane, 4 * q:4 * q + 4]
                            dO = d_h * tc
                            d_c = dc[w, j, pt, lane, 4 * q:4 * q + 4] + d_h * og * (1 - tc * tc)
                            di, dg, df, do = d_c * gg * ig * (1 - ig), d_c * ig * (1 - gg * gg), d_c * cp * fg * (1 - fg), dO * og * (1 - og)
                            dc[w, j, pt, lane, 4 * q:4 * q + 4] = d_c * fg
                            buf[pt, 2 * w + half, ln] = np.concatenate([di, dg])
                            buf[pt, 2 * w + half, 32 + ln] = np.concatenate([df, do])
                            n, u0 = tile * 64 + 32 * pt + ln, 32 * c + 8 * w + 4 * half
                            for gt, v in enumerate((di, dg, df, do)):
                                dA[t, n, gt * H + u0:gt * H + u0 + 4] = v
                # write_T (lane pieces -> transposed tile rows gate 32 + 8 w + 4 half + r) and emit_T (thread (oct, kc0): rows kc0 + 32 i)
                tt = np.zeros((128, 64))
                for w in range(NW):
                    for pt in range(NPT):
                        for lane in range(64):
                            ln, half = lane & 31, lane >> 5
                            pcs = (buf[pt, 2 * w + half, ln], buf[pt, 2 * w + half, 32 + ln])
                            for g4 in range(4):
                                for r in range(4):
                                    tt[g4 * 32 + 8 * w + 4 * half + r, 32 * pt + ln] = pcs[g4 >> 1][4 * (g4 & 1) + r]
                for tid in range(256):
                    oct_, kc0 = tid & 7, tid >> 3
                    for i4 in range(4):
                        v = tt[32 * i4 + kc0, 8 * oct_:8 * oct_ + 8]
                        row = i4 * H + 32 * c + kc0
                        dAT[row, t, tile * 64 + 8 * oct_:tile * 64 + 8 * oct_ + 8] = v
                        db[row] += v.sum()
                if t > 0:
                    for w in range(NW):
                        for ks in range(KSC):
                            for jj in range(MJ):
                                fidx = (c * KSC + ks) * MJ + jj
                                for pt in range(NPT):
                                    acc[w, jj, pt] = mfma_32x32x16(WpB[w, fidx], buf[pt, ks], acc[w, jj, pt])
            dh = acc
    return dA, dAT, db


def reference_bwd(i, g, f, o, c, dS, Wc, Wo, H):
    T, N, _ = i.shape
    dh = np.outer(dS, Wc); dc = np.zeros((N, H))
    dA = np.zeros((T, N, 4 * H))
    for t in range(T - 1, -1, -1):
        tc = np.tanh(c[t]); cp = c[t - 1] if t > 0 else np.zeros((N, H))
        dO = dh * tc
        d_c = dc + dh * o[t] * (1 - tc * tc)
        dA[t] = np.concatenate([d_c * g[t] * i[t] * (1 - i[t]), d_c * i[t] * (1 - g[t] ** 2), d_c * cp * f[t] * (1 - f[t]), dO * o[t] * (1 - o[t])], axis=1)
        dc = d_c * f[t]
        dh = dA[t] @ Wo
    return dA


def _bwd_case(H, MJ, T, N, seed):
    rng = np.random.default_rng(seed)
    i, f, o = (1 / (1 + np.exp(-rng.normal(size=(T, N, H)))) for _ in range(3))
    g = np.tanh(rng.normal(size=(T, N, H)))
    c = rng.normal(size=(T, N, H))
    dS, Wc = rng.normal(size=N), rng.normal(size=H)
    Wo = rng.normal(size=(4 * H, H)) * 0.2
    A0, A1, CF = forward_records(i, g, f, o, c, H)
    dA, dAT, db = run_bwd_model(A0, A1, CF, dS, Wc, Wo, T, N, H, MJ)
    ref = reference_bwd(i, g, f, o, c, dS, Wc, Wo, H)
    assert np.max(np.abs(dA - ref)) < 1e-10 * max(1.0, np.max(np.abs(ref)))
    assert np.max(np.abs(dAT - ref.transpose(2, 0, 1))) < 1e-10 * max(1.0, np.max(np.abs(ref)))
    assert np.max(np.abs(db - ref.sum(axis=(0, 1)))) < 1e-9 * max(1.0, np.max(np.abs(ref)))


def test_persistent_bptt_index_algebra_small():
    _bwd_case(H=128, MJ=1, T=3, N=128, seed=7)     # two tiles, NCH = 4: every formula with the same code path as the instantiated shape


def test_persistent_bptt_index_algebra_instantiated_shape():
    _bwd_case(H=384, MJ=3, T=2, N=64, seed=8)      # H = 384: three result tiles per wave, twelve chunks, six forward chunks
