// Fused persistent LSTM path kernels for gfx950 (D = H = 64, L <= 2: BASELINE config C2-A).
//
// Replaces, in one launch, the reference's
//   nn.SplitTable(3) -> FeatureEmbedding (3 LookupTables + CAddTable + JoinTable)    net/FeatureEmbedding.lua:112-121
//   -> nn.SplitTable(2) -> nn.Sequencer(nn.FastLSTM(D,H)) x L -> nn.SelectTable(-1)   model/OneModel.lua:223,236,268-274
//   -> nn.Linear(H,46)                                                                model/OneModel.lua:275
//
// Design (MI355X-first, see DESIGN.md "fused forward"):
//  * persistent workgroups, one per CU, each walks 64-path tiles; no [N,T,D] embedding tensor,
//    no per-step activation tensors in HBM (scoring); x_t rows are gathered straight from the
//    three tables into LDS one step ahead of use (loads issued before the MFMA block, LDS
//    write after it).
//  * 4 waves per workgroup; wave j owns hidden units [16j,16j+16) for ALL four
//    gates, so the LSTM cell math is lane-local on the MFMA accumulators (C/D layout
//    col = lane&15, row = 4*(lane>>4)+reg) and c_t never leaves registers.
//  * the 4-gate GEMM runs on v_mfma_f32_16x16x4_f32 (exact fp32).  Each wave keeps ITS slice of
//    [W_i2g | W_o2g] (4 gates x 16 cols x K=128 = 128 VGPRs) register-stationary for the whole
//    launch: weights are read from HBM/L2 once per CU, not once per step.
//  * one wave per SIMD (the 512-entry unified VGPR/AGPR file is what makes the weights fit): the
//    layers of a step run back-to-back in the same waves, handing h_l over through LDS with one
//    s_barrier per layer per step.  (A 2-waves-per-SIMD layer-pipelined variant needs 128 weight
//    registers + accumulators inside 256 and spilled ~100-180 VGPRs: measured, rejected.)
//  * k-order trick: one ds_read_b128 of A[row][16S+4g..+3] feeds 4 consecutive MFMAs (slot g of
//    MFMA jj <-> k = 16S+4g+jj); the matching B fragment is one 16-byte load of the ROW-MAJOR
//    weight row, so no packed weight copy is needed.
#include "lstm_fused_common.h"

#ifdef KPRN_TIMING_PROBES
#define KPRN_PROBES_ON 1
#else
#define KPRN_PROBES_ON 0
#endif

namespace fused {

// ---- MFMA issue, hand-placed ---------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 as inline asm with the register FILES chosen here: the B operand (a register-
// stationary weight) is read straight from the accumulation half of the unified register file ("a"), the
// accumulators live in architectural VGPRs ("v") where the cell math can touch them.  Left to hipcc, the 256
// weight registers of the 2-layer kernel were parked in AGPRs but staged through v_accvgpr_read in front of
// most MFMAs; a staging copy that feeds the very next MFMA costs +16 cycles per 32-cycle MFMA
// (scripts/ubench/mfma_rate.hip: 48.0 vs 32.07 ticks/MFMA), which is where the old kernel's 30 % went.
constexpr float NLOG2E = -1.4426950408889634f;   // sigmoid(x) = rcp(1 + exp2(-x log2 e))
constexpr float N2LOG2E = -2.8853900817779268f;  // tanh(x) = 2 rcp(1 + exp2(-2x log2 e)) - 1

// The LSTM cell of ONE accumulator register (r) of the previous unit, cut into 16 small steps; step K is issued
// behind MFMA K of a 16-MFMA group.  With one wave per SIMD nothing else hides the cell: a VALU op issues in
// the shadow of the running MFMA only if it does not wait on the op right before it, so each step holds at
// most one transcendental (16 cycles) and never consumes a value produced in the same step.
struct CellRegs { float m0, m1, m2, m3, e0, e1, e2, e3, i, g, f, o, ig, c, cp, t; };
template <bool SAVE, int R, int K>
__device__ __forceinline__ void cell_step(CellRegs& x, const f32x4 (&acc)[4], float (&cst)[4], float* out_row, f32x4 (&sv)[NPL]) {
  // The accumulators hold the pre-activations ALREADY scaled for exp2 (-log2 e for the sigmoid gates, -2 log2 e for the tanh
  // gate: the kernel folds the factors into its register copies of the weights and the bias), so no multiply is spent here.
  // They are read gate 0 first: the MFMAs that wrote gates 2, 3 last are the most recent ones.
  if (K == 0) { x.m0 = acc[0][R]; x.m1 = acc[1][R]; }
  if (K == 1) { x.e0 = __builtin_amdgcn_exp2f(x.m0); x.m2 = acc[2][R]; }
  if (K == 2) { x.e1 = __builtin_amdgcn_exp2f(x.m1); x.m3 = acc[3][R]; }
  if (K == 3) { x.e2 = __builtin_amdgcn_exp2f(x.m2); x.e0 += 1.0f; }
  if (K == 4) { x.e3 = __builtin_amdgcn_exp2f(x.m3); x.e1 += 1.0f; }
  if (K == 5) { x.i = __builtin_amdgcn_rcpf(x.e0); x.e2 += 1.0f; }
  if (K == 6) { x.g = __builtin_amdgcn_rcpf(x.e1); x.e3 += 1.0f; }
  if (K == 7) { x.f = __builtin_amdgcn_rcpf(x.e2); x.g = 2.0f * x.g - 1.0f; x.cp = cst[R]; }
  if (K == 8) { x.o = __builtin_amdgcn_rcpf(x.e3); x.ig = x.i * x.g; }
  if (K == 9) { x.c = x.f * x.cp + x.ig; }
  if (K == 10) { x.t = x.c * N2LOG2E; cst[R] = x.c; }
  if (K == 11) { x.t = __builtin_amdgcn_exp2f(x.t); }
  if (K == 12) { x.t += 1.0f; }
  if (K == 13) { x.t = __builtin_amdgcn_rcpf(x.t); }
  if (K == 14) { x.t = 2.0f * x.t - 1.0f; }
  if (K == 15) {
    const float hh = x.o * x.t;
    out_row[R * LDA] = hh;
    if (SAVE) {  // backward-ready factors (lstm_fused_common.h, NPL)
#ifdef KPRN_EXP_NOFACT
      sv[0][R] = x.i; sv[1][R] = x.g; sv[2][R] = x.cp; sv[3][R] = x.o; sv[4][R] = x.t; sv[5][R] = x.f; sv[6][R] = hh;
#else
      // each factor as ONE fused multiply-add (six VALU instructions per element where the products of (1 - x) took twelve)
      const float cf = x.cp * x.f;
      sv[0][R] = __builtin_fmaf(-x.ig, x.i, x.ig);   // i g (1 - i)
      sv[1][R] = __builtin_fmaf(-x.ig, x.g, x.i);    // i (1 - g^2)
      sv[2][R] = __builtin_fmaf(-cf, x.f, cf);       // c_{t-1} f (1 - f)
      sv[3][R] = __builtin_fmaf(-hh, x.o, hh);       // h (1 - o)        (h = o tanh c)
      sv[4][R] = __builtin_fmaf(-hh, x.t, x.o);      // o (1 - tanh^2 c)
      sv[5][R] = x.f;
      sv[6][R] = hh;
#endif
    }
  }
}

template <bool SAVE, int R>
__device__ __forceinline__ void cell_q(const f32x4 (&acc)[4], float (&cst)[4], float* out_row, f32x4 (&sv)[NPL]) {
  CellRegs x;
  cell_step<SAVE, R, 0>(x, acc, cst, out_row, sv);  cell_step<SAVE, R, 1>(x, acc, cst, out_row, sv);
  cell_step<SAVE, R, 2>(x, acc, cst, out_row, sv);  cell_step<SAVE, R, 3>(x, acc, cst, out_row, sv);
  cell_step<SAVE, R, 4>(x, acc, cst, out_row, sv);  cell_step<SAVE, R, 5>(x, acc, cst, out_row, sv);
  cell_step<SAVE, R, 6>(x, acc, cst, out_row, sv);  cell_step<SAVE, R, 7>(x, acc, cst, out_row, sv);
  cell_step<SAVE, R, 8>(x, acc, cst, out_row, sv);  cell_step<SAVE, R, 9>(x, acc, cst, out_row, sv);
  cell_step<SAVE, R, 10>(x, acc, cst, out_row, sv); cell_step<SAVE, R, 11>(x, acc, cst, out_row, sv);
  cell_step<SAVE, R, 12>(x, acc, cst, out_row, sv); cell_step<SAVE, R, 13>(x, acc, cst, out_row, sv);
  cell_step<SAVE, R, 14>(x, acc, cst, out_row, sv); cell_step<SAVE, R, 15>(x, acc, cst, out_row, sv);
}

// One k-group: 16 MFMAs (k-slots jj x gates q) on the A fragment a4 and the weights w[q][S].  BIAS: these are
// the first MFMAs of the unit's accumulation chains (srcC = bias).  The A fragment of the NEXT group is
// requested half way through (a read placed first would wait for the previous group's last MFMA to pick up
// the register it overwrites).  CELL: one cell_step of the previous unit behind every MFMA; the order is
// pinned with sched_barrier (the asm MFMAs carry no latency the scheduler could reason about).
// ST (training, a half without a cell): the 7 planes of the unit whose cell ran in the half before leave here, ONE store per 8 MFMAs.  A store holds its wave
// for the 16 cycles its 1 KiB takes on the CU's 64 B/clk store path, and the four waves -- in step behind a barrier -- used to issue the 7 back to back at the
// same moment: 4 x 7 x 16 cycles during which every wave stood still for most of the time (round 6, knock-out builds: the stores were 7.4 % of the launch).
// Spread out, the first one after a barrier still collides and leaves the waves 16 cycles apart; none of the later ones does.
template <bool SAVE, bool CELL, bool BIAS, bool PF, int S, bool ST = false>
__device__ __forceinline__ void k_group(const f32x4 a4, f32x4& apre, const float* next_addr, const f32x4 (&w)[4][4], const f32x4 (&bias4)[4],
                                        f32x4 (&acc)[4], CellRegs& x, const f32x4 (&pacc)[4], float (&pc)[4], float* pout_row, f32x4 (&sv)[NPL],
                                        gchar* fb = nullptr) {
#define KPRN_G1(K)                                                                                  \
  {                                                                                                 \
    constexpr int jj = (K) >> 2, q = (K) & 3;                                                       \
    if (BIAS && jj == 0) KPRN_MFMA_C(acc[q], a4[jj], w[q][S][jj], bias4[q]);                        \
    else KPRN_MFMA(acc[q], a4[jj], w[q][S][jj]);                                                    \
    if (PF && (K) == 7) apre = *(const f32x4*)(next_addr);                                          \
    if (CELL) { cell_step<SAVE, S, (K)>(x, pacc, pc, pout_row, sv); __builtin_amdgcn_sched_barrier(0); } \
    if (ST && ((K) == 3 || (K) == 11) && 2 * S + ((K) == 11) < NPL) {                               \
      constexpr int pk = 2 * S + ((K) == 11);                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                            \
      *(gf32x4*)(fb + pk * 1024) = sv[pk < NPL ? pk : 0];                                           \
      __builtin_amdgcn_sched_barrier(0);                                                            \
    }                                                                                               \
  }
  KPRN_G1(0) KPRN_G1(1) KPRN_G1(2) KPRN_G1(3) KPRN_G1(4) KPRN_G1(5) KPRN_G1(6) KPRN_G1(7)
  KPRN_G1(8) KPRN_G1(9) KPRN_G1(10) KPRN_G1(11) KPRN_G1(12) KPRN_G1(13) KPRN_G1(14) KPRN_G1(15)
#undef KPRN_G1
}

// Half of a unit's 4-gate GEMM: 4 k-groups over one LDS tile (the recurrent h_{t-1} tile or the step-input
// tile).  apre always holds the A fragment of the group about to run.
template <bool SAVE, bool CELL, bool BIAS, bool PF, bool ST = false>
__device__ __forceinline__ void half_unit(const float* abase, const f32x4 (&w)[4][4], const f32x4 (&bias4)[4], f32x4 (&acc)[4], f32x4& apre,
                                          const float* next_abase, const f32x4 (&pacc)[4], float (&pc)[4], float* pout_row, f32x4 (&sv)[NPL],
                                          gchar* fb = nullptr) {
  static_assert(!(ST && CELL), "the planes leave in the half behind the one that forms them");
  CellRegs x;
  f32x4 a4 = apre;
  k_group<SAVE, CELL, BIAS, true, 0, ST>(a4, apre, abase + 16, w, bias4, acc, x, pacc, pc, pout_row, sv, fb);
  a4 = apre;
  k_group<SAVE, CELL, false, true, 1, ST>(a4, apre, abase + 32, w, bias4, acc, x, pacc, pc, pout_row, sv, fb);
  a4 = apre;
  k_group<SAVE, CELL, false, true, 2, ST>(a4, apre, abase + 48, w, bias4, acc, x, pacc, pc, pout_row, sv, fb);
  a4 = apre;
  k_group<SAVE, CELL, false, PF, 3, ST>(a4, apre, next_abase, w, bias4, acc, x, pacc, pc, pout_row, sv, fb);
}

template <bool SAVE>
__device__ __forceinline__ void cell_all(const f32x4 (&pacc)[4], float (&pc)[4], float* pout_row, f32x4 (&sv)[NPL]) {
  cell_q<SAVE, 0>(pacc, pc, pout_row, sv);
  cell_q<SAVE, 1>(pacc, pc, pout_row, sv);
  cell_q<SAVE, 2>(pacc, pc, pout_row, sv);
  cell_q<SAVE, 3>(pacc, pc, pout_row, sv);
}

// nn.Linear(H, C) on the tile's h_T (LDS) -> S[n][0..C)
// Round 6, per-phase cycle counters: this was 10-11 k cycles per tile (4-5 % of the launch) for 64 MFMAs -- a chain of round trips, not arithmetic: the weight
// fragments and every row's output position (perm[n]) were loaded under conditions (hipcc puts such a load AND its wait inside a branch: DESIGN.md 3.4c), the
// positions one by one in front of their stores, and each m-tile's 16 MFMAs chained on one accumulator.  Now: every load is unconditional from a clamped address
// and issued up front (positions first: they are needed last), the four m-tiles' chains are interleaved.
template <int NMT>
__device__ __forceinline__ void head_tile(const FwdArgs& a, const float* hbuf, int64_t tile, int j, int lane) {
  const int ntiles = (a.C + 15) >> 4;
  const int arow = lane & 15, ag = lane >> 4;
  int pr[NMT][4];   // output row of the tile's rows 16 mt + 4 ag + r
#pragma unroll
  for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t n = tile * (16 * NMT) + mt * 16 + ag * 4 + r;
      const int64_t nc = n < a.N ? n : a.N - 1;
      pr[mt][r] = a.perm ? a.perm[nc] : (int)nc;   // (uniform branch; the load itself is unconditional)
    }
  for (int nt = j; nt < ntiles; nt += 4) {
    const int col = nt * 16 + arow;
    const bool cv = col < a.C;
    const int colc = cv ? col : a.C - 1;   // (columns past C compute a copy of the last one and are not stored)
    const float b = a.bout[colc];
    f32x4 w4[4];
#pragma unroll
    for (int S = 0; S < 4; ++S) w4[S] = *(const f32x4*)(a.Wout + (int64_t)colc * DH + S * 16 + ag * 4);
    f32x4 acc[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) acc[mt] = f32x4{b, b, b, b};
#pragma unroll
    for (int S = 0; S < 4; ++S) {
      f32x4 a4[NMT];
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) a4[mt] = *(const f32x4*)(hbuf + (mt * 16 + arow) * LDA + S * 16 + ag * 4);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][jj], w4[S][jj], acc[mt], 0, 0, 0);
    }
    if (cv) {
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t n = tile * (16 * NMT) + mt * 16 + ag * 4 + r;
          if (n < a.N) a.S[(int64_t)pr[mt][r] * a.C + col] = acc[mt][r];
        }
    }
  }
}

// NMT: 16-row m-tiles of a tile -- 4 (64-path tiles), or 1 for small batches (fused::small_tiles: four times as many workgroups, each a quarter
// of the latency; the unit pipeline below is the same, a slot is then the chain of L units (layer l, m-tile 0))
// bx / G_: this workgroup's index among, and the number of, the workgroups that walk THIS pass's tiles (k_lstm_fwd: the launch's; k_lstm_fwd_dual: a part of it)
template <int L, bool SAVE, int NMT>
__device__ __forceinline__ void fwd_body(const FwdArgs& a, const int bx, const int G_) {
  constexpr int NT = 256, MTR = 16 * NMT;
  static_assert(NMT == 4 || (NMT == 1 && L == 2), "the accumulator ping-pong needs an even number of units per slot");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // LDS carve (floats): x double buffer | h(layer l) double buffer, l = 0..L-1
  auto xbuf = [&](int i) -> float* { return lds + i * (MT * LDA); };
  auto hbuf = [&](int g, int i) -> float* { return lds + (2 + 2 * g + i) * (MT * LDA); };
  // id tiles (double-buffered by tile parity): [64][T][4] ints each
  auto idbuf = [&](int i) -> int32_t* { return (int32_t*)(lds + (2 + 2 * L) * (MT * LDA)) + i * (MT * MAXT_LDS * 4); };
  // prefix table [KCAP+1][L][PFB]: per class k the recurrent half of a tile's first executed step (W_o2g h_prefix(k)) and c_prefix(k)
  const float* pft = (const float*)idbuf(2);

  const int lane = threadIdx.x & 63;
  const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // hidden tile owned by this wave
  const int arow = lane & 15, ag = lane >> 4;
  unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = (KPRN_PROBES_ON && a.timing) ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long tstart = tlast;
#define FPROBE(slot_)                                              \
  if (KPRN_PROBES_ON && a.timing) {                                \
    const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
    tacc[slot_] += now__ - tlast;                                  \
    tlast = now__;                                                 \
  }

  // ---- register-stationary weights of EVERY layer: rows (q*H + 16j + arow), 16-byte pieces at k = 16S + 4ag
  f32x4 wi[L][4][4], wo[L][4][4];
  f32x4 bias4[L][4];
#pragma unroll
  for (int l = 0; l < L; ++l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t row = (int64_t)q * DH + j * 16 + arow;
      const float sc = (q == 1) ? N2LOG2E : NLOG2E;  // gate order i, g, f, o: g is the tanh gate
      const float bv = a.bi[l][row] * sc;
      bias4[l][q] = f32x4{bv, bv, bv, bv};
#pragma unroll
      for (int S = 0; S < 4; ++S) {
        wi[l][q][S] = *(const f32x4*)(a.Wi[l] + row * DH + S * 16 + ag * 4) * sc;
        wo[l][q][S] = *(const f32x4*)(a.Wo[l] + row * DH + S * 16 + ag * 4) * sc;
      }
    }
  }
  // B-operand-only values: pin them to the accumulation half of the unified register file so that the 256
  // architectural VGPRs stay free for everything the VALU touches (all loads are issued before the first pin:
  // a pin waits for its load)
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int S = 0; S < 4; ++S) {
        asm volatile("" : "+a"(wi[l][q][S]));
        asm volatile("" : "+a"(wo[l][q][S]));
      }
  FPROBE(6)  // (measurement build) the weights
  float c[L][NMT][4];
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int m = 0; m < NMT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[l][m][r] = 0.f;

  const int T = a.T;
  if ((int64_t)bx >= a.n_tiles) return;
  auto tile_k0 = [&](int64_t tl) -> int { return a.tile_k ? __builtin_amdgcn_readfirstlane(a.tile_k[tl]) : 0; };

  f32x4 gv[MTR * 16 / NT];
  const GatherSrc gsrc = gather_src(a);
  ids_stage_dma<NT, MTR>(a.idx, a.N, T, a.F, a.nT, bx, idbuf(0));
  if (a.tile_k) {  // classes 1 .. longest prefix of the batch (class 0 = no prefix: nothing to look up)
    const int n_cls = __builtin_amdgcn_readfirstlane(a.pmeta[0]) + 1;
    for (int c = L * PFB + threadIdx.x; c < n_cls * L * PFB; c += NT) ((float*)pft)[c] = a.pfb[c];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's id pieces have landed
  lds_barrier();
  FPROBE(11)  // ... first ids + prefix table
  int k0 = tile_k0(bx);  // the tile runs steps k0 .. T-1 (k0 <= T-2)
  gather_load_planes<NT, MTR>(a, gsrc, bx, k0, idbuf(0), gv);
  gather_store<NT, MTR>(xbuf(0), gv);

  // The work of a slot (one step t of one tile) is a chain of units u = (layer l, 16-row m-tile mt).  Unit u:
  //   [recurrent half: 64 MFMAs over h^l_{t-1}, with the CELL of unit u-1 interleaved]  (skipped at t == 0)
  //   [mt == 0: LDS barrier -- the tile this unit's input half reads is complete]
  //   [input half: 64 MFMAs over x_t / h^{l-1}_t]                                        (cell of u-1 here at t == 0)
  // so the cell math (40 transcendentals per lane per m-tile) and the barrier skew sit under MFMAs of the
  // next unit; accumulators ping-pong between two register sets.
  f32x4 accs[2][4];
  f32x4 apre = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 sv[NPL];
  const int64_t frag_unit = (int64_t)NPL * 256;           // floats per (m-tile, t, layer, wave)
  const int64_t frag_mt_stride = (int64_t)T * L * 4 * frag_unit;
  const int a_off = arow * LDA + ag * 4;                   // this lane's A-fragment offset inside a 16-row block
  const int o_off = (ag * 4) * LDA + j * 16 + arow;        // this lane's cell-output offset inside a 16-row block

  // A unit's planes sit at (its tile's region) + (an offset that fits 32 bits: m-tile, step, layer, wave, lane).  The region's address is wave-uniform and is
  // held in scalar registers (readfirstlane: hipcc kept the tile index and the strides in vector registers and formed every unit's address with 64-bit VALU
  // multiplies -- quarter rate, in a kernel whose VALU time nothing hides); the offset is one scalar term + lane * 16.
  const uint32_t mt_stride32 = (uint32_t)frag_mt_stride;   // floats; T <= 16, L <= 2: 4 m-tiles of it are < 4 MB
  auto tile_region = [&](int64_t tl) -> gchar* { return uniform_global(a.save_frag + tl * NMT * frag_mt_stride); };
  auto save_addr = [&](gchar* region, int p_t, int pl, int pm) -> gchar* {
    const uint32_t u = ((uint32_t)pm * mt_stride32 + (uint32_t)((p_t * L + pl) * 4 + j) * (uint32_t)frag_unit) * 4u;
    return region + (u + (uint32_t)lane * 16u);
  };
  auto save_unit = [&](gchar* region, int p_t, int pl, int pm) {
    if (!SAVE) return;
    gchar* fb = save_addr(region, p_t, pl, pm);
#pragma unroll
#ifdef KPRN_EXP_NOSTORE
    for (int k = 0; k < NPL; ++k) asm volatile("" ::"v"(sv[k]));
    if (a.T == 77) *(gf32x4*)(fb) = sv[0];
#else
    for (int k = 0; k < NPL; ++k) *(gf32x4*)(fb + k * 1024) = sv[k];
#endif
  };
  // FIRST: the tile's first executed step.  There is no recurrent half: the tile's common prefix state enters as one
  // extra k-slot per gate (A = 1 in k-slot 0, B = (W_o2g h_prefix)[col] in k-slot 0: adds that vector to every row) and
  // through c, set here to c_prefix.  Class 0 (no prefix): nothing to add, c = 0.
  auto slot = [&](auto first_tag, const int64_t tile, const int t, const int par, const bool has_prev, const int64_t p_tile, const int p_t,
                  const int cls) {
    constexpr bool FIRST = decltype(first_tag)::value;
    gchar* const reg_c = SAVE ? tile_region(tile) : nullptr;      // where this slot's tile, and the tile of the slot before, keep their planes
    gchar* const reg_p = SAVE ? tile_region(p_tile) : nullptr;
    float cinit[L];
    float rec0[L][4];  // k-slot 0 of the B operand: (W_o2g h_prefix)[gate q, col 16j + arow]
    const float one0 = (ag == 0) ? 1.0f : 0.f;
    if (FIRST) {
#pragma unroll
      for (int l = 0; l < L; ++l) {
        cinit[l] = 0.f;
        if (cls > 0) {  // (uniform)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float v = pft[(cls * L + l) * PFB + q * DH + j * 16 + arow];
            rec0[l][q] = (ag == 0) ? v * ((q == 1) ? N2LOG2E : NLOG2E) : 0.f;
          }
          cinit[l] = pft[(cls * L + l) * PFB + 4 * DH + j * 16 + arow];
        }
#pragma unroll
        for (int m = 0; m < NMT; ++m)
          if (!(l == L - 1 && m == NMT - 1)) {  // c[L-1][NMT-1] still belongs to the previous slot's last cell
#pragma unroll
            for (int r = 0; r < 4; ++r) c[l][m][r] = cinit[l];
          }
      }
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float* in_buf = (l == 0) ? xbuf(par) : hbuf(l - 1, par);
      const float* hp_buf = hbuf(l, par ^ 1);
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        // the unit whose cell is still outstanding
        const int pl = (mt > 0) ? l : ((l > 0) ? l - 1 : L - 1);
        const int pm = (mt > 0) ? mt - 1 : NMT - 1;
        const bool cross = (l == 0 && mt == 0);             // it belongs to the previous slot
        gchar* const q_reg = cross ? reg_p : reg_c;
        const int q_t = cross ? p_t : t;
        const int q_par = cross ? (par ^ 1) : par;
        float* pout = hbuf(pl, q_par) + pm * 16 * LDA + o_off;
        f32x4(&acc)[4] = accs[(l * NMT + mt) & 1];
        f32x4(&pacc)[4] = accs[((l * NMT + mt) & 1) ^ 1];
        const float* in_base = in_buf + mt * 16 * LDA + a_off;
        // first A fragment of the unit that follows this one (always a readable LDS address; unused when that
        // unit starts behind a barrier)
        const float* nxt;
        if (mt < NMT - 1) nxt = (FIRST ? in_buf : hp_buf) + (mt + 1) * 16 * LDA + a_off;
        else if (l + 1 < L) nxt = hbuf(l + 1, par ^ 1) + a_off;
        else nxt = hbuf(0, par) + a_off;
        if (!FIRST) {
          if (L == 1 && mt == 0) {
            // One layer: this recurrent half reads rows 0-15 of h_{t-1}, which the four waves wrote behind the previous slot's only barrier (the
            // interleaved cell of that slot's unit (0, 1)) -- with two layers the upper layer's barrier lies in between.  Nothing else orders the
            // waves here, and they do drift: in a tile's first slot waves 0-2 run the head of the tile before (46 classes = 3 column tiles) and
            // wave 3 does not, so wave 3 arrived a whole slot early and read rows the others had not written yet (wrong scores, run to run, in
            // the second and later tiles of a workgroup: found in round 6, batches of more than 256 tiles).  The fragment the previous slot
            // prefetched was that same early read.
            lds_barrier();
            apre = *(const f32x4*)(hp_buf + a_off);
          }
          half_unit<SAVE, true, true, true>(hp_buf + mt * 16 * LDA + a_off, wo[l], bias4[l], acc, apre, in_base, pacc, c[pl][pm], pout, sv);
          if (mt == 0) {
            lds_barrier();
            apre = *(const f32x4*)(in_base);
          }
          // (training: the planes of the unit whose cell just ran leave under this half, one store per 8 MFMAs)
#ifdef KPRN_EXP_BURST
          save_unit(q_reg, q_t, pl, pm);
          half_unit<SAVE, false, false, true>(in_base, wi[l], bias4[l], acc, apre, nxt, pacc, c[pl][pm], pout, sv);
#else
          half_unit<SAVE, false, false, true, SAVE>(in_base, wi[l], bias4[l], acc, apre, nxt, pacc, c[pl][pm], pout, sv,
                                                    SAVE ? save_addr(q_reg, q_t, pl, pm) : nullptr);
#endif
        } else if (mt == 0) {
          if (!cross || has_prev) {
            KPRN_MFMA_DRAIN();  // last MFMAs of the previous unit -> VALU reads
            KPRN_PIN_V4(pacc);
            cell_all<SAVE>(pacc, c[pl][pm], pout, sv);
            save_unit(q_reg, q_t, pl, pm);
          }
          if (cross) {
#pragma unroll
            for (int r = 0; r < 4; ++r) c[L - 1][NMT - 1][r] = cinit[L - 1];
          }
          if (cross) { FPROBE(8) }   // (measurement build) first slot: prefix lookup + the pending cell of the tile before
          lds_barrier();
          if (cross) { FPROBE(9) }   // ... barrier
          if (cross && has_prev) head_tile<NMT>(a, hbuf(L - 1, q_par), p_tile, j, lane);
          if (cross) { FPROBE(10) }  // ... the head of the tile before
          apre = *(const f32x4*)(in_base);
          half_unit<SAVE, false, true, true>(in_base, wi[l], bias4[l], acc, apre, nxt, pacc, c[pl][pm], pout, sv);
          if (cls > 0) {  // (uniform)
#pragma unroll
            for (int q = 0; q < 4; ++q) KPRN_MFMA_VV(acc[q], one0, rec0[l][q]);
          }
        } else {
          half_unit<SAVE, true, true, true>(in_base, wi[l], bias4[l], acc, apre, nxt, pacc, c[pl][pm], pout, sv);
          save_unit(q_reg, q_t, pl, pm);
          if (cls > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) KPRN_MFMA_VV(acc[q], one0, rec0[l][q]);
          }
        }
      }
    }
  };

  FPROBE(0)  // prologue: weights, first ids, first gather
  int64_t tile = bx;
  int t = k0;
  int tpar = 0;  // parity of the tile's id buffer
  int64_t p_tile = tile;
  int p_t = t;
  int par = 0;
  const int G = __builtin_amdgcn_readfirstlane(G_);   // (kept in a register: hipcc re-read gridDim from the dispatch packet, with the wait, at every tile boundary)
  int k0_req = 0;   // the next tile's first step, requested a slot or more ahead of the tile boundary (per lane; made uniform there)
  for (int64_t s = 0;; ++s) {
    par = (int)(s & 1);
    // (1) issue the gather for the NEXT slot (latency hidden under this slot's MFMAs)
    int tn = t + 1;
    int64_t tile_n = tile;
    int tpar_n = tpar;
    int k0_n = k0;
    if (tn == T) {
      tile_n += G;
      tpar_n ^= 1;
      k0_n = (tile_n < a.n_tiles) ? __builtin_amdgcn_readfirstlane(k0_req) : 0;
      tn = k0_n;
    }
    const bool have_next = tile_n < a.n_tiles;
    // the next tile's ids are requested (LDS-DMA) while this tile's first step computes.  They are read at the top of the tile's LAST slot: by then every
    // wave has passed the counted wait of this slot's gather_store (its pieces have landed) and at least one barrier of a later slot -- unless the tile
    // has only two steps: then the last slot is the next one, and the waves meet here first.
    if (t == k0 && tile + G < a.n_tiles) {
      ids_stage_dma<NT, MTR>(a.idx, a.N, T, a.F, a.nT, tile + G, idbuf(tpar ^ 1));
      k0_req = a.tile_k ? a.tile_k[tile + G] : 0;   // (a load whose wait sits at the tile boundary, not here)
    }
    if (have_next && tile_n != tile && t == k0 + 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
    }
    if (have_next) gather_load_planes<NT, MTR>(a, gsrc, tile_n, tn, idbuf(tpar_n), gv);
    FPROBE(1)  // id staging + gather issue
    // (2) the units of this slot
    if (t == k0) { slot(std::true_type{}, tile, t, par, s > 0, p_tile, p_t, k0); FPROBE(2) }
    else { slot(std::false_type{}, tile, t, par, true, p_tile, p_t, 0); FPROBE(3) }
    // (3) land the gathered rows of the next slot (xbuf[par^1] was last read one slot ago); visible to the
    //     other waves after the next slot's first barrier
    // the last unit's accumulators stay pending across the loop back-edge: keep hipcc from touching them (phi
    // copies, spills) before the MFMAs that wrote them have landed
    KPRN_MFMA_DRAIN();
    KPRN_PIN_V4(accs[0]);
    KPRN_PIN_V4(accs[1]);
    if (have_next) gather_store<NT, MTR>(xbuf(par ^ 1), gv);
    FPROBE(4)  // landing the gathered rows (waits for the loads -- and, when saving, for the stores in flight)
    p_tile = tile; p_t = t;
    if (!have_next) break;
    t = tn; tile = tile_n; tpar = tpar_n; k0 = k0_n;
  }
  // drain: the cell of the very last unit, then the last tile's head
  {
    KPRN_MFMA_DRAIN();
    constexpr int LAST = (L * NMT - 1) & 1;   // accumulator set of a slot's last unit
    KPRN_PIN_V4(accs[LAST]);
    cell_all<SAVE>(accs[LAST], c[L - 1][NMT - 1], hbuf(L - 1, par) + (NMT - 1) * 16 * LDA + o_off, sv);
    save_unit(SAVE ? tile_region(p_tile) : nullptr, p_t, L - 1, NMT - 1);
    lds_barrier();
    head_tile<NMT>(a, hbuf(L - 1, par), p_tile, j, lane);
  }
  FPROBE(5)  // drain
  if (KPRN_PROBES_ON && a.timing && threadIdx.x == 0) {
    tacc[7] = __builtin_amdgcn_s_memtime() - tstart;
    for (int k = 0; k < 12; ++k) a.timing[(int64_t)bx * 12 + k] = tacc[k];
  }
#undef FPROBE
}

template <int L, bool SAVE, int NMT>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd(FwdArgs a) { fwd_body<L, SAVE, NMT>(a, (int)blockIdx.x, (int)gridDim.x); }

// The training forward AND a scoring pass in ONE launch (option "score_dual"): workgroups [0, g0) are the training forward's, the rest the pass's -- two
// branches of one kernel.  The dispatcher places the first g0 on the chip and hands the pass's workgroups the CUs as those retire: what the two streams
// of "score_overlap" do, without the fork / join events between the streams (6-8 us of idle queue each: profiles/r06/kernel_timelines.txt).
template <int L, int NMT>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd_dual(FwdArgs a0, FwdArgs a1, int g0) {
  if ((int)blockIdx.x < g0) fwd_body<L, true, NMT>(a0, (int)blockIdx.x, g0);
  else fwd_body<L, false, NMT>(a1, (int)blockIdx.x - g0, (int)gridDim.x - g0);
}

// ---- host side ----
// Small batches: below ~8 k paths the 64-path tiles leave most CUs idle and a step is the latency of ONE tile through the persistent launches
// (0.38 ms whatever the size).  Tiles of one 16-row m-tile put 4x the workgroups on the chip at a quarter of the latency each.
// No identical-prefix plan in this mode (kprn_api.hip batch_wants_plan agrees): its classes are per 64-path tile.
bool small_tiles(const kprn_handle* h, int64_t N, bool has_plan) {
  static const int64_t max_paths = KPRN_DEV_ENV("KPRN_SMALL_TILES_MAX") ? atoll(KPRN_DEV_ENV("KPRN_SMALL_TILES_MAX")) : SMALL_TILES_MAX_PATHS;   // (measurement: the 16-row tiles at any size)
  return h->small_tiles_on && !has_plan && h->cfg.compute_dtype == 0 && h->cfg.L == 2 && N <= max_paths;
}

// The hand-over context of the launch about to be queued on h->stream: slots + flags of that stream's own set (the scoring pass runs beside the
// training forward on its stream; launches of one stream are ordered, so they share a set under distinct epochs), a fresh epoch.  epoch 0 (off) when the
// option is off, or for a stream that is neither the engine's own nor the scoring stream as a whole-batch launch.
HoArgs handover_args(kprn_handle* h, int grid) {
  HoArgs ho;
  State* s = st(h);
  if (!h->tile_handover || grid < 2 || grid > s->num_cu) return ho;
  const int ctx = (h->score_stream && h->stream == h->score_stream) ? 1 : 0;
  if (!s->ho_state[ctx]) {
    HIP_TRY(kprn_dev_malloc((void**)&s->ho_state[ctx], (size_t)s->num_cu * HO_STATE * sizeof(float)));
    HIP_TRY(kprn_dev_malloc((void**)&s->ho_flag[ctx], (size_t)s->num_cu * sizeof(unsigned)));
    HIP_TRY(hipMemsetAsync(s->ho_flag[ctx], 0, (size_t)s->num_cu * sizeof(unsigned), h->stream));   // (ordered before the first launch that reads them)
  }
  if (!h->ho_fault) {
    HIP_TRY(hipHostMalloc((void**)&h->ho_fault, 64, hipHostMallocDefault));
    *h->ho_fault = 0;
  }
  if (++s->ho_epoch == 0) ++s->ho_epoch;
  ho.state = s->ho_state[ctx]; ho.flag = s->ho_flag[ctx]; ho.epoch = s->ho_epoch; ho.fault = h->ho_fault; ho.mode = h->tile_handover;
  return ho;
}

// what the hand-over rule does with this batch on the fused kernels' grid (diagnostics, tests, bench.py): evaluated on the host from the device's tile_k
void handover_stats(kprn_handle* h, const kprn_batch* b, int64_t* out) {
  State* s = st(h);
  const int64_t N = (int64_t)b->B * b->P;
  const bool small = small_tiles(h, N, b->tile_k != nullptr);
  const int64_t n_tiles = small ? (N + 15) / 16 : (N + MT - 1) / MT;
  const int G = (int)std::min<int64_t>(n_tiles, (int64_t)s->num_cu);
  std::vector<int32_t> tk((size_t)n_tiles, 0);
  if (b->tile_k) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(tk.data(), b->tile_k, (size_t)n_tiles * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  out[0] = out[1] = out[2] = out[3] = 0;
  const bool on = h->tile_handover && G >= 2 && fwd_supported(h, b->T) && h->cfg.compute_dtype == 0;
  for (int bx = 0; bx < G; ++bx) {
    int load = 0;
    const HoPlan p = ho_plan_host(on, h->tile_handover, G, bx, n_tiles, b->T, [&](int64_t tl) -> int { return tk[(size_t)tl]; }, &load);
    out[2] = std::max<int64_t>(out[2], load);
    if (p.role == 1) { out[0] += 1; out[1] += p.d; load -= 2 * p.d; }
    else if (p.role == 2) load += 2 * p.d;
    out[3] = std::max<int64_t>(out[3], load);
  }
}

bool fwd_supported(const kprn_handle* h, int T) {
  const kprn_config& c = h->cfg;
  return (h->D == DH && c.H == DH && c.L >= 1 && c.L <= 2 && (c.dt % 4) == 0 && (c.de % 4) == 0 && (c.dr % 4) == 0 && T >= 2 && T <= MAXT_LDS);
}

template <int L, bool SAVE, int NMT = 4>
static void launch_fwd(kprn_handle* h, const FwdArgs& a, int grid) {
  const size_t lds_bytes = (size_t)(2 + 2 * L) * MT * LDA * sizeof(float) + 2 * MT * MAXT_LDS * 4 * sizeof(int32_t) + (size_t)(KCAP + 1) * L * PFB * sizeof(float);
  static PerDeviceOnce attr_done;  // one per template instantiation
  if (attr_done.need()) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_lstm_fwd<L, SAVE, NMT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  }
  hipLaunchKernelGGL((k_lstm_fwd<L, SAVE, NMT>), dim3(grid), dim3(256), lds_bytes, h->stream, a);
  HIP_TRY(hipGetLastError());
}

// tile_begin / tile_end (scoring only, tile_end < 0 = all): the pass restricted to a range of the batch's 64-path tiles -- a scoring pass split
// in two around a data-parallel step's collective (kprn_set_option "score_split").  The kernel sees a shorter batch: the per-tile arrays are
// handed over shifted, scores still land at their path's own row of S.
// the arguments of one pass (S: where its scores go); false: nothing to launch (an empty tile range)
static bool fwd_args(kprn_handle* h, const kprn_batch* b, bool save, int64_t tile_begin, int64_t tile_end, bool ignore_reserve, float* S, FwdArgs& a, int& grid,
                     bool& small) {
  const kprn_config& c = h->cfg;
  State* s = st(h);
  const int64_t N = (int64_t)b->B * b->P;
  a.idx = b->idx_s ? b->idx_s : b->idx; a.N = N; a.T = b->T; a.F = b->F; a.nT = c.num_types;
  a.perm = b->perm; a.tile_k = b->tile_k; a.pmeta = b->pmeta; a.pfb = s->pfb;
  a.Wt = h->dense + h->off_Wt; a.We = h->We; a.Wr = h->dense + h->off_Wr;
  a.dt = c.dt; a.de = c.de; a.dr = c.dr;
  for (int l = 0; l < 2; ++l) {
    const int ll = l < c.L ? l : 0;
    a.Wi[l] = h->dense + h->layer[ll].Wi; a.bi[l] = h->dense + h->layer[ll].bi; a.Wo[l] = h->dense + h->layer[ll].Wo;
  }
  a.Wout = h->dense + h->off_outW; a.bout = h->dense + h->off_outb; a.C = c.C;
  a.S = S;
  small = small_tiles(h, N, b->tile_k != nullptr);   // (c.L == 2: the only small-tile instantiation)
  a.n_tiles = small ? (N + 15) / 16 : (N + MT - 1) / MT;
  if (!small && !save && (tile_begin > 0 || tile_end >= 0)) {
    const int64_t t0 = std::min<int64_t>(tile_begin, a.n_tiles), t1 = tile_end < 0 ? a.n_tiles : std::min<int64_t>(tile_end, a.n_tiles);
    if (t1 <= t0) return false;
    a.idx += t0 * MT * a.T * a.F;
    if (a.perm) a.perm += t0 * MT; else a.S += t0 * MT * (int64_t)c.C;
    if (a.tile_k) a.tile_k += t0;
    a.N = std::min<int64_t>(N, t1 * MT) - t0 * MT;
    a.n_tiles = t1 - t0;
  }
  a.save_frag = nullptr;
  if (save) {
    if (N > s->cap_N || b->T > s->cap_T) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (s->save_frag) hipFree(s->save_frag);
      const int64_t cn = std::max<int64_t>(N, s->cap_N);
      const int ct = std::max(b->T, s->cap_T);
      const int64_t mts = (cn + 15) / 16 + 4;
      HIP_TRY(kprn_dev_malloc((void**)&s->save_frag, (size_t)mts * ct * c.L * 4 * NPL * 256 * sizeof(float)));
      s->cap_N = cn; s->cap_T = ct;
    }
    a.save_frag = s->save_frag;
  }
  const int cus = (!save && h->reserve_cus > 0 && !ignore_reserve) ? std::max(1, s->num_cu - h->reserve_cus) : s->num_cu;
  grid = (int)std::min<int64_t>(a.n_tiles, (int64_t)cus);
  static const bool want_timing = KPRN_DEV_ENV("KPRN_TIMING") != nullptr;
  if (want_timing && !s->timing) HIP_TRY(kprn_dev_malloc((void**)&s->timing, (size_t)s->num_cu * 12 * sizeof(unsigned long long)));
  a.timing = s->timing;
  return true;
}

// The training forward of `bt` and a whole scoring pass over `bs` (scores to S_score) as one launch (k_lstm_fwd_dual); false: not applicable -- the caller
// launches them the usual way.  Both passes read ONE identical-prefix table: the two batches must be the same one, or neither may have a plan.
bool forward_dual(kprn_handle* h, const kprn_batch* bt, const kprn_batch* bs, float* S_score) {
  const kprn_config& c = h->cfg;
  State* s = st(h);
  if (c.compute_dtype != 0 || c.L != 2 || !fwd_supported(h, bt->T) || !fwd_supported(h, bs->T) || s->timing) return false;
  const bool same = bt->serial == bs->serial;
  const bool planless = (!bt->tile_k || bt->h_kmax == 0) && (!bs->tile_k || bs->h_kmax == 0);
  if (!same && !planless) return false;
  FwdArgs a0, a1;
  int g0 = 0, g1 = 0;
  bool small0 = false, small1 = false;
  prefix_forward(h, bt);
  if (h->score_rest_batch != bs) return false;   // (a rewrite of the prefix table joins the scoring stream first: the deferred pass has then run the usual way)
  if (!fwd_args(h, bt, true, 0, -1, false, h->ws.S, a0, g0, small0) || !fwd_args(h, bs, false, 0, -1, true, S_score, a1, g1, small1) || small0 != small1) return false;
  a0.timing = a1.timing = nullptr;
  const size_t lds_bytes = (size_t)(2 + 2 * 2) * MT * LDA * sizeof(float) + 2 * MT * MAXT_LDS * 4 * sizeof(int32_t) + (size_t)(KCAP + 1) * 2 * PFB * sizeof(float);
  ProfScope ps(h, "lstm_fused_fwd_dual");
  if (small0) {
    static PerDeviceOnce once;
    if (once.need()) HIP_TRY(hipFuncSetAttribute((const void*)k_lstm_fwd_dual<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL((k_lstm_fwd_dual<2, 1>), dim3(g0 + g1), dim3(256), lds_bytes, h->stream, a0, a1, g0);
  } else {
    static PerDeviceOnce once;
    if (once.need()) HIP_TRY(hipFuncSetAttribute((const void*)k_lstm_fwd_dual<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL((k_lstm_fwd_dual<2, 4>), dim3(g0 + g1), dim3(256), lds_bytes, h->stream, a0, a1, g0);
  }
  HIP_TRY(hipGetLastError());
  return true;
}

void forward(kprn_handle* h, const kprn_batch* b, bool save, int64_t tile_begin, int64_t tile_end, bool ignore_reserve) {
  const kprn_config& c = h->cfg;
  if (c.compute_dtype != 0) { forward_mc(h, b, save); return; }  // bf16 / f32x6: the matrix-core forward (lstm_fused_fwd_mc.hip)
  State* s = st(h);
  const int64_t N = (int64_t)b->B * b->P;
  FwdArgs a;
  int grid = 0;
  bool small = false;
  prefix_forward(h, b);  // (cached while neither the parameters nor the batch change)
  if (!fwd_args(h, b, save, tile_begin, tile_end, ignore_reserve, h->ws.S, a, grid, small)) return;
  ProfScope ps(h, save ? "lstm_fused_fwd_train" : "lstm_fused_fwd");
  if (c.L == 1) { if (save) launch_fwd<1, true>(h, a, grid); else launch_fwd<1, false>(h, a, grid); }
  else if (small) { if (save) launch_fwd<2, true, 1>(h, a, grid); else launch_fwd<2, false, 1>(h, a, grid); }
  else { if (save) launch_fwd<2, true>(h, a, grid); else launch_fwd<2, false>(h, a, grid); }
  if (s->timing) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> tb((size_t)grid * 12);
    HIP_TRY(hipMemcpy(tb.data(), s->timing, tb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum[12] = {0};
    for (int g = 0; g < grid; ++g) for (int k = 0; k < 12; ++k) sum[k] += (double)tb[(size_t)g * 12 + k];
    fprintf(stderr, "[kprn timing] fwd save=%d N=%lld grid=%d avg cycles/WG: prologue %.0f (weights %.0f, first ids + prefix table %.0f, first gather %.0f) ids+gather-issue %.0f first-slots %.0f (of which, before their units: prefix + pending cell %.0f, barrier %.0f, head %.0f) rec-slots %.0f gather-land %.0f drain %.0f total %.0f\n",
            (int)save, (long long)N, grid, (sum[0] + sum[6] + sum[11]) / grid, sum[6] / grid, sum[11] / grid, sum[0] / grid, sum[1] / grid, (sum[2] + sum[8] + sum[9] + sum[10]) / grid, sum[8] / grid, sum[9] / grid, sum[10] / grid, sum[3] / grid, sum[4] / grid, sum[5] / grid, sum[7] / grid);
  }
}

}  // namespace fused
