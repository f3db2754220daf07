// Fused LSTM forward on the bf16 MATRIX CORES of gfx950 (D = H = 64), one launch per layer.
//
// Why a second forward: v_mfma_f32_16x16x4_f32 runs on the SIMD's fp32 lanes -- 32 cycles per issue and no overlap with VALU
// (DESIGN.md 3.0), so the fp32 forward (lstm_fused_fwd.hip) pays GEMM + cell math in sequence.  v_mfma_f32_16x16x32_bf16 runs
// on the matrix cores: 8x the flops per issue in half the cycles, and VALU instructions issued behind it execute beside it
// (scripts/ubench/mfma_bf16_overlap.hip: +8 cycles for exp2 + rcp behind a 17-cycle bf16 MFMA, +26 behind a 32-cycle fp32 one).
//
// compute_dtype (include/kprn.h):
//   1  bf16:   operands rounded to bf16, products exact, fp32 accumulation                      1 MFMA per K = 32
//   2  f32x6:  every fp32 operand is split EXACTLY into three bf16 pieces (x = x1 + x2 + x3, 3 x 8 mantissa bits); the six
//              partial products of weight >= 2^-16 relative to the leading one (11 12 21 13 22 31) are accumulated in fp32:
//              the dropped ones are below 2^-24.  Error <= that of an fp32 FMA chain (measured on a 256x128x256 product:
//              1.6e-7 of the largest result against 5.1e-7 for fp32 BLAS).                       6 MFMAs per K = 32
//   3  f32x3:  two fp16 pieces (2 x 11 mantissa bits) of every power-of-two pre-scaled operand, three products (11 12 21); what
//              is dropped is below 2^-22.                                                         3 MFMAs per K = 32
//
// Replaces, per layer, nn.Sequencer(nn.FastLSTM(D,H)) (model/OneModel.lua:268-274) [+ FeatureEmbedding gather, bottom layer;
// + nn.Linear(H,46), top layer]; same tiles, wave ownership (wave j: hidden units 16j..16j+15 of all four gates), C layout,
// cell math, training saves and identical-prefix plan as lstm_fused_fwd.hip, so the fp32 backward runs unchanged behind it.
// The split weights (3 x 2 B per element) of ONE layer fill 192 of the 256 accumulation registers; the layers therefore run
// as separate launches and hand h over through HBM ([tile][t][64 rows][64] fp32, 100 MB per pass at 65 536 paths).
//
// LDS: the step input (x_t / h^{l-1}_t) and h^l as one plane per piece, 2-byte elements [64 rows][72], double-buffered: an A fragment (row,
// k-group of 8) is one ds_read_b128; the producer of a value splits it once (the gather for x, the cell for h).
#include "lstm_fused_common.h"

#ifndef MC_EXP
#define MC_EXP 0  // timing experiments (wrong results): 1 no cell, 2 no MFMAs, 4 no h hand-over stores, 8 no barrier B, 16 no A-fragment reads
#endif

namespace fused {

// How an fp32 operand travels to the matrix cores (template parameter M of everything below):
//   M = 1  one bf16 piece (rounded), 1 product per K chunk                                                  compute_dtype 1
//   M = 3  three bf16 pieces (exact), 6 products: 11 12 21 13 22 31                                         compute_dtype 2
//   M = 2  two fp16 pieces (2 x 11 bits; what is dropped is below 2^-22), 3 products: 11 12 21.  fp16 has 5 exponent bits, so the
//          operands are pre-scaled by powers of two (exact): W_o2g by 2^KW, x by 2^KA, W_i2g by 2^(KW-KA) -- every product then
//          carries 2^KW, which the cell removes with one multiply per gate.  Residual pieces of values below 2^-3 / scale go
//          subnormal: an ABSOLUTE error of 2^-25 / scale, negligible against the operand's range.               compute_dtype 3
template <int M> struct McFmt;
template <> struct McFmt<1> { typedef __bf16 E; static constexpr int NP = 1, NTERM = 1, KW = 0, KA = 0; };
template <> struct McFmt<3> { typedef __bf16 E; static constexpr int NP = 3, NTERM = 6, KW = 0, KA = 0; };
template <> struct McFmt<2> { typedef _Float16 E; static constexpr int NP = 2, NTERM = 3, KW = 8, KA = 4; };
__device__ __host__ constexpr float mc_pow2(int k) { return k >= 0 ? (float)(1u << k) : 1.0f / (float)(1u << (-k)); }

constexpr int LDB = DH + 8;  // bf16 row stride of an LDS plane: 144 B (16-byte aligned, spreads the ds_read_b128 slots)
constexpr float MC_NLOG2E = -1.4426950408889634f;
constexpr float MC_N2LOG2E = -2.8853900817779268f;

#define MC_MFMA(ACC, A_, B_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A_), "a"(B_))
#define MC_MFMA_H(ACC, A_, B_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(A_), "a"(B_))
#define MC_MFMA_HC(ACC, A_, B_, C_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(ACC) : "v"(A_), "a"(B_), "v"(C_))
#define MC_DRAIN() asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory")
#define MC_MFMA_C(ACC, A_, B_, C_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(ACC) : "v"(A_), "a"(B_), "v"(C_))

struct McArgs {
  const int32_t* idx; int64_t N; int T, F, nT;
  const float *Wt, *We, *Wr; int dt, de, dr;
  int L, layer;
  const uint4* wsp;        // this layer's split weights in register order: [2 src][4 q][2 kc][NS][4 waves][64 lanes] x 16 B
  const float* bias_sc;    // [256] bias * (-log2 e | -2 log2 e)
  const float* Hin;        // layers above the bottom: h of the layer below, [tile][T][64][64] fp32
  float* Hout;             // layers below the top: this layer's h, same layout
  const float* Wout; const float* bout; int C; float* S;   // top layer: head
  const int32_t* perm; const int32_t* tile_k; const int32_t* pmeta; const float* pfb;  // identical-prefix plan (lstm_fused_prefix.hip)
  float* save_frag;        // training saves (nullable), layout of lstm_fused_common.h
  int64_t n_tiles;
  unsigned long long* timing;  // optional [grid][8] cycle counters (KPRN_TIMING=1)
};

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1 (every index of the MFMA stream must be a constant:
// register arrays, the cell step issued behind each MFMA)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// the partial products of a K chunk, (piece of A, piece of B), smallest first
__device__ constexpr int mc_ta(int M, int term) { return M == 3 ? (term == 0 ? 2 : (term == 1 || term == 3) ? 1 : 0) : M == 2 ? (term == 0 ? 1 : 0) : 0; }
__device__ constexpr int mc_tb(int M, int term) { return M == 3 ? (term == 2 ? 2 : (term == 1 || term == 4) ? 1 : 0) : M == 2 ? (term == 1 ? 1 : 0) : 0; }

template <int M>
__device__ __forceinline__ void split_store(float x, typename McFmt<M>::E* p, int plane_stride) {
  typedef typename McFmt<M>::E E;
  float r = x;
#pragma unroll
  for (int s = 0; s < McFmt<M>::NP; ++s) {
    const E piece = (E)r;
    p[s * plane_stride] = piece;
    r -= (float)piece;
  }
}

// ---- the cell of one accumulator register of the previous unit, in 16 small steps (same arithmetic as lstm_fused_fwd.hip) ----
struct McCell { float m0, m1, m2, m3, e0, e1, e2, e3, i, g, f, o, ig, c, cp, t; };
template <int M, bool SAVE, int R, int K>
__device__ __forceinline__ void mc_cell_step(McCell& x, const f32x4 (&acc)[4], float (&cst)[4], typename McFmt<M>::E* hrow, float* hout, f32x4 (&sv)[NPL]) {
  constexpr float INV = mc_pow2(-McFmt<M>::KW);  // (1 for the bf16 forms: no multiply is emitted)
  if (K == 0) { x.m0 = acc[0][R] * INV; x.m1 = acc[1][R] * INV; }  // (pre-activations arrive scaled for exp2)
  if (K == 1) { x.e0 = __builtin_amdgcn_exp2f(x.m0); x.m2 = acc[2][R] * INV; }
  if (K == 2) { x.e1 = __builtin_amdgcn_exp2f(x.m1); x.m3 = acc[3][R] * INV; }
  if (K == 3) { x.e2 = __builtin_amdgcn_exp2f(x.m2); x.e0 += 1.0f; }
  if (K == 4) { x.e3 = __builtin_amdgcn_exp2f(x.m3); x.e1 += 1.0f; }
  if (K == 5) { x.i = __builtin_amdgcn_rcpf(x.e0); x.e2 += 1.0f; }
  if (K == 6) { x.g = __builtin_amdgcn_rcpf(x.e1); x.e3 += 1.0f; }
  if (K == 7) { x.f = __builtin_amdgcn_rcpf(x.e2); x.g = 2.0f * x.g - 1.0f; x.cp = cst[R]; }
  if (K == 8) { x.o = __builtin_amdgcn_rcpf(x.e3); x.ig = x.i * x.g; }
  if (K == 9) { x.c = x.f * x.cp + x.ig; }
  if (K == 10) { x.t = x.c * MC_N2LOG2E; cst[R] = x.c; }
  if (K == 11) { x.t = __builtin_amdgcn_exp2f(x.t); }
  if (K == 12) { x.t += 1.0f; }
  if (K == 13) { x.t = __builtin_amdgcn_rcpf(x.t); }
  if (K == 14) { x.t = 2.0f * x.t - 1.0f; }
  if (K == 15) {
    const float hh = x.o * x.t;
    split_store<M>(hh, hrow + R * LDB, MT * LDB);
    if (hout) hout[R * DH] = hh;
    if (SAVE) {
      sv[0][R] = x.ig * (1.0f - x.i);
      sv[1][R] = x.i * (1.0f - x.g * x.g);
      sv[2][R] = x.cp * x.f * (1.0f - x.f);
      sv[3][R] = hh * (1.0f - x.o);
      sv[4][R] = x.o * (1.0f - x.t * x.t);
      sv[5][R] = x.f;
      sv[6][R] = hh;
    }
  }
}
template <int M, bool SAVE>
__device__ __forceinline__ void mc_cell_all(const f32x4 (&acc)[4], float (&cst)[4], typename McFmt<M>::E* hrow, float* hout, f32x4 (&sv)[NPL]) {
  McCell x;
  static_for<0, 64>([&](auto ic) __attribute__((always_inline)) {
    constexpr int n = decltype(ic)::value;
    mc_cell_step<M, SAVE, (n >> 4), (n & 15)>(x, acc, cst, hrow, hout, sv);
  });
}

// nn.Linear(H, C) on the tile's h_T (LDS planes; fp32 = exact sum of the pieces) -> S[n][0..C), fp32 MFMA (1 % of the work)
template <int M>
__device__ __forceinline__ void mc_head_tile(const McArgs& a, const typename McFmt<M>::E* hpl, int64_t tile, int j, int lane) {
  typedef typename McFmt<M>::E E;
  const int ntiles = (a.C + 15) >> 4;
  const int arow = lane & 15, ag = lane >> 4;
  for (int nt = j; nt < ntiles; nt += 4) {
    const int col = nt * 16 + arow;
    const bool cv = col < a.C;
    const float b = cv ? a.bout[col] : 0.f;
    f32x4 w4[4];
#pragma unroll
    for (int S = 0; S < 4; ++S) w4[S] = cv ? *(const f32x4*)(a.Wout + (int64_t)col * DH + S * 16 + ag * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      f32x4 acc = f32x4{b, b, b, b};
#pragma unroll
      for (int S = 0; S < 4; ++S) {
        const E* p = hpl + (mt * 16 + arow) * LDB + S * 16 + ag * 4;
        f32x4 a4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = (float)p[e];
#pragma unroll
          for (int s2 = 1; s2 < McFmt<M>::NP; ++s2) v += (float)p[s2 * MT * LDB + e];
          a4[e] = v;
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], w4[S][jj], acc, 0, 0, 0);
      }
      if (cv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t n = tile * MT + mt * 16 + ag * 4 + r;
          if (n < a.N) a.S[(a.perm ? (int64_t)a.perm[n] : n) * a.C + col] = acc[r];
        }
      }
    }
  }
}

template <int M, bool SAVE>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd_mc(McArgs a) {
  typedef McFmt<M> F;
  typedef typename F::E E;                   // 2-byte element of the pieces (bf16 or fp16)
  typedef E Ex4 __attribute__((ext_vector_type(4)));
  constexpr int NS = F::NP;                  // pieces per operand
  constexpr int NT = 256;
  constexpr int PLANE = MT * LDB;            // bf16 elements per plane
  constexpr int NTERM = F::NTERM;            // partial products per K chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];
  E* planes = (E*)lds;
  auto inb = [&](int i) -> E* { return planes + (i * NS) * PLANE; };          // step input, buffer i
  auto hb = [&](int i) -> E* { return planes + ((2 + i) * NS) * PLANE; };     // this layer's h, buffer i
  int32_t* idb0 = (int32_t*)(planes + 4 * NS * PLANE);
  auto idbuf = [&](int i) -> int32_t* { return idb0 + i * (MT * MAXT_LDS * 4); };
  float* pft = (float*)idbuf(2);             // [KCAP+1][PFB] this layer's prefix classes

  const int lane = threadIdx.x & 63;
  const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int arow = lane & 15, ag = lane >> 4;
  const int T = a.T, L = a.L, ly = a.layer;
  const bool bottom = (ly == 0), top = (ly == L - 1);
  if ((int64_t)blockIdx.x >= a.n_tiles) return;
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = a.timing ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long tstart = tlast;
#define MPROBE(slot_)                                              \
  if (a.timing) {                                                  \
    const unsigned long long now__ = __builtin_amdgcn_s_memtime(); \
    tacc[slot_] += now__ - tlast;                                  \
    tlast = now__;                                                 \
  }

  // ---- register-stationary split weights: B fragment (col 16j + arow of gate q, k = 32 kc + 8 ag + e) per (src, q, kc, piece)
  f32x4 w[2][4][2][NS];  // (128-bit containers of 8 bf16)
#pragma unroll
  for (int src = 0; src < 2; ++src)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const uint4 u = a.wsp[((((src * 4 + q) * 2 + kc) * NS + s) * 4 + j) * 64 + lane];
          w[src][q][kc][s] = __builtin_bit_cast(f32x4, u);
        }
  f32x4 bias4[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { const float bv = a.bias_sc[q * DH + j * 16 + arow]; bias4[q] = f32x4{bv, bv, bv, bv}; }
#pragma unroll
  for (int src = 0; src < 2; ++src)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int kc = 0; kc < 2; ++kc)
#pragma unroll
        for (int s = 0; s < NS; ++s) asm volatile("" : "+a"(w[src][q][kc][s]));

  float c[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) c[m][r] = 0.f;

  auto tile_k0 = [&](int64_t tl) -> int { return a.tile_k ? __builtin_amdgcn_readfirstlane(a.tile_k[tl]) : 0; };

  // ---- the step input of (tile, t): table gather (bottom) or the h of the layer below; each thread serves one 16-byte
  //      chunk column of rows (tid >> 4) + 16 k
  f32x4 gv[4];
  GatherSrc gsrc;
  if (bottom) gsrc = gather_src(a);
  auto in_load = [&](int64_t tile, int t, const int32_t* ids) {
    if (bottom) {
      gather_load<NT>(a, gsrc, tile, t, ids, gv);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = (threadIdx.x >> 4) + k * 16, ch = threadIdx.x & 15;
        gv[k] = *(const f32x4*)(a.Hin + (((int64_t)tile * T + t) * MT + row) * DH + ch * 4);
      }
    }
  };
  auto in_store = [&](E* dst) {
    const float xs = bottom ? mc_pow2(F::KA) : 1.0f;  // (fp16 pieces: the table rows are pre-scaled, W_i2g carries the inverse)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = (threadIdx.x >> 4) + k * 16, ch = threadIdx.x & 15;
      E* p = dst + row * LDB + ch * 4;
      f32x4 r = gv[k] * xs;
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) {
        Ex4 piece;
#pragma unroll
        for (int e = 0; e < 4; ++e) { piece[e] = (E)r[e]; r[e] -= (float)piece[e]; }
        *(Ex4*)(p + s2 * PLANE) = piece;
      }
    }
  };

  // the NEXT tile's ids: requested (registers) before a tile's first step, landed in LDS behind it -- the load latency sits
  // under a whole step of MFMAs instead of in front of it
  int32_t idreg[MT * MAXT_LDS / NT][3];
  auto ids_request = [&](int64_t tl) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < MT * MAXT_LDS / NT; ++it) {
      const int cc = threadIdx.x + it * NT;
      if (cc < MT * T) {
        const int row = cc / T, tt = cc - row * T;
        int64_t n = tl * MT + row;
        if (n >= a.N) n = a.N - 1;
        const int32_t* f = a.idx + (n * T + tt) * a.F;
        idreg[it][0] = f[a.F - a.nT - 2]; idreg[it][1] = f[a.F - 2]; idreg[it][2] = f[a.F - 1];
      }
    }
  };
  auto ids_land = [&](int32_t* ids) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < MT * MAXT_LDS / NT; ++it) {
      const int cc = threadIdx.x + it * NT;
      if (cc < MT * T) { ids[cc * 4 + 0] = idreg[it][0] - 1; ids[cc * 4 + 1] = idreg[it][1] - 1; ids[cc * 4 + 2] = idreg[it][2] - 1; }
    }
  };
  if (bottom) ids_stage<NT>(a.idx, a.N, T, a.F, a.nT, blockIdx.x, idbuf(0));
  if (a.tile_k) {
    const int n_cls = __builtin_amdgcn_readfirstlane(a.pmeta[0]) + 1;
    for (int cc = PFB + threadIdx.x; cc < n_cls * PFB; cc += NT) pft[cc] = a.pfb[(int64_t)(cc / PFB) * L * PFB + (int64_t)ly * PFB + cc % PFB];
  }
  lds_barrier();
  int k0 = tile_k0(blockIdx.x);
  in_load(blockIdx.x, k0, idbuf(0));
  in_store(inb(0));

  f32x4 accs[2][4];
  f32x4 sv[NPL];
  const int64_t frag_unit = (int64_t)NPL * 256;
  const int64_t frag_mt_stride = (int64_t)T * L * 4 * frag_unit;
  const int a_off = arow * LDB + ag * 8;               // this lane's A fragment inside a 16-row block (bf16 elements)
  const int o_off = (ag * 4) * LDB + j * 16 + arow;    // this lane's cell output inside a 16-row block

  auto save_unit = [&](int64_t p_tile, int p_t, int pm) __attribute__((always_inline)) {
    if (!SAVE) return;
    float* fb = a.save_frag + (p_tile * 4 + pm) * frag_mt_stride + ((int64_t)(p_t * L + ly) * 4 + j) * frag_unit + lane * 4;
#pragma unroll
    for (int k = 0; k < NPL; ++k) *(f32x4*)(fb + k * 256) = sv[k];
  };
  auto hout_ptr = [&](int64_t q_tile, int q_t, int pm) -> float* {
    return (top || (MC_EXP & 4)) ? nullptr : a.Hout + (((int64_t)q_tile * T + q_t) * MT + pm * 16 + ag * 4) * DH + j * 16 + arow;
  };

  // One unit = the 4-gate GEMM of a 16-row m-tile: [recurrent half over h_{t-1}] + [input half], K = 64 each = 2 chunks of 32,
  // NTERM MFMAs per (gate, chunk); the cell of the PREVIOUS unit is issued one step behind each of the first 64 MFMAs and
  // (two MFMAs later: the previous unit's last results have landed by then) executes beside them.  Accumulator chains: the gates alternate, so an accumulator is touched every 4th MFMA.
  auto unit = [&](auto rec_tag, auto cell_tag, const E* in_base, const E* h_base, f32x4 (&acc)[4], const f32x4 (&pacc)[4], float (&pc)[4],
                  E* phrow, float* phout) __attribute__((always_inline)) {
    constexpr bool REC = decltype(rec_tag)::value;
    constexpr bool CELL = decltype(cell_tag)::value;
    constexpr int PER_CHUNK = 4 * NTERM;                   // MFMAs per K chunk of 32
    constexpr int TOTAL = (REC ? 4 : 2) * PER_CHUNK;       // chunks: [h 0, h 1,] in 0, in 1
    McCell x;
    // A pieces of the running chunk / of the next one, ping-pong by chunk parity: never copied (a register copy in front of
    // an asm MFMA is a VALU write the hazard checker cannot see)
    f32x4 fr[2][NS];
    {
      const E* first = (REC ? h_base : in_base);
#pragma unroll
      for (int s = 0; s < NS; ++s) fr[0][s] = *(const f32x4*)(first + s * PLANE);
    }
    static_for<0, TOTAL>([&](auto ic) __attribute__((always_inline)) {
      constexpr int n = decltype(ic)::value;
      constexpr int q = n & 3;
      constexpr int term = (n >> 2) % NTERM;
      constexpr int chunk = n / PER_CHUNK;
      constexpr int src = REC ? (chunk >> 1) : 1;            // 0 = recurrent half (h_{t-1}), 1 = input half
      constexpr int kc = chunk & 1;
      constexpr int sa = mc_ta(M, term), sb = mc_tb(M, term);
      constexpr bool has_next = (n / PER_CHUNK) + 1 < TOTAL / PER_CHUNK;
      constexpr int cur = chunk & 1;   // (chunk counts from 0 in both forms of the unit)
      f32x4(&af)[NS] = fr[cur];
      if constexpr (!(MC_EXP & 2)) {
        if constexpr (M == 2) {
          if constexpr (n < 4) MC_MFMA_HC(acc[q], af[sa], w[src][q][kc][sb], bias4[q]);
          else MC_MFMA_H(acc[q], af[sa], w[src][q][kc][sb]);
        } else {
          if constexpr (n < 4) MC_MFMA_C(acc[q], af[sa], w[src][q][kc][sb], bias4[q]);  // first MFMA of a gate's chain: srcC = the scaled bias
          else MC_MFMA(acc[q], af[sa], w[src][q][kc][sb]);
        }
      }
      if constexpr (has_next && term == 0 && q == 3 && !(MC_EXP & 16)) {
        // the next chunk's A pieces: next 32 k of the same tile, or the first chunk of the input half
        const E* nb = (kc == 0) ? ((src == 0) ? h_base : in_base) + 32 : in_base;
#pragma unroll
        for (int s = 0; s < NS; ++s) fr[cur ^ 1][s] = *(const f32x4*)(nb + s * PLANE);
      }
      // The 64 cell steps of the previous unit are spread EVENLY over the MFMA slots 3 .. TOTAL-1 (three MFMAs
      // behind the previous unit's last results, whichever side of its neighbours the scheduler puts a step) (a step is 1-3 VALU ops,
      // at most one transcendental): the VALU work per slot stays below the MFMA cadence, so neither pipe waits for the other.
      constexpr int SLOTS = TOTAL - 3;
      if constexpr (CELL && n == 3) {
        // the first cell step reads the previous unit's accumulators: without this pin the scheduler may hoist that read up to
        // the previous sched_barrier, i.e. to right behind the previous unit's last MFMA (34 cycles behind the writer of gate 1:
        // less than the 8-pass latency).  Behind three MFMAs of this unit every result has landed.
        KPRN_PIN_V4(const_cast<f32x4(&)[4]>(pacc));
      }
      if constexpr (CELL && n >= 3) {
        constexpr int k = (SLOTS >= 64) ? ((n - 3) * 64 + SLOTS - 1) / SLOTS : (n - 3);   // candidate step for this slot
        constexpr bool here = (SLOTS >= 64) ? (k < 64 && 3 + (k * SLOTS) / 64 == n) : (k < 64);
        if constexpr (here && !(MC_EXP & 1)) {
          mc_cell_step<M, SAVE, (k >> 4), (k & 15)>(x, pacc, pc, phrow, phout, sv);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
    if constexpr (CELL && TOTAL - 3 < 64) {
      // (bf16 mode, or a tile's first step: fewer MFMA slots than cell steps) the rest of the cell, exposed
      static_for<0, 64>([&](auto ic) __attribute__((always_inline)) {
        constexpr int m = decltype(ic)::value;
        if constexpr (m >= TOTAL - 3) mc_cell_step<M, SAVE, (m >> 4), (m & 15)>(x, pacc, pc, phrow, phout, sv);
      });
    }
  };

  auto slot = [&](auto first_tag, const int64_t tile, const int t, const int par, const bool has_prev, const int64_t p_tile, const int p_t,
                  const int cls) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;
    float cinit = 0.f, rec0[4] = {0.f, 0.f, 0.f, 0.f};
    if (FIRST) {
      if (cls > 0) {  // (uniform) prefix class: c_prefix, and W_o2g h_prefix joins the first step's pre-activations
        cinit = pft[cls * PFB + 4 * DH + j * 16 + arow];
#pragma unroll
        for (int q = 0; q < 4; ++q) rec0[q] = pft[cls * PFB + q * DH + j * 16 + arow] * ((q == 1) ? MC_N2LOG2E : MC_NLOG2E) * mc_pow2(F::KW);
      }
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[m][r] = cinit;
    }
    static_for<0, 4>([&](auto mt_c) __attribute__((always_inline)) {
      constexpr int mt = decltype(mt_c)::value;
      constexpr int pm = (mt > 0) ? mt - 1 : 3;
      constexpr bool cross = (mt == 0);  // the pending cell belongs to the previous slot
      const int64_t q_tile = cross ? p_tile : tile;
      const int q_t = cross ? p_t : t;
      const int q_par = cross ? (par ^ 1) : par;
      f32x4(&acc)[4] = accs[mt & 1];
      f32x4(&pacc)[4] = accs[(mt & 1) ^ 1];
      E* phrow = hb(q_par) + pm * 16 * LDB + o_off;
      float* phout = hout_ptr(q_tile, q_t, pm);
      const E* in_base = inb(par) + mt * 16 * LDB + a_off;
      const E* h_base = hb(par ^ 1) + mt * 16 * LDB + a_off;
      if constexpr (mt == 0) {
        if (FIRST) {
          // tile switch: the previous tile's last cell first (its h_T row block completes the head's input), then the head
          if (has_prev) {
            MC_DRAIN();
            KPRN_PIN_V4(pacc);
            mc_cell_all<M, SAVE>(pacc, c[3], phrow, phout, sv);
            save_unit(q_tile, q_t, pm);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) c[3][r] = cinit;
          lds_barrier();  // (A) this step's input tile is complete; previous tile's h_T complete
          if (has_prev && top) mc_head_tile<M>(a, hb(q_par), p_tile, j, lane);
          unit(std::false_type{}, std::false_type{}, in_base, h_base, acc, pacc, c[pm], phrow, phout);
        } else {
          MPROBE(1)  // slot entry -> barrier A
          lds_barrier();  // (A) input tile of this step + rows 0..47 of h_{t-1} are complete
          MPROBE(2)  // waiting at barrier A
          unit(std::true_type{}, std::true_type{}, in_base, h_base, acc, pacc, c[pm], phrow, phout);
          save_unit(q_tile, q_t, pm);
          MPROBE(3)  // unit 0 (recurrent slots)
        }
        if (FIRST && cls > 0) {
          MC_DRAIN();
          KPRN_PIN_V4(acc);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][r] += rec0[q];
        }
        if (!FIRST && !(MC_EXP & 8)) { lds_barrier(); MPROBE(4) }  // (B) rows 48..63 of h_{t-1} (the cell that ran beside this unit) are complete in every wave
      } else {
        if (FIRST) {
          unit(std::false_type{}, std::true_type{}, in_base, h_base, acc, pacc, c[pm], phrow, phout);
          if (cls > 0) {
            MC_DRAIN();
            KPRN_PIN_V4(acc);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[q][r] += rec0[q];
          }
        } else {
          unit(std::true_type{}, std::true_type{}, in_base, h_base, acc, pacc, c[pm], phrow, phout);
        }
        save_unit(q_tile, q_t, pm);
      }
    });
  };

  int64_t tile = blockIdx.x;
  int t = k0;
  int tpar = 0;
  int64_t p_tile = tile;
  int p_t = t;
  int par = 0;
  for (int64_t s = 0;; ++s) {
    par = (int)(s & 1);
    int tn = t + 1;
    int64_t tile_n = tile;
    int tpar_n = tpar;
    int k0_n = k0;
    if (tn == T) {
      tile_n += gridDim.x;
      tpar_n ^= 1;
      k0_n = (tile_n < a.n_tiles) ? tile_k0(tile_n) : 0;
      tn = k0_n;
    }
    const bool have_next = tile_n < a.n_tiles;
    const bool ids_pending = bottom && t == k0 && tile + gridDim.x < a.n_tiles;
    if (ids_pending) ids_request(tile + gridDim.x);
    if (have_next) in_load(tile_n, tn, idbuf(tpar_n));
    MPROBE(0)  // slot control, id staging, input request
    if (t == k0) { slot(std::true_type{}, tile, t, par, s > 0, p_tile, p_t, k0); MPROBE(6) }
    else { slot(std::false_type{}, tile, t, par, true, p_tile, p_t, 0); MPROBE(5) }  // (5: units 1..3 of recurrent slots; 6: first slots, whole)
    MC_DRAIN();
    KPRN_PIN_V4(accs[0]);
    KPRN_PIN_V4(accs[1]);
    if (ids_pending) ids_land(idbuf(tpar ^ 1));  // (read from the next tile's first gather, >= 1 step and 1 barrier later)
    if (have_next) in_store(inb(par ^ 1));
    MPROBE(7)  // drain + landing the next input tile
    p_tile = tile; p_t = t;
    if (!have_next) break;
    t = tn; tile = tile_n; tpar = tpar_n; k0 = k0_n;
  }
  // drain: the cell of the very last unit, then the last tile's head
  {
    MC_DRAIN();
    KPRN_PIN_V4(accs[1]);
    mc_cell_all<M, SAVE>(accs[1], c[3], hb(par) + 3 * 16 * LDB + o_off, hout_ptr(p_tile, p_t, 3), sv);
    save_unit(p_tile, p_t, 3);
    lds_barrier();
    if (top) mc_head_tile<M>(a, hb(par), p_tile, j, lane);
  }
  if (a.timing && threadIdx.x == 0) {
    tacc[0] += 0;
    for (int k = 0; k < 8; ++k) a.timing[(int64_t)blockIdx.x * 8 + k] = tacc[k];
    (void)tstart;
  }
#undef MPROBE
}

// ---- split weights in register order (rebuilt when the parameters change) ----
struct McPrepArgs { const float* Wi; const float* Wo; const float* bi; uint4* wsp; float* bias_sc; int layer; };
template <int M>
__global__ void k_mc_prep(McPrepArgs a) {
  typedef McFmt<M> F;
  typedef typename F::E E;
  typedef E Ex8 __attribute__((ext_vector_type(8)));
  // one thread per (src, q, kc, wave, lane): 8 consecutive k of one gate column
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 256) a.bias_sc[i] = a.bi[i] * (((i >> 6) == 1) ? MC_N2LOG2E : MC_NLOG2E) * mc_pow2(F::KW);
  if (i >= 2 * 4 * 2 * 4 * 64) return;
  const int lane = i & 63, jw = (i >> 6) & 3, kc = (i >> 8) & 1, q = (i >> 9) & 3, src = i >> 11;
  const int arow = lane & 15, ag = lane >> 4;
  const float* W = (src == 0) ? a.Wo : a.Wi;  // src 0 = recurrent half (W_o2g), 1 = input half (W_i2g)
  // exp2 scaling of the gate, and (fp16 pieces) the power-of-two range scaling: the input half of the bottom layer sees
  // table rows scaled by 2^KA, so its weights carry 2^(KW-KA); everything else 2^KW
  const int ksc = (src == 1 && a.layer == 0) ? F::KW - F::KA : F::KW;
  const float sc = ((q == 1) ? MC_N2LOG2E : MC_NLOG2E) * mc_pow2(ksc);
  const float* row = W + (int64_t)(q * DH + jw * 16 + arow) * DH + kc * 32 + ag * 8;
  Ex8 p[F::NP];
  for (int e = 0; e < 8; ++e) {
    float r = row[e] * sc;
    for (int s = 0; s < F::NP; ++s) { const E piece = (E)r; p[s][e] = piece; r -= (float)piece; }
  }
  for (int s = 0; s < F::NP; ++s) a.wsp[((((src * 4 + q) * 2 + kc) * F::NP + s) * 4 + jw) * 64 + lane] = __builtin_bit_cast(uint4, p[s]);
}

// ---- host side ----
template <int M, bool SAVE>
static void launch_mc(kprn_handle* h, const McArgs& a, int grid) {
  const size_t lds_bytes = (size_t)4 * McFmt<M>::NP * MT * LDB * 2 + 2 * MT * MAXT_LDS * 4 * sizeof(int32_t) + (size_t)(KCAP + 1) * PFB * sizeof(float);
  static PerDeviceOnce attr_done;
  if (attr_done.need()) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_lstm_fwd_mc<M, SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  }
  hipLaunchKernelGGL((k_lstm_fwd_mc<M, SAVE>), dim3(grid), dim3(256), lds_bytes, h->stream, a);
  HIP_TRY(hipGetLastError());
}

// split weights / scaled biases of every layer for the current parameters (on the handle's current stream; cached)
void mc_prepare(kprn_handle* h) {
  const kprn_config& c = h->cfg;
  if (c.compute_dtype == 0) return;
  State* s = st(h);
  const int ns = (c.compute_dtype == 2) ? 3 : (c.compute_dtype == 3) ? 2 : 1;  // format M of McFmt
  if (!s->mc_wsp) {
    HIP_TRY(kprn_dev_malloc((void**)&s->mc_wsp, (size_t)2 * 2 * 4 * 2 * 3 * 4 * 64 * sizeof(uint4)));
    HIP_TRY(kprn_dev_malloc((void**)&s->mc_bias, (size_t)2 * 256 * sizeof(float)));
    s->mc_dirty = true;
  }
  const size_t wsp_layer = (size_t)2 * 4 * 2 * 3 * 4 * 64;  // uint4 per layer (room for 3 pieces)
  if (s->mc_dirty || s->mc_ns != ns) {
    join_score(h);  // (a scoring pass on the second stream may still read the old pieces)
    ProfScope ps(h, "mc_weight_split");
    for (int l = 0; l < c.L; ++l) {
      McPrepArgs pa;
      pa.Wi = h->dense + h->layer[l].Wi; pa.Wo = h->dense + h->layer[l].Wo; pa.bi = h->dense + h->layer[l].bi;
      pa.wsp = (uint4*)s->mc_wsp + l * wsp_layer; pa.bias_sc = s->mc_bias + l * 256; pa.layer = l;
      if (ns == 3) hipLaunchKernelGGL(k_mc_prep<3>, dim3(16), dim3(256), 0, h->stream, pa);
      else if (ns == 2) hipLaunchKernelGGL(k_mc_prep<2>, dim3(16), dim3(256), 0, h->stream, pa);
      else hipLaunchKernelGGL(k_mc_prep<1>, dim3(16), dim3(256), 0, h->stream, pa);
    }
    HIP_TRY(hipGetLastError());
    s->mc_dirty = false; s->mc_ns = ns;
  }
}

// forward of the whole stack on the matrix cores; compute_dtype 1 (bf16) or 2 (f32x6)
void forward_mc(kprn_handle* h, const kprn_batch* b, bool save) {
  const kprn_config& c = h->cfg;
  State* s = st(h);
  const int ns = (c.compute_dtype == 2) ? 3 : (c.compute_dtype == 3) ? 2 : 1;  // format M of McFmt
  const int64_t N = (int64_t)b->B * b->P;
  const int L = c.L;
  const int64_t n_tiles = (N + MT - 1) / MT;
  prefix_forward(h, b);
  mc_prepare(h);
  const size_t wsp_layer = (size_t)2 * 4 * 2 * 3 * 4 * 64;  // uint4 per layer (room for 3 pieces)
  // hand-over buffer between the layers: one per stream (a scoring pass on the second stream runs beside the training forward)
  const int hs = (h->score_stream && h->stream == h->score_stream) ? 1 : 0;
  if (L > 1 && (N > s->mc_capN[hs] || b->T > s->mc_capT[hs])) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (s->mc_hseq[hs]) HIP_TRY(hipFree(s->mc_hseq[hs]));
    s->mc_capN[hs] = std::max<int64_t>(N, s->mc_capN[hs]); s->mc_capT[hs] = std::max(b->T, s->mc_capT[hs]);
    HIP_TRY(kprn_dev_malloc((void**)&s->mc_hseq[hs], (size_t)((s->mc_capN[hs] + MT - 1) / MT + 1) * s->mc_capT[hs] * MT * DH * sizeof(float)));
  }
  if (save) {
    if (N > s->cap_N || b->T > s->cap_T) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (s->save_frag) HIP_TRY(hipFree(s->save_frag));
      const int64_t cn = std::max<int64_t>(N, s->cap_N);
      const int ct = std::max(b->T, s->cap_T);
      const int64_t mts = (cn + 15) / 16 + 4;
      HIP_TRY(kprn_dev_malloc((void**)&s->save_frag, (size_t)mts * ct * c.L * 4 * NPL * 256 * sizeof(float)));
      s->cap_N = cn; s->cap_T = ct;
    }
  }
  const int cus = (!save && h->reserve_cus > 0) ? std::max(1, s->num_cu - h->reserve_cus) : s->num_cu;
  const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)cus);
  for (int l = 0; l < L; ++l) {
    McArgs a;
    a.idx = b->idx_s ? b->idx_s : b->idx; a.N = N; a.T = b->T; a.F = b->F; a.nT = c.num_types;
    a.Wt = h->dense + h->off_Wt; a.We = h->We; a.Wr = h->dense + h->off_Wr;
    a.dt = c.dt; a.de = c.de; a.dr = c.dr;
    a.L = L; a.layer = l;
    a.wsp = (const uint4*)s->mc_wsp + l * wsp_layer; a.bias_sc = s->mc_bias + l * 256;
    a.Hin = (l > 0) ? s->mc_hseq[hs] : nullptr;
    a.Hout = (l < L - 1) ? s->mc_hseq[hs] : nullptr;   // (L <= 2: one hand-over buffer)
    a.Wout = h->dense + h->off_outW; a.bout = h->dense + h->off_outb; a.C = c.C; a.S = h->ws.S;
    a.perm = b->perm; a.tile_k = b->tile_k; a.pmeta = b->pmeta; a.pfb = s->pfb;
    a.save_frag = save ? s->save_frag : nullptr;
    a.n_tiles = n_tiles;
    static const bool want_timing = KPRN_DEV_ENV("KPRN_TIMING") != nullptr;
    if (want_timing && !s->timing) HIP_TRY(kprn_dev_malloc((void**)&s->timing, (size_t)s->num_cu * 8 * sizeof(unsigned long long)));
    a.timing = s->timing;
    ProfScope ps(h, save ? "lstm_mc_fwd_train" : "lstm_mc_fwd");
    if (ns == 3) { if (save) launch_mc<3, true>(h, a, grid); else launch_mc<3, false>(h, a, grid); }
    else if (ns == 2) { if (save) launch_mc<2, true>(h, a, grid); else launch_mc<2, false>(h, a, grid); }
    else { if (save) launch_mc<1, true>(h, a, grid); else launch_mc<1, false>(h, a, grid); }
    if (s->timing) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      std::vector<unsigned long long> tb((size_t)grid * 8);
      HIP_TRY(hipMemcpy(tb.data(), s->timing, tb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double sum[8] = {0};
      for (int g = 0; g < grid; ++g) for (int k = 0; k < 8; ++k) sum[k] += (double)tb[(size_t)g * 8 + k];
      fprintf(stderr, "[kprn timing] mc fwd layer %d save=%d grid=%d avg cycles/WG: control+request %.0f to-barrierA %.0f wait-A %.0f unit0 %.0f wait-B %.0f units1-3 %.0f first-slots %.0f drain+land %.0f\n",
              l, (int)save, grid, sum[0] / grid, sum[1] / grid, sum[2] / grid, sum[3] / grid, sum[4] / grid, sum[5] / grid, sum[6] / grid, sum[7] / grid);
    }
  }
}

}  // namespace fused
