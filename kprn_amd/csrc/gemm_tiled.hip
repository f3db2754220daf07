// LDS-tiled fp32-MFMA GEMM for the wide configurations (D = H = 192 "reading B", config.sh's H = 250, configs[3] d = 128 -> D = 384),
// with the recurrent cell in its epilogue.  gfx950 only.
//
// Stands for the TH/THC BLAS calls under nn.Linear / FastLSTM's i2g + o2g / nn.Recurrence's i2h + h2h
// (release/songPathRnn/model/OneModel.lua:231-236,268-275) and the nn graph around them (the ~25 element-wise modules of a
// FastLSTM step): one launch per (layer, step) computes
//     pre = [x_t | h_{t-1}] [W_i2g | W_o2g]^T + b      (K runs over the input columns, then over the hidden columns)
// on v_mfma_f32_16x16x4_f32 and applies the cell to the accumulators, so neither the pre-activations nor a hoisted x W_i2g^T
// tensor ever exist in HBM.  A workgroup owns 128 paths x 32 hidden units x all gates: each wave holds the gates of ITS 16 units
// (cell math is lane-local on the MFMA C layout, as in the D = H = 64 persistent kernels).
//
// Tile 128 x 128 x 32, 4 waves (2 x 2), each wave 64 x 64 = 4 x 4 MFMA tiles (64 accumulator VGPRs), 2 workgroups per CU.
// Operand tiles come in with 16-byte global loads along the operand's contiguous dimension and sit in LDS
//   * k-contiguous operand (row-major [m][k]):   [128][32 + 4]: one ds_read_b128 feeds the k-slots of 4 consecutive MFMAs
//     (k-slot ag of MFMA jj <-> k = 4 ag + jj inside a 16-k group -- the same order on both operands, so any order is valid);
//   * m-contiguous operand ([k][m], a transposed use): [32][128 + 4], ds_read_b32 per fragment;
// double-buffered, one barrier per 32-k chunk, the next chunk's global loads in flight under the MFMAs.  Launch order is
// XCD-aware: the n-tiles of one m-tile run back to back on ONE XCD (ids 8 apart), so the A tile they share is fetched into that
// XCD's L2 once.
#include <string.h>

#include <algorithm>

#include "kprn_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 128, TN = 128, TK = 32;
constexpr int LDK = TK + 4;   // row stride of a [rows][k] tile (floats): 16-byte aligned, spreads the b128 fragment reads
constexpr int LDM = TM + 4;   // row stride of a [k][rows] tile
constexpr int TILE_F = TM * LDK;  // floats per operand tile (the larger of the two layouts)

enum { EPI_STORE = 0, EPI_ACCUM = 1, EPI_LSTM = 2, EPI_RNN = 3 };

struct TArgs {
  // C[M,N] (+)= A(M,K) B(K,N).  Layout 0: the operand is row-major over its m (n) index with k contiguous, ld = row stride;
  // layout 1: row-major over k with m (n) contiguous.
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  // second K segment (recurrent steps): A2 (M, K2), B2 (K2, N), same layouts
  const float* A2; int64_t lda2; const float* B2; int64_t ldb2; int64_t K2;
  float* C; int64_t ldc;
  int64_t M; int N; int64_t K;
  const float* bias;
  int64_t kchunk; int use_atomic;
  int64_t mtiles; int ntiles; int n_begin;   // column tiles of this launch start at n_begin
  // cell epilogues
  int H;                                   // hidden units (EPI_LSTM: N = 4 H in gate-major columns; EPI_RNN: N = H)
  const float* cprev; float* cout; float* hout; int64_t ldh;   // [M][ldh]
  float* act;                               // EPI_LSTM: gate values [M][4H] (training saves; nullable)  EPI_RNN: pre-activations [M][H]
  const float* bias2; const float* mask; int relu;   // EPI_RNN: h2h bias, MaskZero flags [M], ReLU / Tanh
};

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }

// one operand tile [ROWS][32] (layout 0) or [32][ROWS] (layout 1), ROWS = 128 or 64: this thread's ROWS / 32 float4 pieces
template <int LAY, int ROWS>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int64_t ld, int64_t r0, int64_t rmax, int64_t k0, int64_t kend,
                                          f32x4 (&v)[ROWS / 32], const int* rowmap /*EPI_LSTM B rows: tile row -> matrix row, -1 = none*/) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int e = 0; e < ROWS / 32; ++e) {
    const int f = tid + 256 * e;
    f32x4 x = f32x4{0.f, 0.f, 0.f, 0.f};
    if (LAY == 0) {
      const int row = f >> 3, kq = (f & 7) * 4;
      int64_t gr = r0 + row;
      bool ok = gr < rmax;
      if (rowmap) { const int mr = rowmap[row]; ok = mr >= 0; gr = mr; }
      const int64_t gk = k0 + kq;
      if (ok && gk < kend) x = *(const f32x4*)(P + gr * ld + gk);
    } else {
      const int kr = f / (ROWS / 4), mq = (f % (ROWS / 4)) * 4;
      const int64_t gk = k0 + kr, gr = r0 + mq;
      if (gk < kend && gr < rmax) x = *(const f32x4*)(P + gk * ld + gr);
    }
    v[e] = x;
  }
}
template <int LAY, int ROWS>
__device__ __forceinline__ void tile_store(float* __restrict__ T, const f32x4 (&v)[ROWS / 32]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int e = 0; e < ROWS / 32; ++e) {
    const int f = tid + 256 * e;
    if (LAY == 0) *(f32x4*)(T + (f >> 3) * LDK + (f & 7) * 4) = v[e];
    else *(f32x4*)(T + (f / (ROWS / 4)) * (ROWS + 4) + (f % (ROWS / 4)) * 4) = v[e];
  }
}
// fragments of the NT MFMA tiles of this wave for one 16-k group: frag[t][jj] <-> row base + step t + arow, k = 16 kg + 4 ag + jj
template <int LAY, int ROWS, int NT>
__device__ __forceinline__ void frag_read(const float* __restrict__ T, int row_base, int row_step, int kg, int arow, int ag, f32x4 (&fr)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int row = row_base + t * row_step + arow;
    if (LAY == 0) fr[t] = *(const f32x4*)(T + row * LDK + kg * 16 + ag * 4);
    else {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) fr[t][jj] = T[(kg * 16 + ag * 4 + jj) * (ROWS + 4) + row];
    }
  }
}

// NTW: MFMA column tiles per wave: 4 (workgroup tile 128 x 128) or 2 (128 x 64: the last column block of N = 192, 64, ...)
template <int LA, int LB, int EPI, int NTW>
__global__ __launch_bounds__(256, 2) void k_gemm_tiled(TArgs a) {
  constexpr int TNn = 32 * NTW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  auto As = [&](int i) -> float* { return lds + i * (2 * TILE_F); };              // buffer i: A tile | B tile
  auto Bs = [&](int i) -> float* { return lds + i * (2 * TILE_F) + TILE_F; };
  __shared__ int rowmap[TN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int arow = lane & 15, ag = lane >> 4;
  // XCD-aware tile order: ids 8 apart share an XCD; they walk the n-tiles of one m-tile before moving on
  const int64_t id = blockIdx.x;
  const int xcd = (int)(id & 7);
  const int64_t j = id >> 3;
  const int nt_idx = (int)(j % a.ntiles);
  const int64_t mt_idx = (j / a.ntiles) * 8 + xcd;
  if (mt_idx >= a.mtiles) return;
  const int64_t m0 = mt_idx * TM;
  const int n0 = a.n_begin + nt_idx * TNn;   // EPI_LSTM: first hidden unit of the tile is nt_idx * 32
  constexpr bool CELL = (EPI == EPI_LSTM);
  if (CELL) {
    // tile row n = q * 32 + u  ->  matrix row q * H + (32 nt + u)
    if (tid < TN) {
      const int q = tid >> 5, u = nt_idx * 32 + (tid & 31);
      rowmap[tid] = (u < a.H) ? q * a.H + u : -1;
    }
    __syncthreads();
  }
  const int* rmap = CELL ? rowmap : nullptr;
  const int64_t k_beg = (int64_t)blockIdx.y * a.kchunk;
  const int64_t k_end1 = (k_beg + a.kchunk < a.K) ? k_beg + a.kchunk : a.K;   // (split-K applies to the first segment only)
  const int64_t nch1 = (k_end1 > k_beg) ? (k_end1 - k_beg + TK - 1) / TK : 0;
  const int64_t nch2 = (a.A2 != nullptr) ? (a.K2 + TK - 1) / TK : 0;
  const int64_t nch = nch1 + nch2;

  f32x4 acc[4][NTW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jn = 0; jn < NTW; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 ra[4], rb[NTW];
  auto load_chunk = [&](int64_t c) {
    if (c < nch1) {
      const int64_t k0 = k_beg + c * TK;
      tile_load<LA, TM>(a.A, a.lda, m0, a.M, k0, k_end1, ra, nullptr);
      tile_load<LB, TNn>(a.B, a.ldb, CELL ? 0 : n0, CELL ? (int64_t)4 * a.H : (int64_t)a.N, k0, k_end1, rb, rmap);
    } else {
      const int64_t k0 = (c - nch1) * TK;
      tile_load<LA, TM>(a.A2, a.lda2, m0, a.M, k0, a.K2, ra, nullptr);
      tile_load<LB, TNn>(a.B2, a.ldb2, CELL ? 0 : n0, CELL ? (int64_t)4 * a.H : (int64_t)a.N, k0, a.K2, rb, rmap);
    }
  };
  if (nch > 0) {
    load_chunk(0);
    tile_store<LA, TM>(As(0), ra);
    tile_store<LB, TNn>(Bs(0), rb);
  }
  __syncthreads();
  // B fragment rows: plain tiles: (16 NTW) wn + 16 nt + arow; cell tiles: gate nt, units 16 wn + arow  ->  32 nt + 16 wn + arow
  const int b_base = CELL ? wn * 16 : wn * (16 * NTW);
  const int b_step = CELL ? 32 : 16;
  for (int64_t c = 0; c < nch; ++c) {
    const int cur = (int)(c & 1);
    const bool more = c + 1 < nch;
    if (more) load_chunk(c + 1);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      f32x4 fa[4], fb[NTW];
      frag_read<LA, TM, 4>(As(cur), wm * 64, 16, kg, arow, ag, fa);
      frag_read<LB, TNn, NTW>(Bs(cur), b_base, b_step, kg, arow, ag, fb);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jn = 0; jn < NTW; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][jj], fb[jn][jj], acc[i][jn], 0, 0, 0);
    }
    if (more) {
      tile_store<LA, TM>(As(cur ^ 1), ra);   // (last read one chunk ago, before the previous barrier)
      tile_store<LB, TNn>(Bs(cur ^ 1), rb);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane & 15, row = 4 (lane >> 4) + reg
  if constexpr (EPI == EPI_STORE || EPI == EPI_ACCUM) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jn = 0; jn < NTW; ++jn) {
        const int col = n0 + wn * (16 * NTW) + jn * 16 + arow;
        if (col >= a.N) continue;
        const float bv = (EPI == EPI_STORE && a.bias) ? a.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = m0 + wm * 64 + i * 16 + ag * 4 + r;
          if (row >= a.M) continue;
          float* dst = a.C + row * a.ldc + col;
          if (EPI == EPI_STORE) *dst = acc[i][jn][r] + bv;
          else if (a.use_atomic) unsafeAtomicAdd(dst, acc[i][jn][r]);
          else *dst += acc[i][jn][r];
        }
      }
  } else if constexpr (EPI == EPI_LSTM && NTW == 4) {
    // nn.FastLSTM step (gate order i, g, f, o in the 4H rows; OneModel.lua:236): acc[i][q] = pre-activation of gate q, unit u
    const int u = nt_idx * 32 + wn * 16 + arow;
    if (u < a.H) {
      float bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = a.bias[q * a.H + u];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = m0 + wm * 64 + i * 16 + ag * 4 + r;
          if (row >= a.M) continue;
          const float ig = sigm(acc[i][0][r] + bq[0]);
          const float gg = tanhf(acc[i][1][r] + bq[1]);
          const float fg = sigm(acc[i][2][r] + bq[2]);
          const float og = sigm(acc[i][3][r] + bq[3]);
          const float cp = a.cprev ? a.cprev[row * a.ldh + u] : 0.f;
          const float cc = fg * cp + ig * gg;
          a.cout[row * a.ldh + u] = cc;
          a.hout[row * a.ldh + u] = og * tanhf(cc);
          if (a.act) {
            float* g = a.act + row * (int64_t)4 * a.H + u;
            g[0] = ig; g[a.H] = gg; g[2 * a.H] = fg; g[3 * a.H] = og;
          }
        }
    }
  } else if constexpr (EPI == EPI_RNN) {
    // nn.Recurrence(nn.MaskZero(act(i2h x_t + h2h h_{t-1}), 1)) (OneModel.lua:240-266): N = H plain columns
#pragma unroll
    for (int jn = 0; jn < NTW; ++jn) {
      const int col = n0 + wn * (16 * NTW) + jn * 16 + arow;
      if (col >= a.N) continue;
      const float bv = a.bias[col] + a.bias2[col];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = m0 + wm * 64 + i * 16 + ag * 4 + r;
          if (row >= a.M) continue;
          const float p = acc[i][jn][r] + bv;
          a.act[row * a.ldh + col] = p;
          const float v = a.relu ? fmaxf(p, 0.f) : tanhf(p);
          a.hout[row * a.ldh + col] = (a.mask[row] != 0.f) ? v : 0.f;
        }
    }
  }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <int LA, int LB, int EPI, int NTW = 4>
void launch(hipStream_t s, const TArgs& a, int split_k) {
  const size_t lds_bytes = (size_t)4 * TILE_F * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_gemm_tiled<LA, LB, EPI, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done = true;
  }
  const int64_t mgroups = (a.mtiles + 7) / 8;
  const dim3 grid((unsigned)(mgroups * a.ntiles * 8), (unsigned)split_k);
  hipLaunchKernelGGL((k_gemm_tiled<LA, LB, EPI, NTW>), grid, dim3(256), lds_bytes, s, a);
  HIP_TRY(hipGetLastError());
}
// the column range [n_begin, n_begin + ntiles * 32 NTW) of one problem
template <int EPI, int NTW>
void launch_layouts(hipStream_t s, const TArgs& a, int LA, int LB, int split_k) {
  if (LA == 0 && LB == 0) launch<0, 0, EPI, NTW>(s, a, split_k);
  else if (LA == 0 && LB == 1) launch<0, 1, EPI, NTW>(s, a, split_k);
  else launch<1, 1, EPI, NTW>(s, a, split_k);
}

}  // namespace

namespace gemm {

// the tiled kernel takes a call when both operands are in one of its two layouts with 16-byte granularity; returns false otherwise
bool run_tiled(hipStream_t s, const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBk, int64_t sBn, float* C, int64_t ldc, int64_t M, int N,
               int64_t K, bool accumulate, const float* bias, int split_k) {
  if (M < 256 || N < 64 || K < 16) return false;
  int LA, LB;
  int64_t lda, ldb;
  if (sAk == 1) { LA = 0; lda = sAm; if ((K & 3) || (lda & 3)) return false; }
  else if (sAm == 1) { LA = 1; lda = sAk; if ((M & 3) || (lda & 3)) return false; }
  else return false;
  if (sBk == 1) { LB = 0; ldb = sBn; if ((K & 3) || (ldb & 3)) return false; }
  else if (sBn == 1) { LB = 1; ldb = sBk; if ((ldb & 3) || ldb < (((int64_t)N + 3) & ~(int64_t)3)) return false; }
  else return false;
  if (!al16(A) || !al16(B)) return false;
  if (LA == 1 && LB == 0) return false;  // (no caller)
  TArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.bias = bias;
  a.mtiles = (M + TM - 1) / TM;
  // column blocks: 128-wide tiles, and a 64-wide tile for a remainder of at most 64 columns (N = 192: 128 + 64, no idle half tile)
  const int n_full = N / TN, rem = N - n_full * TN;
  const int nt128 = n_full + (rem > 64 ? 1 : 0);
  const int nt64 = (rem > 0 && rem <= 64) ? 1 : 0;
  if (split_k < 1) split_k = 1;
  if (split_k > 1) {
    // split-K only as far as the chip needs it: ~3 workgroups per CU in flight (every extra split is a tile of atomics)
    const int64_t tiles = a.mtiles * (nt128 + nt64);
    const int64_t want = (3 * 256 + tiles - 1) / tiles;
    if (split_k > want) split_k = (int)std::max<int64_t>(1, want);
  }
  int64_t kchunk = (K + split_k - 1) / split_k;
  kchunk = ((kchunk + TK - 1) / TK) * TK;
  split_k = (int)((K + kchunk - 1) / kchunk);
  a.kchunk = kchunk; a.use_atomic = split_k > 1 ? 1 : 0;
  KPRN_REQUIRE(!(split_k > 1 && !accumulate), KPRN_E_ARG, "gemm: split-K needs accumulate mode");
  if (nt128 > 0) {
    a.ntiles = nt128; a.n_begin = 0;
    if (accumulate) launch_layouts<EPI_ACCUM, 4>(s, a, LA, LB, split_k); else launch_layouts<EPI_STORE, 4>(s, a, LA, LB, split_k);
  }
  if (nt64 > 0) {
    a.ntiles = 1; a.n_begin = n_full * TN;
    if (accumulate) launch_layouts<EPI_ACCUM, 2>(s, a, LA, LB, split_k); else launch_layouts<EPI_STORE, 2>(s, a, LA, LB, split_k);
  }
  return true;
}

// shapes the fused step kernels take: 16-byte rows everywhere
bool step_supported(const float* X, int64_t ldx, int Din, const float* Hprev, int64_t ldh, int H, const float* Wi, const float* Wo, int64_t N) {
  return N >= 256 && (Din & 3) == 0 && (H & 3) == 0 && (ldx & 3) == 0 && (ldh & 3) == 0 && al16(X) && al16(Wi) && al16(Wo) && (!Hprev || al16(Hprev));
}

// one nn.FastLSTM step of one layer: gates = [x_t | h_{t-1}] [W_i2g | W_o2g]^T + b, cell in the epilogue.
// act (nullable): gate values [N][4H] for the backward; Hprev / Cprev null at t = 0.
void lstm_step(hipStream_t s, const float* X, int64_t ldx, int Din, const float* Wi, const float* bi, const float* Hprev, const float* Wo,
               const float* Cprev, float* Cout, float* Hout, int64_t ldh, float* act, int64_t N, int H) {
  TArgs a;
  memset(&a, 0, sizeof(a));
  a.A = X; a.lda = ldx; a.B = Wi; a.ldb = Din; a.K = Din;
  if (Hprev) { a.A2 = Hprev; a.lda2 = ldh; a.B2 = Wo; a.ldb2 = H; a.K2 = H; }
  a.M = N; a.N = 4 * H; a.H = H; a.bias = bi;
  a.cprev = Cprev; a.cout = Cout; a.hout = Hout; a.ldh = ldh; a.act = act;
  a.mtiles = (N + TM - 1) / TM; a.ntiles = (H + 31) / 32;
  a.kchunk = ((Din + TK - 1) / TK) * TK;
  launch<0, 0, EPI_LSTM>(s, a, 1);
}

// one nn.Recurrence step: pre = i2h x_t + b_i2h + h2h h_{t-1} + b_h2h, h = MaskZero(act(pre)); pre is kept for the backward
void rnn_step(hipStream_t s, const float* X, int64_t ldx, int Din, const float* Wi, const float* bi, const float* Hprev, const float* Wh,
              const float* bh, const float* mask, float* pre, float* Hout, int64_t ldh, int64_t N, int H, int relu) {
  TArgs a;
  memset(&a, 0, sizeof(a));
  a.A = X; a.lda = ldx; a.B = Wi; a.ldb = Din; a.K = Din;
  if (Hprev) { a.A2 = Hprev; a.lda2 = ldh; a.B2 = Wh; a.ldb2 = H; a.K2 = H; }
  a.M = N; a.N = H; a.H = H; a.bias = bi; a.bias2 = bh; a.mask = mask; a.relu = relu;
  a.hout = Hout; a.act = pre; a.ldh = ldh;
  a.mtiles = (N + TM - 1) / TM; a.ntiles = (H + TN - 1) / TN;
  a.kchunk = ((Din + TK - 1) / TK) * TK;
  launch<0, 0, EPI_RNN>(s, a, 1);
}

}  // namespace gemm
