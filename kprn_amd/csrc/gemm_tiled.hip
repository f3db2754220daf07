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
  int split_major; int nsplit;               // split-K launches: XCD <-> K range (see the kernel)
  // cell epilogues
  int H;                                   // hidden units (EPI_LSTM: N = 4 H in gate-major columns; EPI_RNN: N = H)
  const float* cprev; float* cout; float* hout; int64_t ldh;   // [M][ldh]
  float* act;                               // EPI_LSTM: gate values [M][4H] (training saves; nullable)  EPI_RNN: pre-activations [M][H]
  const float* bias2; const float* mask; int relu;   // EPI_RNN: h2h bias, MaskZero flags [M], ReLU / Tanh
};

// Gate functions at fp32 accuracy on v_exp_f32 / v_rcp_f32 (1 ulp each) instead of libm's ~40-instruction expf / tanhf: the cell
// epilogue of a 128 x 32-unit tile evaluates 80 of them per thread, which with libm cost a quarter of the tile's MFMA time.
// e^x = 2^t (1 + ln2 (x log2e - t)): the product's rounding residual (and log2e's low word) is folded back in, so the result
// stays within ~2 ulp for any x; 1 / (1 + e^-x) and tanh x = (1 - e^-2|x|) / (1 + e^-2|x|) then have ABSOLUTE error <= 1.5e-7
// (the same class as the D = H = 64 persistent kernels' gate math; tests/test_gpu_wide.py gates it against the f64 oracle).
__device__ __forceinline__ float exp_fast(float x) {
  const float t = x * 1.4426950408889634f;
  const float lo = __builtin_fmaf(x, 1.9259629911e-8f, __builtin_fmaf(x, 1.4426950408889634f, -t));
  const float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, lo * 0.6931471805599453f, e);
}
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + exp_fast(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
  const float t = exp_fast(-2.0f * __builtin_fabsf(x));
  return __builtin_copysignf((1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t), x);
}
// 16-byte vectors at any 4-byte aligned address: gfx950 runs in unaligned access mode (the compiler itself emits
// global_load/store_dwordx4 for an align-4 <4 x float>), so rows need no particular pitch -- config.sh's H = 250 takes the same
// path as H = 256 (8-byte vectors, the previous answer to odd pitches, cost 25 % on the rnn step: twice the load / LDS-store count)
typedef f32x4 f32x4u __attribute__((aligned(4)));
// four consecutive columns of one row; nv = how many of the four lie inside the matrix
__device__ __forceinline__ void store4(float* __restrict__ p, const f32x4 v, int nv) {
  if (nv >= 4) *(f32x4u*)p = v;
  else {
#pragma unroll
    for (int r = 0; r < 3; ++r) if (r < nv) p[r] = v[r];
  }
}
__device__ __forceinline__ f32x4 load4(const float* __restrict__ p, int nv) {
  f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
  if (nv >= 4) v = *(const f32x4u*)p;
  else {
#pragma unroll
    for (int r = 0; r < 3; ++r) if (r < nv) v[r] = p[r];
  }
  return v;
}

// One operand tile [ROWS][32] (layout 0) or [32][ROWS] (layout 1), ROWS = 128 or 64: this thread's ROWS / 32 pieces of 4 floats.
// Loads are branch-free and address-cheap: one running 64-bit base + 32-bit piece offsets (set up once per K segment), every piece
// fetched from a clamped, always valid address; what lies outside the problem is zeroed ELEMENT-wise when the piece is written
// to LDS (a vector may straddle the end of K or of the rows: every device buffer carries 64 bytes of slack for that) -- and
// AFTER the chunk's MFMAs, so the loads stay in flight under them (a select right behind the load made the compiler wait for
// all eight loads before the first MFMA; conditional loads made each a basic block of its own).
template <int LAY, int ROWS>
struct TileLoader {
  static constexpr int NP = ROWS / 32;
  // piece e of thread tid is float 4 (tid + 256 e) of the tile: in layout 0 all pieces of a thread share their k offset and sit
  // 32 rows apart; in layout 1 they share their m offset and sit 1024 / ROWS k-rows apart
  const float* base;
  const float* safe;
  int off[NP];
  unsigned rmask;     // layout 0: bit e = piece e's row lies inside the problem;  layout 1: valid elements (0..4) of this thread's column group
  int k0ofs;          // k offset of piece 0 inside the chunk
  int64_t step;
  static constexpr int KSTEP = (LAY == 0) ? 0 : 1024 / ROWS;   // k distance between consecutive pieces
  __device__ __forceinline__ void init(const float* __restrict__ P, int64_t ld, int64_t r0, int64_t rmax, int64_t k0, const int* rowmap) {
    const int tid = threadIdx.x;
    safe = P;
    step = (LAY == 0) ? (int64_t)TK : (int64_t)TK * ld;
    rmask = 0;
    if (LAY == 0) {
      const int kq = (tid & 7) * 4;
      k0ofs = kq;
      base = P + (rowmap ? 0 : r0 * ld) + k0 + kq;   // (offsets stay tile-relative: 32 bits hold 128 rows of any pitch)
#pragma unroll
      for (int e = 0; e < NP; ++e) {
        const int row = (tid >> 3) + 32 * e;
        bool ok = r0 + row < rmax;
        int64_t rel = row;
        if (rowmap) { const int mr = rowmap[row]; ok = mr >= 0; rel = mr; }   // (rows of a weight matrix)
        rmask |= ok ? (1u << e) : 0u;
        off[e] = (int)((ok ? rel : 0) * ld);
      }
    } else {
      const int kr = tid / (ROWS / 4), mq = (tid % (ROWS / 4)) * 4;
      const int64_t left = rmax - (r0 + mq);
      rmask = (unsigned)(left >= 4 ? 4 : (left > 0 ? left : 0));
      k0ofs = kr;
      base = P + (k0 + kr) * ld + (left > 0 ? r0 + mq : 0);
#pragma unroll
      for (int e = 0; e < NP; ++e) off[e] = (int)(e * KSTEP * ld);
    }
  }
  // pieces of the chunk at k0; bits 3e .. 3e + 2 of the result: how many leading elements of piece e lie inside the problem
  __device__ __forceinline__ unsigned load(int64_t k0, int64_t kend, f32x4 (&v)[NP]) {
    unsigned mask = 0;
    const int64_t kleft = kend - (k0 + k0ofs);
    const unsigned kv = (unsigned)(kleft >= 4 ? 4 : (kleft > 0 ? kleft : 0));   // layout 0: valid elements along k
#pragma unroll
    for (int e = 0; e < NP; ++e) {
      unsigned nv;
      if (LAY == 0) nv = ((rmask >> e) & 1u) ? kv : 0u;
      else nv = (kleft > e * KSTEP) ? rmask : 0u;
      v[e] = *(const f32x4u*)(nv ? base + off[e] : safe);
      mask |= nv << (3 * e);
    }
    base += step;
    return mask;
  }
};
template <int LAY, int ROWS>
__device__ __forceinline__ void tile_store(float* __restrict__ T, const f32x4 (&v)[ROWS / 32], unsigned mask) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int e = 0; e < ROWS / 32; ++e) {
    const int f = tid + 256 * e;
    f32x4 x = v[e];
    const int nv = (int)((mask >> (3 * e)) & 7u);
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = (q < nv) ? x[q] : 0.f;
    if (LAY == 0) *(f32x4*)(T + (f >> 3) * LDK + (f & 7) * 4) = x;
    else *(f32x4*)(T + (f / (ROWS / 4)) * (ROWS + 4) + (f % (ROWS / 4)) * 4) = x;
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
  constexpr bool TR = (EPI != EPI_ACCUM);   // result blocks transposed in the lanes (see the epilogue)
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
  int nt_idx;
  int64_t mt_idx, split_idx = blockIdx.y;
  if (a.split_major) {
    // split-K problems (dW = dA^T [x | h]: few output tiles, a very long K): every tile of ONE K range runs on ONE XCD, so the
    // rows of both operands in that range come from HBM once and from that XCD's L2 for all the tiles
    const int64_t tiles = a.mtiles * a.ntiles;
    split_idx = (j / tiles) * 8 + xcd;
    const int64_t tl = j % tiles;
    mt_idx = tl / a.ntiles;
    nt_idx = (int)(tl % a.ntiles);
    if (split_idx >= a.nsplit) return;
  } else {
    nt_idx = (int)(j % a.ntiles);
    mt_idx = (j / a.ntiles) * 8 + xcd;
    if (mt_idx >= a.mtiles) return;
  }
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
  const int64_t k_beg = split_idx * a.kchunk;
  const int64_t k_end1 = (k_beg + a.kchunk < a.K) ? k_beg + a.kchunk : a.K;   // (split-K applies to the first segment only)
  const int64_t nch1 = (k_end1 > k_beg) ? (k_end1 - k_beg + TK - 1) / TK : 0;
  const int64_t nch2 = (a.A2 != nullptr) ? (a.K2 + TK - 1) / TK : 0;
  const int64_t nch = nch1 + nch2;

  f32x4 acc[4][NTW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jn = 0; jn < NTW; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 ra[TM / 32], rb[TNn / 32];
  unsigned ma = 0, mb = 0;
  TileLoader<LA, TM> la;
  TileLoader<LB, TNn> lb;
  const int64_t brow0 = CELL ? 0 : n0, bmax = CELL ? (int64_t)4 * a.H : (int64_t)a.N;
  auto seg_init = [&](int seg) {   // K segment 0: (A, B) from k_beg; segment 1: (A2, B2) from 0
    if (seg == 0) { la.init(a.A, a.lda, m0, a.M, k_beg, nullptr); lb.init(a.B, a.ldb, brow0, bmax, k_beg, rmap); }
    else { la.init(a.A2, a.lda2, m0, a.M, 0, nullptr); lb.init(a.B2, a.ldb2, brow0, bmax, 0, rmap); }
  };
  auto load_chunk = [&](int64_t c) {
    if (c == nch1) seg_init(1);   // (uniform)
    const bool s0 = c < nch1;
    const int64_t k0 = s0 ? k_beg + c * TK : (c - nch1) * TK;
    const int64_t ke = s0 ? k_end1 : a.K2;
    ma = la.load(k0, ke, ra);
    mb = lb.load(k0, ke, rb);
  };
  if (nch > 0) {
    if (nch1 > 0) seg_init(0);
    load_chunk(0);
    tile_store<LA, TM>(As(0), ra, ma);
    tile_store<LB, TNn>(Bs(0), rb, mb);
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
          for (int jn = 0; jn < NTW; ++jn) acc[i][jn] = TR ? __builtin_amdgcn_mfma_f32_16x16x4f32(fb[jn][jj], fa[i][jj], acc[i][jn], 0, 0, 0)
                                          : __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][jj], fb[jn][jj], acc[i][jn], 0, 0, 0);
    }
    if (more) {
      tile_store<LA, TM>(As(cur ^ 1), ra, ma);   // (last read one chunk ago, before the previous barrier)
      tile_store<LB, TNn>(Bs(cur ^ 1), rb, mb);
    }
    __syncthreads();
  }

  // ---- epilogue.  Except for +=, the MFMAs took the B fragment as their first operand, so the 16 x 16 blocks sit TRANSPOSED in the lanes:
  // lane (arow, ag) holds row arow, columns 4 ag .. 4 ag + 3 -- four consecutive columns of one row per lane, stored 16 bytes
  // at a time (the plain C/D layout gives a lane four ROWS of one column: 4-byte stores, 64-byte segments)
  if constexpr (EPI == EPI_ACCUM) {
    // += (split-K partial sums land as atomics): plain C/D layout, lane = column -- a wave's atomic covers 64-byte row segments
    // (with the transposed blocks it would touch 16 rows x 16 bytes: measured 14 % slower on the dW GEMMs)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int jn = 0; jn < NTW; ++jn) {
        const int col = n0 + wn * (16 * NTW) + jn * 16 + arow;
        if (col >= a.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = m0 + wm * 64 + i * 16 + ag * 4 + r;
          if (row >= a.M) continue;
          float* dst = a.C + row * a.ldc + col;
          if (a.use_atomic) unsafeAtomicAdd(dst, acc[i][jn][r]);
          else *dst += acc[i][jn][r];
        }
      }
  } else if constexpr (EPI == EPI_STORE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t row = m0 + wm * 64 + i * 16 + arow;
      if (row >= a.M) continue;
#pragma unroll
      for (int jn = 0; jn < NTW; ++jn) {
        const int col = n0 + wn * (16 * NTW) + jn * 16 + ag * 4;
        const int nv = a.N - col;
        if (nv <= 0) continue;
        f32x4 v = acc[i][jn];
        if (a.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (r < nv) v[r] += a.bias[col + r];
        }
        store4(a.C + row * a.ldc + col, v, nv);
      }
    }
  } else if constexpr (EPI == EPI_LSTM && NTW == 4) {
    // nn.FastLSTM step (gate order i, g, f, o in the 4H rows; OneModel.lua:236): acc[i][q] = pre-activations of gate q, units u .. u + 3
    const int u = nt_idx * 32 + wn * 16 + ag * 4;
    const int nv = a.H - u;
    if (nv > 0) {
      f32x4 bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) bq[q][r] = (r < nv) ? a.bias[q * a.H + u + r] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = m0 + wm * 64 + i * 16 + arow;
        if (row >= a.M) continue;
        const f32x4 cp = a.cprev ? load4(a.cprev + row * a.ldh + u, nv) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 ig, gg, fg, og, cc, hh;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ig[r] = sigm(acc[i][0][r] + bq[0][r]);
          gg[r] = tanh_fast(acc[i][1][r] + bq[1][r]);
          fg[r] = sigm(acc[i][2][r] + bq[2][r]);
          og[r] = sigm(acc[i][3][r] + bq[3][r]);
          cc[r] = fg[r] * cp[r] + ig[r] * gg[r];
          hh[r] = og[r] * tanh_fast(cc[r]);
        }
        store4(a.cout + row * a.ldh + u, cc, nv);
        store4(a.hout + row * a.ldh + u, hh, nv);
        if (a.act) {
          float* g = a.act + row * (int64_t)4 * a.H + u;
          store4(g, ig, nv); store4(g + a.H, gg, nv); store4(g + 2 * a.H, fg, nv); store4(g + 3 * a.H, og, nv);
        }
      }
    }
  } else if constexpr (EPI == EPI_RNN) {
    // nn.Recurrence(nn.MaskZero(act(i2h x_t + h2h h_{t-1}), 1)) (OneModel.lua:240-266): N = H plain columns
#pragma unroll
    for (int jn = 0; jn < NTW; ++jn) {
      const int col = n0 + wn * (16 * NTW) + jn * 16 + ag * 4;
      const int nv = a.N - col;
      if (nv <= 0) continue;
      f32x4 bv;
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = (r < nv) ? a.bias[col + r] + a.bias2[col + r] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = m0 + wm * 64 + i * 16 + arow;
        if (row >= a.M) continue;
        const f32x4 pre = acc[i][jn] + bv;
        const bool live = a.mask[row] != 0.f;
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = a.relu ? fmaxf(pre[r], 0.f) : tanh_fast(pre[r]);
          v[r] = live ? x : 0.f;
        }
        store4(a.act + row * a.ldh + col, pre, nv);
        store4(a.hout + row * a.ldh + col, v, nv);
      }
    }
  }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <int LA, int LB, int EPI, int NTW>
void launch(hipStream_t s, const TArgs& a, int split_k) {
  const size_t lds_bytes = (size_t)4 * TILE_F * sizeof(float);
  static PerDeviceOnce attr_done;
  if (attr_done.need()) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_gemm_tiled<LA, LB, EPI, NTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  }
  const int64_t mgroups = (a.mtiles + 7) / 8;
  dim3 grid((unsigned)(mgroups * a.ntiles * 8), (unsigned)split_k);
  TArgs b = a;
  b.split_major = 0; b.nsplit = split_k;
  if (split_k > 1 && a.mtiles * a.ntiles >= 16) {   // (measured: helps 36-tile problems, 4.96 vs 6.42 ms; hurts 6-tile ones)
    b.split_major = 1;
    grid = dim3((unsigned)(((split_k + 7) / 8) * 8 * a.mtiles * a.ntiles), 1);
  }
  hipLaunchKernelGGL((k_gemm_tiled<LA, LB, EPI, NTW>), grid, dim3(256), lds_bytes, s, b);
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
  if (M < 256 || N < 64 || K < 16) return false;   // (tried from 192 rows, for the shipped rnn's dW products with M = H = 250: 0.55 -> 1.54 ms, two row tiles leave the split-K launch three quarters empty -- profiles/r05/bench_k_*)
  int LA, LB;
  int64_t lda, ldb;
  if (sAk == 1) { LA = 0; lda = sAm; }        // A contiguous along k, or along m; B along k, or along n
  else if (sAm == 1) { LA = 1; lda = sAk; }
  else return false;
  if (sBk == 1) { LB = 0; ldb = sBn; }
  else if (sBn == 1) { LB = 1; ldb = sBk; }
  else return false;
  if (LA == 1 && LB == 0) return false;  // (no caller)
  TArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.bias = bias;
  a.mtiles = (M + TM - 1) / TM;
  // column blocks: 128-wide tiles, and a 64-wide tile for a remainder of at most 64 columns (N = 192: 128 + 64, no idle half tile)
  int n_full = N / TN, rem = N - n_full * TN;
  int nt128 = n_full + (rem > 64 ? 1 : 0);
  int nt64 = (rem > 0 && rem <= 64) ? 1 : 0;
  // split-K launches share their operands through one XCD's L2 only inside ONE launch: there a 128 + 64 split of the columns
  // would read A twice from HBM; three 64-wide tiles in one launch do not (N = 192: dW of reading B)
  const bool all64 = false;  // (measured: three 64-wide tiles are slower than 128 + 64 for N = 192: 1.51 vs 1.35 ms)
  if (all64) { nt64 = (N + 63) / 64; nt128 = 0; n_full = 0; }
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
    a.ntiles = nt64; a.n_begin = n_full * TN;
    if (accumulate) launch_layouts<EPI_ACCUM, 2>(s, a, LA, LB, split_k); else launch_layouts<EPI_STORE, 2>(s, a, LA, LB, split_k);
  }
  return true;
}

// the fused step kernels take any layer shape from 256 paths up (below that the launch cannot fill the chip: unfused kernels)
bool step_supported(const float* X, int64_t ldx, int Din, const float* Hprev, int64_t ldh, int H, const float* Wi, const float* Wo, int64_t N) {
  (void)X; (void)ldx; (void)Din; (void)Hprev; (void)ldh; (void)H; (void)Wi; (void)Wo;
  return N >= 256;
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
  launch<0, 0, EPI_LSTM, 4>(s, a, 1);
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
  launch<0, 0, EPI_RNN, 4>(s, a, 1);
}

}  // namespace gemm
