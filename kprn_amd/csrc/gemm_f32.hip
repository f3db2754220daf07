// Generic GEMM on the gfx950 MFMA: fp32 (v_mfma_f32_16x16x4_f32, exact f32, 157 TF peak) or, compute_dtype = bf16,
// bf16 products with fp32 accumulation (v_mfma_f32_16x16x16_bf16; operands stay fp32 in HBM / LDS and are rounded to
// nearest-even bf16 as the fragments are formed -- BASELINE.json configs[3] "bf16 MFMA LSTM", first step: the
// arithmetic; bf16 storage and a fused bf16 kernel are later rounds).
//
// This is the shape-agnostic fallback of the engine: any M, N, K, any operand strides
// (so A*B^T, A*B and A^T*B are one kernel), boundary-checked, optional split-K.  The
// fused LSTM kernels (lstm_fused_*.hip) replace it on the shapes they cover; it stays the
// reference GPU path for odd sizes (e.g. the shipped config's H=250, D=200,
// release/songPathRnn/run_scripts/config.sh:20-23).
//
// Replaces the TH/THC BLAS calls under nn.Linear / FastLSTM's i2g,o2g
// (release/songPathRnn/model/OneModel.lua:236,275).
#include <stdlib.h>

#include "kprn_internal.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// fp32 -> bf16, round to nearest even (finite inputs)
__device__ __forceinline__ short to_bf16(float x) {
  unsigned int u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (short)(u >> 16);
}

// x where keep, +0 elsewhere -- as a bit mask the optimiser cannot see through.  The operand tiles' loads are issued unconditionally from clamped
// (always valid) addresses and masked with this: written as `in range ? load : 0` -- or as a select after the load, which hipcc turns back into
// the same thing -- every element becomes a branch with the load AND its wait inside: 16 dependent round trips per k-step.
__device__ __forceinline__ float masked(float x, bool keep) {
  unsigned m = keep ? 0xffffffffu : 0u;
  asm("" : "+v"(m));
  return __uint_as_float(__float_as_uint(x) & m);
}

constexpr int BM = 64, BN = 64, BK = 32, LDT = 80;  // LDT: k-major tile row stride (2-way = minimal bank sharing for 64 lanes)

template <bool A_KCONTIG, bool B_NCONTIG, bool BF16>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, int64_t sAm, int64_t sAk,
                                                   const float* __restrict__ B, int64_t sBk, int64_t sBn,
                                                   float* __restrict__ C, int64_t ldc, int64_t M, int N, int64_t K,
                                                   int accumulate, const float* __restrict__ bias, int64_t kchunk,
                                                   int use_atomic) {
  // double-buffered k-major tiles: the next tile's global loads are in flight while this tile's MFMAs run,
  // one barrier per k-step
  __shared__ float As[2][BK][LDT];
  __shared__ float Bs[2][BK][LDT];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m_base = (int64_t)blockIdx.x * BM;
  const int n_base = blockIdx.y * BN;
  const int64_t k_beg = (int64_t)blockIdx.z * kchunk;
  const int64_t k_end = (k_beg + kchunk < K) ? k_beg + kchunk : K;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int PER = BM * BK / 256;  // elements of each tile per thread
  float ra[PER], rb[PER];
  unsigned keep_a = 0u, keep_b = 0u;   // bit e: element e lies inside the problem (applied when the tile is written to LDS, behind the MFMAs)
  // element e of this thread: the fast thread index runs along the operand's contiguous dimension (coalesced loads)
  auto a_mk = [&](int e, int& m, int& k) {
    if (A_KCONTIG) { k = tid & (BK - 1); m = (tid / BK) + e * (256 / BK); }
    else           { m = tid & 63; k = (tid >> 6) + e * 4; }
  };
  auto b_nk = [&](int e, int& n, int& k) {
    if (B_NCONTIG) { n = tid & 63; k = (tid >> 6) + e * 4; }
    else           { k = tid & (BK - 1); n = (tid / BK) + e * (256 / BK); }
  };
  // zero-filled outside the problem (masked())
  auto load_tile = [&](int64_t k0) {
    keep_a = keep_b = 0u;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      int m, k;
      a_mk(e, m, k);
      const int64_t gm = m_base + m, gk = k0 + k;
      ra[e] = A[(gm < M ? gm : M - 1) * sAm + (gk < K ? gk : K - 1) * sAk];
      keep_a |= (gm < M && gk < k_end) ? (1u << e) : 0u;
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      int n, k;
      b_nk(e, n, k);
      const int gn = n_base + n;
      const int64_t gk = k0 + k;
      rb[e] = B[(gk < K ? gk : K - 1) * sBk + (int64_t)(gn < N ? gn : N - 1) * sBn];
      keep_b |= (gn < N && gk < k_end) ? (1u << e) : 0u;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      int m, k;
      a_mk(e, m, k);
      As[buf][k][m] = masked(ra[e], (keep_a >> e) & 1u);
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      int n, k;
      b_nk(e, n, k);
      Bs[buf][k][n] = masked(rb[e], (keep_b >> e) & 1u);
    }
  };

  load_tile(k_beg);
  store_tile(0);
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = k_beg; k0 < k_end; k0 += BK) {
    const bool more = k0 + BK < k_end;
    if (more) load_tile(k0 + BK);
    if (BF16) {
      // 16x16x16 bf16 MFMA: lane (col = lane&15, kg = lane>>4) supplies k = 4kg..4kg+3 of each 16-k slab
#pragma unroll
      for (int ks = 0; ks < BK; ks += 16) {
        s16x4 a[2], b[2];
        const int kb = ks + (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q) a[i][q] = to_bf16(As[cur][kb + q][wm * 32 + i * 16 + (lane & 15)]);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) b[j][q] = to_bf16(Bs[cur][kb + q][wn * 32 + j * 16 + (lane & 15)]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BK; kk += 4) {
        float a[2], b[2];
        const int kr = kk + (lane >> 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = As[cur][kr][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = Bs[cur][kr][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) store_tile(cur ^ 1);  // the other buffer was last read one step ago, before the previous barrier
    __syncthreads();
    cur ^= 1;
  }
  // ---- epilogue: C/D layout col = lane&15, row = (lane>>4)*4 + reg
  float bj[2] = {0.f, 0.f};   // this lane's bias values, fetched once and together (inside the element loop each was a load with its own wait)
  if (bias && !accumulate && !use_atomic) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n_base + wn * 32 + j * 16 + (lane & 15);
      bj[j] = bias[col < N ? col : N - 1];
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float oldv[2][4];   // accumulate mode: the 8 old values of this row block, requested together (clamped addresses) before the first is used
    if (accumulate && !use_atomic) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = n_base + wn * 32 + j * 16 + (lane & 15);
          const int64_t row = m_base + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
          oldv[j][r] = C[(row < M ? row : M - 1) * ldc + (col < N ? col : N - 1)];
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n_base + wn * 32 + j * 16 + (lane & 15);
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m_base + wm * 32 + i * 16 + (lane >> 4) * 4 + r;
        if (row >= M) continue;
        float v = acc[i][j][r];
        float* dst = C + row * ldc + col;
        if (use_atomic) {
          unsafeAtomicAdd(dst, v);
        } else if (accumulate) {
          *dst = oldv[j][r] + v;
        } else {
          *dst = bias ? v + bj[j] : v;
        }
      }
    }
  }
}


// ---- 128 x 128 x 16 tile: each wave owns 64 x 64 (4 x 4 MFMA tiles), 8 LDS fragment reads per 16 MFMAs and a quarter
// of the small kernel's global loads per MFMA.  Used when both M and N are at least ~100 (the i2g / dW / dx GEMMs of
// the wider configurations: reading B, config.sh's H = 250, configs[3]).
constexpr int GM = 128, GN = 128, GK = 16, GLD = 136;

template <bool A_KCONTIG, bool B_NCONTIG, bool BF16>
__global__ __launch_bounds__(256) void gemm_kernel_big(const float* __restrict__ A, int64_t sAm, int64_t sAk,
                                                       const float* __restrict__ B, int64_t sBk, int64_t sBn,
                                                       float* __restrict__ C, int64_t ldc, int64_t M, int N, int64_t K,
                                                       int accumulate, const float* __restrict__ bias, int64_t kchunk,
                                                       int use_atomic) {
  __shared__ float As[2][GK][GLD];
  __shared__ float Bs[2][GK][GLD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m_base = (int64_t)blockIdx.x * GM;
  const int n_base = blockIdx.y * GN;
  const int64_t k_beg = (int64_t)blockIdx.z * kchunk;
  const int64_t k_end = (k_beg + kchunk < K) ? k_beg + kchunk : K;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int PER = GM * GK / 256;  // 8
  float ra[PER], rb[PER];
  unsigned keep_a = 0u, keep_b = 0u;   // bit e: element e lies inside the problem (applied when the tile is written to LDS, behind the MFMAs)
  auto a_mk = [&](int e, int& m, int& k) {
    if (A_KCONTIG) { k = tid & (GK - 1); m = (tid / GK) + e * (256 / GK); }
    else           { m = tid & 127; k = (tid >> 7) + e * 2; }
  };
  auto b_nk = [&](int e, int& n, int& k) {
    if (B_NCONTIG) { n = tid & 127; k = (tid >> 7) + e * 2; }
    else           { k = tid & (GK - 1); n = (tid / GK) + e * (256 / GK); }
  };
  auto load_tile = [&](int64_t k0) {
    keep_a = keep_b = 0u;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      int m, k;
      a_mk(e, m, k);
      const int64_t gm = m_base + m, gk = k0 + k;
      ra[e] = A[(gm < M ? gm : M - 1) * sAm + (gk < K ? gk : K - 1) * sAk];
      keep_a |= (gm < M && gk < k_end) ? (1u << e) : 0u;
    }
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      int n, k;
      b_nk(e, n, k);
      const int gn = n_base + n;
      const int64_t gk = k0 + k;
      rb[e] = B[(gk < K ? gk : K - 1) * sBk + (int64_t)(gn < N ? gn : N - 1) * sBn];
      keep_b |= (gn < N && gk < k_end) ? (1u << e) : 0u;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int e = 0; e < PER; ++e) { int m, k; a_mk(e, m, k); As[buf][k][m] = masked(ra[e], (keep_a >> e) & 1u); }
#pragma unroll
    for (int e = 0; e < PER; ++e) { int n, k; b_nk(e, n, k); Bs[buf][k][n] = masked(rb[e], (keep_b >> e) & 1u); }
  };

  load_tile(k_beg);
  store_tile(0);
  __syncthreads();
  int cur = 0;
  for (int64_t k0 = k_beg; k0 < k_end; k0 += GK) {
    const bool more = k0 + GK < k_end;
    if (more) load_tile(k0 + GK);
    if (BF16) {
      s16x4 a[4], b[4];
      const int kb = (lane >> 4) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) a[i][q] = to_bf16(As[cur][kb + q][wm * 64 + i * 16 + (lane & 15)]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) b[j][q] = to_bf16(Bs[cur][kb + q][wn * 64 + j * 16 + (lane & 15)]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[i], b[j], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < GK; kk += 4) {
        float a[4], b[4];
        const int kr = kk + (lane >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = As[cur][kr][wm * 64 + i * 16 + (lane & 15)];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Bs[cur][kr][wn * 64 + j * 16 + (lane & 15)];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  float bj[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias && !accumulate && !use_atomic) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n_base + wn * 64 + j * 16 + (lane & 15);
      bj[j] = bias[col < N ? col : N - 1];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float oldv[4][4];
    if (accumulate && !use_atomic) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = n_base + wn * 64 + j * 16 + (lane & 15);
          const int64_t row = m_base + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
          oldv[j][r] = C[(row < M ? row : M - 1) * ldc + (col < N ? col : N - 1)];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n_base + wn * 64 + j * 16 + (lane & 15);
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m_base + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
        if (row >= M) continue;
        float v = acc[i][j][r];
        float* dst = C + row * ldc + col;
        if (use_atomic) unsafeAtomicAdd(dst, v);
        else if (accumulate) *dst = oldv[j][r] + v;
        else *dst = bias ? v + bj[j] : v;
      }
    }
  }
}

}  // namespace

namespace gemm {

void run(hipStream_t s, const float* A, int64_t sAm, int64_t sAk, const float* B, int64_t sBk, int64_t sBn, float* C,
         int64_t ldc, int64_t M, int N, int64_t K, bool accumulate, const float* bias, int split_k, bool bf16, bool untiled) {
  if (M <= 0 || N <= 0) return;
  if (split_k < 1) split_k = 1;
  static const bool no_tiled = getenv("KPRN_NO_TILED_GEMM") != nullptr;  // (measurement: the round-1 kernels)
  if (!bf16 && !no_tiled && !untiled && run_tiled(s, A, sAm, sAk, B, sBk, sBn, C, ldc, M, N, K, accumulate, bias, split_k)) return;
  int64_t kchunk = (K + split_k - 1) / split_k;
  kchunk = ((kchunk + BK - 1) / BK) * BK;
  if (kchunk <= 0) kchunk = BK;
  split_k = (int)((K + kchunk - 1) / kchunk);
  if (split_k < 1) split_k = 1;
  KPRN_REQUIRE(!(split_k > 1 && !accumulate), KPRN_E_ARG, "gemm: split-K needs accumulate mode");
  const int use_atomic = split_k > 1 ? 1 : 0;
  const bool akc = (sAk == 1), bnc = (sBn == 1);
  const bool big = (M >= 100 && N >= 100);
  if (big) {
    kchunk = ((kchunk + GK - 1) / GK) * GK;
    dim3 grid((unsigned)((M + GM - 1) / GM), (unsigned)((N + GN - 1) / GN), (unsigned)split_k);
#define LAUNCHB(AK, BNC)                                                                                           \
  do {                                                                                                             \
    if (bf16)                                                                                                      \
      hipLaunchKernelGGL((gemm_kernel_big<AK, BNC, true>), grid, dim3(256), 0, s, A, sAm, sAk, B, sBk, sBn, C, ldc, M, N, K, \
                         accumulate ? 1 : 0, bias, kchunk, use_atomic);                                            \
    else                                                                                                           \
      hipLaunchKernelGGL((gemm_kernel_big<AK, BNC, false>), grid, dim3(256), 0, s, A, sAm, sAk, B, sBk, sBn, C, ldc, M, N, K, \
                         accumulate ? 1 : 0, bias, kchunk, use_atomic);                                            \
  } while (0)
    if (akc && bnc) LAUNCHB(true, true);
    else if (akc && !bnc) LAUNCHB(true, false);
    else if (!akc && bnc) LAUNCHB(false, true);
    else LAUNCHB(false, false);
#undef LAUNCHB
    HIP_TRY(hipGetLastError());
    return;
  }
  dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN), (unsigned)split_k);
#define LAUNCH(AK, BNC)                                                                                            \
  do {                                                                                                             \
    if (bf16)                                                                                                      \
      hipLaunchKernelGGL((gemm_kernel<AK, BNC, true>), grid, dim3(256), 0, s, A, sAm, sAk, B, sBk, sBn, C, ldc, M, N, K, \
                         accumulate ? 1 : 0, bias, kchunk, use_atomic);                                            \
    else                                                                                                           \
      hipLaunchKernelGGL((gemm_kernel<AK, BNC, false>), grid, dim3(256), 0, s, A, sAm, sAk, B, sBk, sBn, C, ldc, M, N, K, \
                         accumulate ? 1 : 0, bias, kchunk, use_atomic);                                            \
  } while (0)
  if (akc && bnc) LAUNCH(true, true);
  else if (akc && !bnc) LAUNCH(true, false);
  else if (!akc && bnc) LAUNCH(false, true);
  else LAUNCH(false, false);
#undef LAUNCH
  HIP_TRY(hipGetLastError());
}

}  // namespace gemm
