// bf16 pipeline of the FastLSTM path (BASELINE.json configs[3]: 20 M entities, d = 128 -> D = H = 384, "bf16 MFMA LSTM"): bf16 STORAGE
// of everything the matrix cores read -- a bf16 shadow of the embedding tables and of the weights, bf16 step inputs, hidden states,
// gate values and pre-activation gradients -- products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation, fp32 cell state, fp32 master
// parameters and fp32 lazy-exact Adam (kprn_api.hip), which refreshes the shadow rows it touches.  gfx950 only.
//
// Stands for the same reference graph as gemm_tiled.hip (nn.LookupTable x 3 -> nn.Sequencer(nn.FastLSTM) x L -> nn.Linear:
// release/songPathRnn/net/FeatureEmbedding.lua:112-121, model/OneModel.lua:236,268-275) at reduced precision; tolerance-gated against the
// float64 oracle (tests/test_gpu_parity.py, tests/test_gpu_wide.py).
//
// One tile kernel: C[M,N] = A[M,K] B[N,K]^T, both operands k-contiguous bf16 (16-byte = 8-element global loads, LDS tiles [128][64 + 8],
// one ds_read_b128 per MFMA operand), 128 x 128 x 64 tiles, 4 waves, 2 workgroups per CU, XCD-aware order.  Everything the backward needs
// transposed (dW = dA^T [x | h]: the contraction runs over the path rows) is brought into that one layout by a bf16 transpose kernel --
// memory-bound passes of a few hundred microseconds, instead of a second GEMM formulation with strided LDS reads.
#include <string.h>

#include <algorithm>

#include "kprn_internal.h"

namespace bf16p {

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TK = 64, LDB = TK + 8;   // LDS row pitch in elements (144 bytes)

// gate functions on v_exp_f32 / v_rcp_f32 (1 ulp each): at bf16 precision the cell's transcendental functions, not the MFMAs, were
// most of this kernel's time with libm's expf / tanhf (~40 instructions each against ~17 cycles per 16 KFLOP MFMA)
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f; }
__device__ __forceinline__ bf16 tobf(float x) { return (bf16)x; }   // round to nearest even

enum { EPI_STORE = 0, EPI_ACCUM = 1, EPI_LSTM = 2 };

struct GArgs {
  const bf16* A; int64_t lda; const bf16* B; int64_t ldb; int64_t K;
  const bf16* A2; int64_t lda2; const bf16* B2; int64_t ldb2; int64_t K2;   // second K segment (recurrent half)
  float* C; int64_t ldc; int64_t M; int N; const float* bias;
  int64_t kchunk; int use_atomic; int nsplit;
  int64_t mtiles; int ntiles;
  // EPI_LSTM: N = 4H gate-major rows of B; a column tile = 32 hidden units x 4 gates
  int H; const float* cprev; float* cout; bf16* hout; int64_t ldh; bf16* act; float* hout_f32;
};

// Tile geometry. BIG = 0: 128 x 128 block, 4 waves (2 x 2), 64 x 64 per wave, two blocks per CU.
// BIG = 1: 256 x 256 block, 8 waves (2 x 4), 128 x 64 per wave, one block per CU (half the L2 -> LDS bytes and LDS reads per MFMA).
// Measured on the configs[3] step GEMM (65 536 x 1 536 x 768, 0.27 ms either way, profiles/r02/README.md): the launch is not bound
// by the matrix cores (compiling the MFMAs out saves 6 %) but by the un-overlapped sum of the operand stream, the cell epilogue
// (10 transcendentals per cell, c_prev read) and the h / c / gate stores -- 0.5 GB of HBM traffic per launch against 62 us of MFMA
// work. The 256-wide geometry is therefore only taken for plain GEMMs with >= 512 such tiles, where it trims the operand re-reads.
template <int BIG> struct Geo {
  static constexpr int TMx = BIG ? 256 : 128, TNx = TMx, NT = BIG ? 512 : 256;
  static constexpr int RSTEP = NT / 8;            // tile rows covered by one 16-byte load of every thread
  static constexpr int WN = BIG ? 4 : 2;          // waves along the columns
  static constexpr int FI = BIG ? 8 : 4;          // 16-row fragments per wave
  static constexpr int HC = TNx / 4;              // EPI_LSTM: hidden units per block (x 4 gates = TNx columns)
  static constexpr int TILE = TMx * LDB;          // elements of one operand tile in LDS
};

// pieces of one [TMx][64] bf16 tile: running pointers, clamped loads, zeroing at the LDS write (as gemm_tiled.hip's TileLoader)
template <int RSTEP>
struct Loader {
  const bf16* p[4]; const bf16* safe; bool rok[4]; int kq;
  __device__ __forceinline__ void init(const bf16* __restrict__ P, int64_t ld, int64_t r0, int64_t rmax, int64_t k0, const int* rowmap) {
    const int tid = threadIdx.x;
    safe = P; kq = (tid & 7) * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = (tid >> 3) + RSTEP * e;
      int64_t gr = r0 + row;
      bool ok = gr < rmax;
      if (rowmap) { const int mr = rowmap[row]; ok = mr >= 0; gr = mr; }
      rok[e] = ok;
      p[e] = P + (ok ? gr : 0) * ld + k0 + kq;
    }
  }
  __device__ __forceinline__ unsigned load(int64_t k0, int64_t kend, bf16x8 (&v)[4]) {
    unsigned mask = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = rok[e] && (k0 + kq < kend);
      v[e] = *(const bf16x8*)(ok ? p[e] : safe);
      mask |= ok ? (1u << e) : 0u;
      p[e] += TK;
    }
    return mask;
  }
};
template <int RSTEP>
__device__ __forceinline__ void tile_store(bf16* __restrict__ T, const bf16x8 (&v)[4], unsigned mask) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bf16x8 x = v[e];
    if (!((mask >> e) & 1u)) {
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = (bf16)0.f;
    }
    *(bf16x8*)(T + ((tid >> 3) + RSTEP * e) * LDB + (tid & 7) * 8) = x;
  }
}

// The MFMA takes the B fragment as its first operand and the A fragment as its second: the 16 x 16 result block then comes out
// transposed in the lanes -- lane (arow, ag) holds C[row = arow][col = 4 ag .. 4 ag + 3] -- so every lane owns FOUR CONSECUTIVE
// COLUMNS of one row and the epilogues store 8 / 16 bytes per lane (h, the gate activations, c) instead of 2 / 4.
template <int EPI, int BIG>
__global__ __launch_bounds__(Geo<BIG>::NT, BIG ? 1 : 2) void k_gemm16(GArgs a) {
  using G = Geo<BIG>;
  constexpr int TMx = G::TMx, TNx = G::TNx, FI = G::FI, HC = G::HC, TILE = G::TILE, RS = G::RSTEP;
  extern __shared__ __attribute__((aligned(16))) bf16 lds16[];
  auto As = [&](int i) -> bf16* { return lds16 + i * (2 * TILE); };
  auto Bs = [&](int i) -> bf16* { return lds16 + i * (2 * TILE) + TILE; };
  __shared__ int rowmap[TNx];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / G::WN, wn = wave % G::WN, arow = lane & 15, ag = lane >> 4;
  const int64_t id = blockIdx.x;
  const int xcd = (int)(id & 7);
  const int64_t j = id >> 3;
  int nt_idx; int64_t mt_idx, split_idx = 0;
  if (a.nsplit > 1) {   // all tiles of one K range on one XCD (gemm_tiled.hip)
    const int64_t tiles = a.mtiles * a.ntiles;
    split_idx = (j / tiles) * 8 + xcd;
    const int64_t tl = j % tiles;
    mt_idx = tl / a.ntiles; nt_idx = (int)(tl % a.ntiles);
    if (split_idx >= a.nsplit) return;
  } else {
    nt_idx = (int)(j % a.ntiles);
    mt_idx = (j / a.ntiles) * 8 + xcd;
    if (mt_idx >= a.mtiles) return;
  }
  const int64_t m0 = mt_idx * TMx;
  const int n0 = nt_idx * TNx;
  constexpr bool CELL = (EPI == EPI_LSTM);
  if (CELL) {
    if (tid < TNx) {
      const int q = tid / HC, u = nt_idx * HC + (tid % HC);
      rowmap[tid] = (u < a.H) ? q * a.H + u : -1;
    }
    __syncthreads();
  }
  const int* rmap = CELL ? rowmap : nullptr;
  const int64_t k_beg = split_idx * a.kchunk;
  const int64_t k_end1 = (k_beg + a.kchunk < a.K) ? k_beg + a.kchunk : a.K;
  const int64_t nch1 = (k_end1 > k_beg) ? (k_end1 - k_beg + TK - 1) / TK : 0;
  const int64_t nch2 = (a.A2 != nullptr) ? (a.K2 + TK - 1) / TK : 0;
  const int64_t nch = nch1 + nch2;
  f32x4 acc[FI][4];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ra[4], rb[4];
  unsigned ma = 0, mb = 0;
  Loader<RS> la, lb;
  const int64_t brow0 = CELL ? 0 : n0, bmax = CELL ? (int64_t)4 * a.H : (int64_t)a.N;
  auto seg_init = [&](int seg) {
    if (seg == 0) { la.init(a.A, a.lda, m0, a.M, k_beg, nullptr); lb.init(a.B, a.ldb, brow0, bmax, k_beg, rmap); }
    else { la.init(a.A2, a.lda2, m0, a.M, 0, nullptr); lb.init(a.B2, a.ldb2, brow0, bmax, 0, rmap); }
  };
  auto load_chunk = [&](int64_t c) {   // (chunks are loaded in order)
    if (c == nch1) seg_init(1);
    const bool s0 = c < nch1;
    const int64_t k0 = s0 ? k_beg + c * TK : (c - nch1) * TK;
    const int64_t ke = s0 ? k_end1 : a.K2;
    ma = la.load(k0, ke, ra);
    mb = lb.load(k0, ke, rb);
  };
  if (nch > 0) {
    if (nch1 > 0) seg_init(0);
    load_chunk(0);
    tile_store<RS>(As(0), ra, ma);
    tile_store<RS>(Bs(0), rb, mb);
  }
  __syncthreads();
  const int b_base = CELL ? wn * 16 : wn * 64;
  const int b_step = CELL ? HC : 16;
  for (int64_t c = 0; c < nch; ++c) {
    const int cur = (int)(c & 1);
    const bool more = c + 1 < nch;
    if (more) load_chunk(c + 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {   // two 32-k MFMA blocks per chunk; lane (row, ag) supplies k = 32 kk + 8 ag .. + 7
      bf16x8 fb[4];
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) fb[jn] = *(const bf16x8*)(Bs(cur) + (b_base + jn * b_step + arow) * LDB + kk * 32 + ag * 8);
#pragma unroll
      for (int ih = 0; ih < FI; ih += 4) {
        bf16x8 fa[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *(const bf16x8*)(As(cur) + (wm * (FI * 16) + (ih + i) * 16 + arow) * LDB + kk * 32 + ag * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jn = 0; jn < 4; ++jn)
            acc[ih + i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[jn], fa[i], acc[ih + i][jn], 0, 0, 0);
      }
    }
    if (more) {
      tile_store<RS>(As(cur ^ 1), ra, ma);
      tile_store<RS>(Bs(cur ^ 1), rb, mb);
    }
    __syncthreads();
  }
  if constexpr (EPI == EPI_STORE || EPI == EPI_ACCUM) {
    const bool vec = (EPI == EPI_STORE) && !(a.N & 3) && !(a.ldc & 3);
#pragma unroll
    for (int i = 0; i < FI; ++i) {
      const int64_t row = m0 + wm * (FI * 16) + i * 16 + arow;
      if (row >= a.M) continue;
#pragma unroll
      for (int jn = 0; jn < 4; ++jn) {
        const int col = n0 + wn * 64 + jn * 16 + ag * 4;
        if (col >= a.N) continue;
        float* dst = a.C + row * a.ldc + col;
        if (vec) {   // (N % 4 == 0: the four columns are all inside)
          f32x4 v = acc[i][jn];
          if (a.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += a.bias[col + r];
          }
          *(f32x4*)dst = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (col + r >= a.N) continue;
            if (EPI == EPI_STORE) dst[r] = acc[i][jn][r] + (a.bias ? a.bias[col + r] : 0.f);
            else if (a.use_atomic) unsafeAtomicAdd(dst + r, acc[i][jn][r]);
            else dst[r] += acc[i][jn][r];
          }
        }
      }
    }
  } else {
    const int u = nt_idx * HC + wn * 16 + ag * 4;   // H % 8 == 0: u < H puts u .. u + 3 inside
    if (u < a.H) {
      f32x4 bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) bq[q][r] = a.bias[q * a.H + u + r];   // (the flat parameter vector's offsets are not 16-byte aligned)
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        const int64_t row = m0 + wm * (FI * 16) + i * 16 + arow;
        if (row >= a.M) continue;
        f32x4 cp = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.cprev) cp = *(const f32x4*)(a.cprev + row * a.ldh + u);
        f32x4 cc, hh;
        bf16x4 gi, gg4, gf, go, hb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ig = sigm(acc[i][0][r] + bq[0][r]);
          const float gg = tanh_fast(acc[i][1][r] + bq[1][r]);
          const float fg = sigm(acc[i][2][r] + bq[2][r]);
          const float og = sigm(acc[i][3][r] + bq[3][r]);
          cc[r] = fg * cp[r] + ig * gg;
          hh[r] = og * tanh_fast(cc[r]);
          gi[r] = tobf(ig); gg4[r] = tobf(gg); gf[r] = tobf(fg); go[r] = tobf(og); hb[r] = tobf(hh[r]);
        }
        *(f32x4*)(a.cout + row * a.ldh + u) = cc;
        *(bf16x4*)(a.hout + row * a.ldh + u) = hb;
        if (a.hout_f32) *(f32x4*)(a.hout_f32 + row * a.ldh + u) = hh;
        if (a.act) {
          bf16* g = a.act + row * (int64_t)4 * a.H + u;
          *(bf16x4*)(g) = gi; *(bf16x4*)(g + a.H) = gg4; *(bf16x4*)(g + 2 * a.H) = gf; *(bf16x4*)(g + 3 * a.H) = go;
        }
      }
    }
  }
}

template <int EPI, int BIG>
static void launch16(hipStream_t s, GArgs a, int split_k) {
  const size_t lds_bytes = (size_t)4 * Geo<BIG>::TILE * sizeof(bf16);
  static PerDeviceOnce attr_done;
  if (attr_done.need()) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_gemm16<EPI, BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  }
  a.nsplit = split_k;
  dim3 grid((unsigned)(((a.mtiles + 7) / 8) * 8 * a.ntiles));
  if (split_k > 1) grid = dim3((unsigned)(((split_k + 7) / 8) * 8 * a.mtiles * a.ntiles));
  hipLaunchKernelGGL((k_gemm16<EPI, BIG>), grid, dim3(Geo<BIG>::NT), lds_bytes, s, a);
  HIP_TRY(hipGetLastError());
}

// the 256-wide tile needs enough of them to fill the 256 CUs a few times over (one block per CU)
static bool big_tiles(int64_t M, int64_t N) {
  if (const char* e = getenv("KPRN_BF16_TILE")) {   // "big" | "small": the parity tests run both geometries at sizes the oracle handles
    if (e[0] == 'b') return true;
    if (e[0] == 's') return false;
  }
  return ((M + 255) / 256) * ((N + 255) / 256) >= 512;
}

// C[M][N] (fp32) = or += A[M][K] B[N][K]^T; K, lda, ldb multiples of 8 elements, 16-byte aligned pointers
static const bf16* zero16() {   // 16 zero bytes on the device: the source of every out-of-range DMA piece
  static bf16* z = nullptr;
  if (!z) {
    void* p = nullptr;
    if (hipMalloc(&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
      throw KprnError{KPRN_E_NOMEM, "hipMalloc failed (zero block)"};
    z = (bf16*)p;
  }
  return z;
}
namespace gx { struct XArgs; template <bool ACCUM, int DBG> __global__ void k_gemm16x(XArgs a); }
static bool gemm16x(hipStream_t s, const bf16* A, int64_t lda, const bf16* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int64_t K, bool accumulate,
                    int split_k, int n_lo = 0, int64_t k_lo = 0, int sx_min = 1);

// C[M][N] (fp32) = or += A[M][K] B[N][K]^T; K, lda, ldb multiples of 8 elements, 16-byte aligned pointers
static void gemm16(hipStream_t s, const bf16* A, int64_t lda, const bf16* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int64_t K,
                   bool accumulate, const float* bias, int split_k) {
  if (!bias && gemm16x(s, A, lda, B, ldb, C, ldc, M, N, K, accumulate, split_k)) return;   // the LDS-DMA kernel (large products without a bias)
  GArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.K = K; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.bias = bias;
  if (split_k < 1 || !accumulate) split_k = 1;
  const bool big = split_k == 1 && big_tiles(M, N);
  const int tm = big ? 256 : 128;
  a.mtiles = (M + tm - 1) / tm; a.ntiles = (N + tm - 1) / tm;
  if (split_k > 1) {
    const int64_t tiles = a.mtiles * a.ntiles;
    split_k = (int)std::max<int64_t>(1, std::min<int64_t>(split_k, (3 * 256 + tiles - 1) / tiles));
  }
  int64_t kchunk = (K + split_k - 1) / split_k;
  kchunk = ((kchunk + TK - 1) / TK) * TK;
  split_k = (int)((K + kchunk - 1) / kchunk);
  a.kchunk = kchunk; a.use_atomic = split_k > 1 ? 1 : 0;
  if (big) { if (accumulate) launch16<EPI_ACCUM, 1>(s, a, 1); else launch16<EPI_STORE, 1>(s, a, 1); }
  else if (accumulate) launch16<EPI_ACCUM, 0>(s, a, split_k); else launch16<EPI_STORE, 0>(s, a, split_k);
}

// ---- the same product on 256 x 128 x 64 tiles with LDS-DMA operand staging (round 3) ------------------------------------------------------
// k_gemm16 above moves its operands global -> registers -> LDS (8 ds_write_b128 per thread and chunk) on 128 x 128 tiles and reaches 0.22 of the
// bf16 peak on the backward's products (dx, dh, split-K dW: 3.3 ms of configs[3]'s 8.6 ms step).  Here: 8 waves (4 x 2, 64 x 64 each on
// v_mfma_f32_32x32x16_bf16), three LDS stages of 48 KB filled by global_load_lds_dwordx4 two chunks ahead (counted vmcnt, one barrier per chunk,
// no staging registers, no ds_write), and a third fewer operand bytes per flop through the 64 B/clk L1 path, which is what bounds a 128 x 128 tile.
// LDS image: rows of 128 bytes (64 k); the DMA writes lane l at +16 l, i.e. 8 rows x 8 pieces per instruction, so the XOR swizzle that keeps
// ds_read_b128 of a 32-row fragment conflict-free is applied to the SOURCE piece (piece = slot ^ ((row >> 1) & 7)) and again to the read address.
namespace gx {
constexpr int BM = 256, BN = 128, BK = 64, NSTAGE = 3, NTHR = 512;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
struct XArgs {
  const bf16* A; int64_t lda; const bf16* B; int64_t ldb; float* C; int64_t ldc; int64_t M; int N; int64_t K;
  int64_t kchunk; int nsplit; int64_t mtiles; int ntiles; const bf16* zero;   // zero: 16 zero bytes in global memory (source of every out-of-range piece)
  int touch;                // > 0: every wave touches the cache lines of the chunk `touch` chunks ahead (one dword per tile row: an L2 prefetch; k_gemm16x)
  int n_lo; int64_t k_lo;   // rows n >= n_lo of B are known to be ZERO for k < k_lo (the merged dW product's h_{t-1}^T block has no step -1): column tiles from n_lo on start at k_lo
};
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
typedef float f32x16 __attribute__((ext_vector_type(16)));

// DBG (measurement build only, KPRN_GEMM16_DBG): 1 no MFMAs, 2 no fragment reads either, 4 no DMA (the stages hold whatever they hold), 8 no epilogue,
// 16 the DMA pieces come from where a K-blocked layout [K / 64][rows][64] would hold them (tile rows 128 bytes apart instead of a row pitch apart; the same bytes,
// instructions and reuse between tiles: what the row-major layout costs in address translation / DRAM page locality), 32 every chunk re-reads the first 1 KB of
// its rows (everything an L2 hit: the delivery rate L2 -> LDS of this access shape, no HBM latency in it)
template <bool ACCUM, int DBG = 0>
__global__ __launch_bounds__(NTHR, 2) void k_gemm16x(XArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, kg = lane >> 5;
  const int64_t id = blockIdx.x;
  const int xcd = (int)(id & 7);
  const int64_t j = id >> 3;
  int nt_idx; int64_t mt_idx, split_idx = 0;
  if (a.nsplit > 1) {   // all tiles of one K range on one XCD
    const int64_t tiles = a.mtiles * a.ntiles;
    split_idx = (j / tiles) * 8 + xcd;
    const int64_t tl = j % tiles;
    mt_idx = tl / a.ntiles; nt_idx = (int)(tl % a.ntiles);
    if (split_idx >= a.nsplit) return;
  } else {              // the n-tiles of one m-tile back to back on one XCD: its A rows are fetched into that L2 once
    nt_idx = (int)(j % a.ntiles);
    mt_idx = (j / a.ntiles) * 8 + xcd;
    if (mt_idx >= a.mtiles) return;
  }
  const int64_t m0 = mt_idx * BM;
  const int n0 = nt_idx * BN;
  int64_t k_beg = split_idx * a.kchunk;
  const int64_t k_end = (k_beg + a.kchunk < a.K) ? k_beg + a.kchunk : a.K;
  if (a.k_lo > 0 && n0 >= a.n_lo && k_beg < a.k_lo) k_beg = a.k_lo;   // (k_lo is a multiple of 8: pieces stay 16-byte aligned)
  const int nch = (k_end > k_beg) ? (int)((k_end - k_beg + BK - 1) / BK) : 0;
  if (nch == 0) return;
  // this lane's share of a chunk's 48 DMA instructions (8 rows x 128 bytes each): instruction q = wave + 8 i covers tile rows 8 q .. 8 q + 7
  // (A: q < 32, B: q >= 32); the lane fetches piece (slot ^ key(row)) of row 8 q + (lane >> 3)
  const bf16* rowp[6]; int piece[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = wave + 8 * i;
    const bool isA = q < BM / 8;
    const int row = 8 * (isA ? q : q - BM / 8) + (lane >> 3);
    const int64_t gr = (isA ? m0 : (int64_t)n0) + row;
    const bool ok = gr < (isA ? a.M : (int64_t)a.N);
    rowp[i] = ok ? (isA ? a.A + gr * a.lda : a.B + gr * a.ldb) : nullptr;
    piece[i] = (lane & 7) ^ ((row >> 1) & 7);
  }
  const unsigned lds0 = lds_off(smem);
  auto issue = [&](int c) {
    const int64_t k0 = k_beg + (int64_t)c * BK;
    const unsigned st = lds0 + (unsigned)(c % NSTAGE) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int64_t k = k0 + 8 * piece[i];
      const bf16* src = (rowp[i] && k < k_end) ? rowp[i] + k : a.zero;
      if constexpr ((DBG & 32) != 0) {   // every chunk from the first 1 KB of its row: 2 176 rows x 1 KB stay in every L2 -- the L2-hit delivery rate of this access shape
        if (rowp[i] && k < k_end) src = rowp[i] + (k & 511);
      }
      if constexpr ((DBG & 16) != 0) {   // the K-blocked layout [K / 64][rows][64], emulated inside the same allocations: the same reuse between tiles, rows 128 bytes apart
        if (rowp[i] && k < k_end) {
          const bool isA = (wave + 8 * i) < BM / 8;
          const int64_t gr = isA ? (rowp[i] - a.A) / a.lda : (rowp[i] - a.B) / a.ldb;
          src = (isA ? a.A + (k0 >> 6) * (a.M * 64) : a.B + (k0 >> 6) * ((int64_t)a.N * 64)) + gr * 64 + 8 * piece[i];
        }
      }
      dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + (unsigned)(wave + 8 * i) * 1024u)));
    }
  };
  // L2 prefetch by touch (round 5).  The knock-outs (scripts/gpu_gemm16_knockouts.py) show this launch bound by its operand staging, not by its MFMAs: DMA + waits +
  // barriers alone take 0.82 of the 1.06 ms, MFMAs + fragment reads alone 0.39 -- two 48 KB chunks in flight per CU against ~2 us of HBM latency is 20 B/clk per CU.
  // LDS cannot hold more stages; the L2 can hold the lines: a chunk row is exactly one 128-byte line, so ONE 4-byte load per tile row, `touch` chunks ahead, brings
  // the chunk into this XCD's L2 long before its DMA asks for it.  Thread L < 384 owns tile row L (A rows, then B rows); the loaded dword is never looked at and
  // must not land in a register (an asynchronous load into a VGPR that hipcc believes dead overwrites whatever it put there next -- the first build of this
  // faulted): it is a 4-byte LDS-DMA into a scratch strip behind the stages.  Issued unconditionally (a clamped address): the counted waits below allow for
  // exactly one more load per chunk in flight for waves 0-5.
  const int trow = wave * 64 + lane;
  const bf16* tptr = a.zero;
  if (a.touch > 0 && trow < BM + BN) {
    const bool isA = trow < BM;
    const int64_t gr = isA ? m0 + trow : (int64_t)n0 + (trow - BM);
    if (gr < (isA ? a.M : (int64_t)a.N)) tptr = (isA ? a.A + gr * a.lda : a.B + gr * a.ldb) + k_beg;
  }
  const bool toucher = a.touch > 0 && wave < (BM + BN) / 64;   // (wave-uniform)
  auto touch = [&](int c) {
    if (!toucher) return;
    const int64_t k = k_beg + (int64_t)c * BK;
    const bf16* p = (tptr != a.zero && k < k_end) ? tptr + (int64_t)c * BK : a.zero;
    unsigned keep;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)NSTAGE * STAGE_BYTES + (unsigned)wave * 256u));
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][jn][q] = 0.f;
  if (toucher) {   // the lines of the first chunks beyond the two requested below
    for (int c = 2; c < 2 + a.touch && c < nch; ++c) touch(c);
  }
  if (!(DBG & 4)) {
    issue(0);
    if (nch > 1) issue(1);
  }
  const int key = (r >> 1) & 7;
  const int arow = (wm * 64 + r) * 128, brow = BM * 128 + (wn * 64 + r) * 128;
  for (int c = 0; c < nch; ++c) {
    // chunk c has landed for this wave (its own 6 pieces; the 6 most recent ones belong to chunk c + 1 -- plus, for a touching wave, the one touch issued behind
    // them), then for every wave; and every wave has finished reading the stage chunk c + 2 is about to overwrite (it held chunk c - 1)
    if (c + 1 < nch) { if (toucher && c > 0) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    touch(c + 2 + a.touch);   // (unconditional for a touching wave: a chunk past the range touches the zero block; BEFORE the DMA, so that exactly this one
                              //  touch and the next chunk's six pieces are younger than the pieces the next iteration waits for)
    if (c + 2 < nch && !(DBG & 4)) issue(c + 2);
    const char* st = smem + (c % NSTAGE) * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int po = ((2 * kk + kg) ^ key) << 4;
      bf16x8 fa[2], fb[2];
      if constexpr ((DBG & 2) != 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { fa[i] = bf16x8{}; fb[i] = bf16x8{}; asm volatile("" : "+v"(fa[i]), "+v"(fb[i])); }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *(const bf16x8*)(st + arow + i * 32 * 128 + po);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) fb[jn] = *(const bf16x8*)(st + brow + jn * 32 * 128 + po);
      }
      if constexpr ((DBG & 1) != 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("" :: "v"(fa[i]), "v"(fb[i]));
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jn], acc[i][jn], 0, 0, 0);
      }
    }
  }
  if constexpr ((DBG & 8) != 0) {
    if (acc[0][0][0] == 123.456f) a.C[0] = acc[1][1][3];
    return;
  }
  // D[m][n]: lane (n = lane & 31, kg), register q <-> row (q & 3) + 8 (q >> 2) + 4 kg: 32 consecutive columns per store instruction
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
      const int n = n0 + wn * 64 + jn * 32 + r;
      if (n >= a.N) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
        if (m >= a.M) continue;
        float* dst = a.C + m * a.ldc + n;
        if (ACCUM) unsafeAtomicAdd(dst, acc[i][jn][q]); else *dst = acc[i][jn][q];
      }
    }
}

// ---- the same tiles with the A operand handed over K-MAJOR: C[m][n] = sum_k AT[k][m] B[n][k]  (round 4) ----------------------------------------------
// dx = dA W_i2g with dA as the persistent BPTT launch leaves it for the dW products anyway: TRANSPOSED, [gate column][step][path] (lstm_bf16_bwd_persist.hip).
// Reading it here saves that launch its second, row-major copy of dA (1.2 GB per step of configs[3], written in 16-byte pieces scattered over 64 rows per
// wave instruction: 0.46 ms of its 1.35).  The A stage is the k-major image itself: 64 k-rows of 256 paths (512 bytes), filled by the same LDS-DMA
// (one instruction = two k-rows), and the fragment of v_mfma_f32_32x32x16_bf16 -- lane (m, kg) holds A[m][8 kg .. + 7] -- is formed by two
// ds_read_b64_tr_b16 per lane: inside a 16-lane group, lane p points at 4 consecutive paths of k-row (p >> 2) and gets back the 4 k-rows of ITS path
// (probed: scripts/ubench/tr16_probe.hip).  Bank conflicts: the 32 lanes of a phase touch 4 k-rows x 64 bytes, and 512-byte rows put those on the
// same banks; the 64-byte unit a piece lands in is therefore XORed with (k & 3) -- on the SOURCE side (the DMA writes LDS linearly) and again in the
// read address.  Column m' of AT is (step, path) with Np >= N paths per step (pad columns): the epilogue maps it to row step N + path of C.
struct XTArgs {
  const bf16* AT; int64_t ldat; const bf16* B; int64_t ldb; float* C; int64_t ldc; int64_t Mp; int N; int64_t K;
  int64_t mtiles; int ntiles; int64_t Np, Nv; const bf16* zero;
};
__global__ __launch_bounds__(NTHR, 2) void k_gemm16xt(XTArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, kg = lane >> 5;
  const int64_t id = blockIdx.x;
  const int xcd = (int)(id & 7);
  const int64_t j = id >> 3;
  const int nt_idx = (int)(j % a.ntiles);              // the n-tiles of one m-tile back to back on one XCD: its A columns are fetched into that L2 once
  const int64_t mt_idx = (j / a.ntiles) * 8 + xcd;
  if (mt_idx >= a.mtiles) return;
  const int64_t m0 = mt_idx * BM;
  const int n0 = nt_idx * BN;
  const int nch = (int)((a.K + BK - 1) / BK);
  // this lane's share of a chunk's 48 DMA instructions.  A (q < 32): instruction q = k-rows 2 q, 2 q + 1; the lane fills physical slot lane & 31 of
  // k-row 2 q + (lane >> 5) with the piece (8 paths) whose 64-byte unit is (slot >> 2) ^ (k & 3).  B (q >= 32): 8 rows x 128 bytes, as k_gemm16x.
  const bf16* src0[6]; int64_t step[6]; int kofs[6]; bool isa[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = wave + 8 * i;
    isa[i] = q < 32;
    if (isa[i]) {
      const int krow = 2 * q + (lane >> 5), ps = lane & 31;
      const int ls = ((((ps >> 2) ^ (krow & 3))) << 2) | (ps & 3);
      const int64_t m = m0 + 8 * ls;
      kofs[i] = krow;
      src0[i] = (m < a.Mp) ? a.AT + (int64_t)krow * a.ldat + m : nullptr;
      step[i] = (int64_t)BK * a.ldat;
    } else {
      const int row = 8 * (q - 32) + (lane >> 3);
      const int piece = (lane & 7) ^ ((row >> 1) & 7);
      kofs[i] = 8 * piece;
      src0[i] = (n0 + row < a.N) ? a.B + (int64_t)(n0 + row) * a.ldb + 8 * piece : nullptr;
      step[i] = BK;
    }
  }
  const unsigned lds0 = lds_off(smem);
  auto issue = [&](int c) {
    const unsigned st = lds0 + (unsigned)(c % NSTAGE) * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const bf16* src = (src0[i] && (int64_t)c * BK + kofs[i] < a.K) ? src0[i] + (int64_t)c * step[i] : a.zero;
      dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + (unsigned)(wave + 8 * i) * 1024u)));
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][jn][q] = 0.f;
  issue(0);
  if (nch > 1) issue(1);
  // A fragment addresses: lane p of a 16-lane group -> k-row (p >> 2) of the read's four, paths mb + 4 (p & 3) .. + 3 of the wave's 32-row fragment i
  const int p16 = lane & 15, mb = r & 16;
  int aoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ml = wm * 64 + i * 32 + mb + 4 * (p16 & 3);
    const int ls = ml >> 3, ps = ((((ls >> 2) ^ (p16 >> 2))) << 2) | (ls & 3);
    aoff[i] = (8 * kg + (p16 >> 2)) * 512 + ps * 16 + (ml & 7) * 2;
  }
  const int key = (r >> 1) & 7;
  const int brow = BM * 128 + (wn * 64 + r) * 128;
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  for (int c = 0; c < nch; ++c) {
    if (c + 1 < nch) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (c + 2 < nch) issue(c + 2);
    const char* st = smem + (c % NSTAGE) * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int po = ((2 * kk + kg) ^ key) << 4;
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + aoff[i] + (16 * kk) * 512));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(st + aoff[i] + (16 * kk + 4) * 512));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fa[i] = __builtin_bit_cast(bf16x8, both);
      }
#pragma unroll
      for (int jn = 0; jn < 2; ++jn) fb[jn] = *(const bf16x8*)(st + brow + jn * 32 * 128 + po);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jn], acc[i][jn], 0, 0, 0);
    }
  }
  const bool same = a.Np == a.Nv;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
      const int n = n0 + wn * 64 + jn * 32 + r;
      if (n >= a.N) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t mp = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
        if (mp >= a.Mp) continue;
        int64_t row = mp;
        if (!same) {   // pad columns between the steps (N not a multiple of 8): 32-bit arithmetic (host: Mp < 2^32)
          const unsigned tq = (unsigned)mp / (unsigned)a.Np, path = (unsigned)mp - tq * (unsigned)a.Np;
          if ((int64_t)path >= a.Nv) continue;
          row = (int64_t)tq * a.Nv + path;
        }
        a.C[row * a.ldc + n] = acc[i][jn][q];
      }
    }
}

// ---- 256 x 192 x 32 tiles for N = a multiple of 192 (configs[3]: N = 384); opt-in (KPRN_BF16_GEMM=y), a measured non-improvement ------------------
// The idea: k_gemm16x is balanced on the LDS pipe: a 64 x 64 per-wave tile reads 1 KiB of LDS per v_mfma_f32_32x32x16_bf16 and the CU's 128 B/clk feed exactly
// four SIMDs at 32 cycles per MFMA.  Here a wave owns 64 x 96 (2 x 3 MFMA tiles: 0.83 KiB per MFMA), N = 384 is two column tiles instead of three (A is
// fetched twice, not three times), and four 28 KB stages of 32 k keep three chunks in flight.  LDS rows are 64 bytes (4 pieces of 16): one DMA
// instruction covers 16 rows x 4 pieces, lane l -> row l >> 2, slot l & 3, and fetches piece slot ^ ((row >> 1) & 3) -- eight consecutive rows then
// occupy the eight 16-byte granules of a 128-byte LDS phase exactly once for any piece, which is what a ds_read_b128 of a 32-row fragment touches.
// A stage is 28 instructions: waves 0-3 issue four of them (q = w, w + 8, w + 16, w + 24), waves 4-7 three -- the counted vmcnt differs by wave.
constexpr int YBM = 256, YBN = 192, YBK = 32, YSTAGE = 4;
constexpr int YSTAGE_BYTES = (YBM + YBN) * YBK * 2;
template <bool ACCUM>
__global__ __launch_bounds__(NTHR, 2) void k_gemm16y(XArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, kg = lane >> 5;
  const int64_t id = blockIdx.x;
  const int xcd = (int)(id & 7);
  const int64_t j = id >> 3;
  int nt_idx; int64_t mt_idx, split_idx = 0;
  if (a.nsplit > 1) {   // all tiles of one K range on one XCD
    const int64_t tiles = a.mtiles * a.ntiles;
    split_idx = (j / tiles) * 8 + xcd;
    const int64_t tl = j % tiles;
    mt_idx = tl / a.ntiles; nt_idx = (int)(tl % a.ntiles);
    if (split_idx >= a.nsplit) return;
  } else {              // the n-tiles of one m-tile back to back on one XCD: its A rows are fetched into that L2 once
    nt_idx = (int)(j % a.ntiles);
    mt_idx = (j / a.ntiles) * 8 + xcd;
    if (mt_idx >= a.mtiles) return;
  }
  const int64_t m0 = mt_idx * YBM;
  const int n0 = nt_idx * YBN;
  const int64_t k_beg = split_idx * a.kchunk;
  const int64_t k_end = (k_beg + a.kchunk < a.K) ? k_beg + a.kchunk : a.K;
  const int nch = (k_end > k_beg) ? (int)((k_end - k_beg + YBK - 1) / YBK) : 0;
  if (nch == 0) return;
  const int nq = wave < 4 ? 4 : 3;   // (wave-uniform) DMA instructions of this wave per chunk
  const bf16* rowp[4]; int piece[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = wave + 8 * i;              // instruction q covers tile rows 16 q .. 16 q + 15 (A: q < 16, B: 16 <= q < 28)
    const bool isA = q < YBM / 16;
    const int row = 16 * (isA ? q : q - YBM / 16) + (lane >> 2);
    const int64_t gr = (isA ? m0 : (int64_t)n0) + row;
    const bool ok = q < (YBM + YBN) / 16 && gr < (isA ? a.M : (int64_t)a.N);
    rowp[i] = ok ? (isA ? a.A + gr * a.lda : a.B + gr * a.ldb) : nullptr;
    piece[i] = (lane & 3) ^ ((row >> 1) & 3);
  }
  const unsigned lds0 = lds_off(smem);
  auto issue = [&](int c) {
    const int64_t k0 = k_beg + (int64_t)c * YBK;
    const unsigned st = lds0 + (unsigned)(c % YSTAGE) * YSTAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i == 3 && wave >= 4) break;   // (wave-uniform)
      const int64_t k = k0 + 8 * piece[i];
      const bf16* src = (rowp[i] && k < k_end) ? rowp[i] + k : a.zero;
      dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + (unsigned)(wave + 8 * i) * 1024u)));
    }
  };
  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 3; ++jn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][jn][q] = 0.f;
  issue(0);
  if (nch > 1) issue(1);
  if (nch > 2) issue(2);
  const int key = (r >> 1) & 3;
  const int arow = (wm * 64 + r) * 64, brow = YBM * 64 + (wn * 96 + r) * 64;
  for (int c = 0; c < nch; ++c) {
    // chunk c has landed for this wave: at most the pieces of the chunks behind it (up to two) may still be in flight
    const int behind = (nch - 1 - c) < 2 ? (nch - 1 - c) : 2;
    if (nq == 4) {
      if (behind == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (behind == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else if (behind == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... for every wave; and every wave is done reading the stage chunk c + 3 overwrites
    if (c + 3 < nch) issue(c + 3);
    const char* st = smem + (c % YSTAGE) * YSTAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int po = ((2 * kk + kg) ^ key) << 4;
      bf16x8 fa[2], fb[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *(const bf16x8*)(st + arow + i * 32 * 64 + po);
#pragma unroll
      for (int jn = 0; jn < 3; ++jn) fb[jn] = *(const bf16x8*)(st + brow + jn * 32 * 64 + po);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 3; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jn], acc[i][jn], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 3; ++jn) {
      const int n = n0 + wn * 96 + jn * 32 + r;
      if (n >= a.N) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
        if (m >= a.M) continue;
        float* dst = a.C + m * a.ldc + n;
        if (ACCUM) unsafeAtomicAdd(dst, acc[i][jn][q]); else *dst = acc[i][jn][q];
      }
    }
}

// ---- the x tile with its operands staged through REGISTERS, four chunks in flight (round 5) --------------------------------------------------------------------------
// Knock-outs of k_gemm16x on the merged dW product (scripts/gpu_gemm16_knockouts.py, profiles/r05/gemm16_knockouts.txt): DMA + waits + barriers alone 0.82 of the
// 1.06 ms, MFMAs + fragment reads alone 0.39 -- the launch is bound by its operand staging, and that staging by LATENCY: two 48 KB chunks in flight per CU against
// ~2 us from HBM is 20 bytes per clock and CU, a third of what the L1 path carries.  LDS cannot hold more stages (3 x 48 KB).  Registers can: here every thread fetches
// its six 16-byte pieces of a chunk into VGPRs THREE chunks before it writes them to LDS (ds_write_b128, the same XOR swizzle applied to the write address), so four
// chunks = 192 KB per CU are in flight; LDS is a plain double buffer, one barrier per chunk.  No inline-asm loads: hipcc counts its own vmcnt exactly (the chunk loop
// is unrolled by the ring's period so that the ring is statically indexed), fences keep the loads where they are written.
constexpr int RRING = 4;   // chunk buffers per thread (3 ahead + the one being written)
template <bool ACCUM>
__global__ __launch_bounds__(NTHR, 2) void k_gemm16r(XArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, kg = lane >> 5;
  const int64_t id = blockIdx.x;
  const int xcd = (int)(id & 7);
  const int64_t j = id >> 3;
  int nt_idx; int64_t mt_idx, split_idx = 0;
  if (a.nsplit > 1) {
    const int64_t tiles = a.mtiles * a.ntiles;
    split_idx = (j / tiles) * 8 + xcd;
    const int64_t tl = j % tiles;
    mt_idx = tl / a.ntiles; nt_idx = (int)(tl % a.ntiles);
    if (split_idx >= a.nsplit) return;
  } else {
    nt_idx = (int)(j % a.ntiles);
    mt_idx = (j / a.ntiles) * 8 + xcd;
    if (mt_idx >= a.mtiles) return;
  }
  const int64_t m0 = mt_idx * BM;
  const int n0 = nt_idx * BN;
  int64_t k_beg = split_idx * a.kchunk;
  const int64_t k_end = (k_beg + a.kchunk < a.K) ? k_beg + a.kchunk : a.K;
  if (a.k_lo > 0 && n0 >= a.n_lo && k_beg < a.k_lo) k_beg = a.k_lo;
  const int nch = (k_end > k_beg) ? (int)((k_end - k_beg + BK - 1) / BK) : 0;
  if (nch == 0) return;
  // piece i of this thread: tile row 8 q + (lane >> 3) of instruction slot q = wave + 8 i (A: q < 32, B: q >= 32), 16-byte slot lane & 7 of the row's 128 bytes
  const bf16* src[6]; unsigned wofs[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int q = wave + 8 * i;
    const bool isA = q < BM / 8;
    const int row = 8 * (isA ? q : q - BM / 8) + (lane >> 3);
    const int64_t gr = (isA ? m0 : (int64_t)n0) + row;
    const bool ok = gr < (isA ? a.M : (int64_t)a.N);
    src[i] = ok ? (isA ? a.A + gr * a.lda : a.B + gr * a.ldb) + k_beg + 8 * (lane & 7) : nullptr;
    wofs[i] = (unsigned)((isA ? 0 : BM * 128) + row * 128 + (((lane & 7) ^ ((row >> 1) & 7)) << 4));
  }
  bf16x8 ring[RRING][6];
  auto fetch = [&](int c, bf16x8 (&dst)[6]) {   // chunk c of this thread's rows (pieces past the K range / the problem: the zero block)
    const int64_t kk0 = (int64_t)c * BK;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const bf16* p = (src[i] && c < nch && k_beg + kk0 + 8 * (lane & 7) < k_end) ? src[i] + kk0 : a.zero;
      dst[i] = *(const bf16x8*)p;
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto stash = [&](int stage, const bf16x8 (&v)[6]) {
    char* st = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < 6; ++i) *(bf16x8*)(st + wofs[i]) = v[i];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][jn][q] = 0.f;
  const int key = (r >> 1) & 7;
  const int arow = (wm * 64 + r) * 128, brow = BM * 128 + (wn * 64 + r) * 128;
  auto product = [&](int stage) {
    const char* st = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int po = ((2 * kk + kg) ^ key) << 4;
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[i] = *(const bf16x8*)(st + arow + i * 32 * 128 + po);
#pragma unroll
      for (int jn = 0; jn < 2; ++jn) fb[jn] = *(const bf16x8*)(st + brow + jn * 32 * 128 + po);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[jn], acc[i][jn], 0, 0, 0);
    }
  };
  // prologue: chunks 0 .. 3 requested, chunk 0 written to stage 0
  fetch(0, ring[0]); fetch(1, ring[1]); fetch(2, ring[2]); fetch(3, ring[3]);
  stash(0, ring[0]);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  // iteration c: chunk c + 4 is requested into the ring slot chunk c just left, chunk c + 1 goes from its registers to the other stage, chunk c is multiplied
  for (int c0 = 0; c0 < nch; c0 += RRING) {
#pragma unroll
    for (int u = 0; u < RRING; ++u) {
      const int c = c0 + u;
      if (c < nch) {   // (workgroup-uniform)
        fetch(c + RRING, ring[u]);
        stash((c + 1) & 1, ring[(u + 1) % RRING]);
        __builtin_amdgcn_sched_barrier(0);
        product(c & 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
      const int n = n0 + wn * 64 + jn * 32 + r;
      if (n >= a.N) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
        if (m >= a.M) continue;
        float* dst = a.C + m * a.ldc + n;
        if (ACCUM) unsafeAtomicAdd(dst, acc[i][jn][q]); else *dst = acc[i][jn][q];
      }
    }
}

// ---- 256 x 256 x 32 tiles, two wave groups one barrier apart (round 5) --------------------------------------------------------------------------------
// What k_gemm16x / k_gemm16y share is their lockstep: every wave meets the same barrier, then every wave reads its fragments, then every wave issues its MFMAs --
// the two waves of a SIMD wait for LDS at the same moment and want the matrix pipe at the same moment (0.28-0.36 of the bf16 peak whatever the tile).  Here the
// workgroup's eight waves form two groups, G0 = waves 0-3 (rows 0-127 of the tile) and G1 = waves 4-7 (rows 128-255): wave w and w + 4 share a SIMD.  A K tile of
// 32 is one READ segment (this wave's DMA requests three K tiles ahead, 12 ds_read_b128: 8 A + 4 B fragments of v_mfma_f32_32x32x16_bf16, the counted wait for its
// own pieces of the next K tile) and one MFMA segment (16 MFMAs = 512 matrix cycles on a 128 x 64 wave tile), each closed by s_barrier -- and G1 runs ONE BARRIER
// BEHIND G0, so on every SIMD one wave reads while the other multiplies:
//     barrier index       B0        B1          B2          B3          B4
//     G0            | R(0) | M(0)     | R(1)     | M(1)      | R(2) ...
//     G1            |  --  | R(0)     | M(0)     | R(1)      | M(1) ...
// Four LDS buffers of 32 KB (A 256 x 32 | B 256 x 32 bf16; rows of 64 bytes, the 16-byte piece a lane fetches chosen on the SOURCE side so that a 32-row fragment
// read is conflict-free: as k_gemm16y).  Hazards: K tile kt + 1 is complete for everybody before G0's R(kt + 1) -- every wave waits for its own pieces of kt + 1 at
// the end of its R(kt) (G0: interval 2 kt, G1: 2 kt + 1) and a barrier follows both; the buffer DMA(kt + 3) overwrites held K tile kt - 1, last read by G1's R(kt - 1) in
// interval 2 kt - 1, which ends with lgkmcnt(0) + barrier before G0 requests it in interval 2 kt.
constexpr int PBM = 256, PBN = 256, PBK = 32, PNBUF = 4, PBUF_BYTES = (PBM + PBN) * PBK * 2;
template <bool ACCUM>
__global__ __launch_bounds__(NTHR, 2) void k_gemm16p(XArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wg = wave >> 2, wc = wave & 3, r = lane & 31, kg = lane >> 5;
  const int64_t id = blockIdx.x;
  const int xcd = (int)(id & 7);
  const int64_t j = id >> 3;
  int nt_idx; int64_t mt_idx, split_idx = 0;
  if (a.nsplit > 1) {   // all tiles of one K range on one XCD
    const int64_t tiles = a.mtiles * a.ntiles;
    split_idx = (j / tiles) * 8 + xcd;
    const int64_t tl = j % tiles;
    mt_idx = tl / a.ntiles; nt_idx = (int)(tl % a.ntiles);
    if (split_idx >= a.nsplit) return;
  } else {
    nt_idx = (int)(j % a.ntiles);
    mt_idx = (j / a.ntiles) * 8 + xcd;
    if (mt_idx >= a.mtiles) return;
  }
  const int64_t m0 = mt_idx * PBM;
  const int n0 = nt_idx * PBN;
  int64_t k_beg = split_idx * a.kchunk;
  const int64_t k_end = (k_beg + a.kchunk < a.K) ? k_beg + a.kchunk : a.K;
  if (a.k_lo > 0 && n0 >= a.n_lo && k_beg < a.k_lo) k_beg = a.k_lo;
  const int nkt = (k_end > k_beg) ? (int)((k_end - k_beg + PBK - 1) / PBK) : 0;
  if (nkt == 0) return;   // (workgroup-uniform: no barrier has been executed)
  // this lane's share of a K tile's 32 DMA instructions: instruction q = wave + 8 i covers tile rows 16 q .. 16 q + 15 (A: q < 16, B: q >= 16), lane -> row
  // 16 q' + (lane >> 2), slot lane & 3, piece slot ^ ((row >> 1) & 3)
  const bf16* rowp[4]; int piece[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = wave + 8 * i;
    const bool isA = q < PBM / 16;
    const int row = 16 * (isA ? q : q - PBM / 16) + (lane >> 2);
    const int64_t gr = (isA ? m0 : (int64_t)n0) + row;
    const bool ok = gr < (isA ? a.M : (int64_t)a.N);
    rowp[i] = ok ? (isA ? a.A + gr * a.lda : a.B + gr * a.ldb) : nullptr;
    piece[i] = (lane & 3) ^ ((row >> 1) & 3);
  }
  const unsigned lds0 = lds_off(smem);
  auto issue = [&](int kt) {
    const int64_t k0 = k_beg + (int64_t)kt * PBK;
    const unsigned st = lds0 + (unsigned)(kt & (PNBUF - 1)) * PBUF_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t k = k0 + 8 * piece[i];
      const bf16* src = (rowp[i] && k < k_end) ? rowp[i] + k : a.zero;
      dma16(src, (unsigned)__builtin_amdgcn_readfirstlane((int)(st + (unsigned)(wave + 8 * i) * 1024u)));
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][jn][q] = 0.f;
  const int key = (r >> 1) & 3;
  const int arow = (wg * 128 + r) * 64, brow = PBM * 64 + (wc * 64 + r) * 64;
  int po[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) po[s] = ((2 * s + kg) ^ key) << 4;
  // prologue: three K tiles requested, the first one complete for everybody
  issue(0);
  if (nkt > 1) issue(1);
  if (nkt > 2) issue(2);
  {
    const int behind = (nkt - 1) < 2 ? (nkt - 1) : 2;
    if (behind == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // B0
  if (wg == 1) asm volatile("s_barrier" ::: "memory");               // G1 falls one barrier behind
  for (int kt = 0; kt < nkt; ++kt) {
    // ---- READ segment
    if (kt + 3 < nkt) issue(kt + 3);
    const char* st = smem + (kt & (PNBUF - 1)) * PBUF_BYTES;
    bf16x8 fa[4][2], fb[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int jn = 0; jn < 2; ++jn) fb[jn][s] = *(const bf16x8*)(st + brow + jn * 32 * 64 + po[s]);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i][s] = *(const bf16x8*)(st + arow + i * 32 * 64 + po[s]);
    }
    {   // this wave's pieces of K tile kt + 1 have landed (the pieces of kt + 2, kt + 3 may still be in flight)
      const int last = nkt - 1;
      const int issued = (kt + 3 < last) ? kt + 3 : last;
      const int behind = issued - (kt + 1);
      if (behind >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- MFMA segment (the partner wave on this SIMD is in its READ segment)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][s], fb[jn][s], acc[i][jn], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_barrier" ::: "memory");
  }
  if (wg == 0) asm volatile("s_barrier" ::: "memory");   // G0 meets G1's last barrier
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) {
      const int n = n0 + wc * 64 + jn * 32 + r;
      if (n >= a.N) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t m = m0 + wg * 128 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kg;
        if (m >= a.M) continue;
        float* dst = a.C + m * a.ldc + n;
        if (ACCUM) unsafeAtomicAdd(dst, acc[i][jn][q]); else *dst = acc[i][jn][q];
      }
    }
}
}  // namespace gx

// kprn_set_option "bf16_gemm_pingpong" (default off): the split-K products of the bf16 backward on k_gemm16p where the shape suits it.  MEASURED SLOWER than the
// lockstep kernel (profiles/r05/bench_c4_m_*, bench_c4_n_*: merged dW 1.50 against 1.10 ms at 64 K ranges, 1.62 / 2.05 / 1.97 at 32 / 16 / 8; k_gemm16x 1.08-1.12
// at every count): neither the SIMD-level phase collision nor the number of atomic passes is what holds these products at 0.28 of the bf16 peak.  Kept as the
// record of that experiment; tests/test_gpu_persist.py holds it equal to k_gemm16x to fp32 reordering.
int g_t_pad = 64;                // kprn_set_option "bf16_t_pad": pad (elements, a multiple of 8, <= 512) of the row pitch of dA^T / Z^T on the small-table route
void set_t_pad(int n) { g_t_pad = n < 0 ? 0 : (n > 512 ? 512 : (n & ~7)); }
bool g_gemm16_pingpong = false;
// kprn_set_option "bf16_gemm_touch" (default 0 = off): chunks ahead of its DMA k_gemm16x touches a chunk's cache lines.  MEASURED SLOWER (merged dW 1.06 -> 1.64 ms
// at 3 / 6 / 12 chunks: loads retire in issue order, so a touch that misses to HBM holds back the retirement of every DMA piece issued behind it -- the prefetch
// sits on the critical path it was meant to shorten).  Kept as the record; the test holds it equal to the untouched launch.
int g_gemm16_touch = 0;
bool g_gemm16_regstage = false;  // kprn_set_option "bf16_gemm_regstage": the split-K products on k_gemm16r (operands through registers, four chunks in flight); opt-in, measured no faster
void set_gemm_regstage(bool on) { g_gemm16_regstage = on; }
void set_gemm_touch(int n) { g_gemm16_touch = n < 0 ? 0 : (n > 32 ? 32 : n); }
void set_gemm_pingpong(bool on) { g_gemm16_pingpong = on; }

static bool gemm16x(hipStream_t s, const bf16* A, int64_t lda, const bf16* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int64_t K, bool accumulate,
                    int split_k, int n_lo, int64_t k_lo, int sx_min) {
  static const bool off = KPRN_DEV_ENV("KPRN_BF16_GEMM") && KPRN_DEV_ENV("KPRN_BF16_GEMM")[0] == 'o';   // "old": k_gemm16 everywhere (A/B measurements, tests)
  if (off || M < gx::BM || N < gx::BN || (K & 7) || (lda & 7) || (ldb & 7)) return false;
  gx::XArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.zero = zero16();
  a.n_lo = n_lo; a.k_lo = k_lo;
  a.touch = g_gemm16_touch;
  // KPRN_BF16_GEMM=y: 256 x 192 tiles where they divide N (configs[3]: N = 384).  Measured equal to the 256 x 128 kernel on dx / dh (0.743 : 0.750,
  // 0.110 : 0.107 ms) and slower on the split-K dW (12 tiles deal worse over 8 XCDs than 18): opt-in, kept as the record of that measurement
  static const bool want_y = KPRN_DEV_ENV("KPRN_BF16_GEMM") && KPRN_DEV_ENV("KPRN_BF16_GEMM")[0] == 'y';
  const bool y = want_y && (N % gx::YBN) == 0 && k_lo == 0;
  if (split_k < 1 || !accumulate) split_k = 1;
  // the two-group kernel (k_gemm16p) for the split-K products whose output holds whole 256 x 256 tiles' worth of rows and at least one tile of columns
  // (configs[3]'s merged dW: 1 536 x 640 = 6 x 3 tiles, the last one half empty)
  const bool p = g_gemm16_pingpong && !y && accumulate && split_k > 1 && M >= gx::PBM && N >= gx::PBN;
  const int bn = p ? gx::PBN : (y ? gx::YBN : gx::BN), bk = p ? gx::PBK : (y ? gx::YBK : gx::BK), bm = p ? gx::PBM : gx::BM;
  a.mtiles = (M + bm - 1) / bm; a.ntiles = (N + bn - 1) / bn;
  if (split_k > 1) {
    // K ranges are dealt to the XCDs (range r on XCD r % 8, all its tiles there), one workgroup per CU (147 KB of LDS), 32 CUs per XCD:
    // with s ranges per XCD the launch takes ceil(tiles s / 32) rounds of 1 / (8 s) of K each.  Pick the s <= 8 with the least
    // rounds / s (configs[3]'s dW: 18 tiles -> s = 7: 126 workgroups = 3.94 rounds per XCD; the old "3 x 256 workgroups" rule gave 43
    // ranges = 6 on three XCDs, 4 rounds of 1 / 43: a quarter more time)
    const int64_t tiles = a.mtiles * a.ntiles;
    static const int s_env = KPRN_DEV_ENV("KPRN_GEMM16_SX") ? atoi(KPRN_DEV_ENV("KPRN_GEMM16_SX")) : 0;   // (measurement)
    int best = 1;
    double best_t = 1e30;
    for (int sx = std::max(1, std::min(sx_min, 8)); sx <= 8 && sx * 8 <= std::max(split_k, 8); ++sx) {
      const double t = (double)((tiles * sx + 31) / 32) / sx;
      if (t < best_t - 1e-9) { best_t = t; best = sx; }
    }
    if (s_env > 0) best = s_env;
    split_k = best * 8;
  }
  int64_t kchunk = (K + split_k - 1) / split_k;
  kchunk = ((kchunk + bk - 1) / bk) * bk;
  split_k = (int)((K + kchunk - 1) / kchunk);
  a.kchunk = kchunk; a.nsplit = split_k;
  const size_t lds_bytes = p ? (size_t)gx::PNBUF * gx::PBUF_BYTES : (y ? (size_t)gx::YSTAGE * gx::YSTAGE_BYTES : (size_t)gx::NSTAGE * gx::STAGE_BYTES + 2048 /* touch scratch */);
  static PerDeviceOnce attr_done;
  if (attr_done.need()) {
    HIP_TRY(hipFuncSetAttribute((const void*)gx::k_gemm16p<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)gx::PNBUF * gx::PBUF_BYTES)));
    HIP_TRY(hipFuncSetAttribute((const void*)gx::k_gemm16x<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)gx::NSTAGE * gx::STAGE_BYTES + 2048)));
    HIP_TRY(hipFuncSetAttribute((const void*)gx::k_gemm16x<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)gx::NSTAGE * gx::STAGE_BYTES + 2048)));
    HIP_TRY(hipFuncSetAttribute((const void*)gx::k_gemm16y<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)gx::YSTAGE * gx::YSTAGE_BYTES)));
    HIP_TRY(hipFuncSetAttribute((const void*)gx::k_gemm16y<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)gx::YSTAGE * gx::YSTAGE_BYTES)));
  }
  dim3 grid((unsigned)(((a.mtiles + 7) / 8) * 8 * a.ntiles));
  if (split_k > 1) grid = dim3((unsigned)(((split_k + 7) / 8) * 8 * a.mtiles * a.ntiles));
  if (!p && !y && accumulate && split_k > 1 && g_gemm16_regstage) {
    static PerDeviceOnce r_attr;
    const size_t lds_r = (size_t)2 * gx::STAGE_BYTES;
    if (r_attr.need()) { HIP_TRY(hipFuncSetAttribute((const void*)gx::k_gemm16r<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r)); }
    hipLaunchKernelGGL((gx::k_gemm16r<true>), grid, dim3(gx::NTHR), lds_r, s, a);
    HIP_TRY(hipGetLastError());
    return true;
  }
  if (p) hipLaunchKernelGGL((gx::k_gemm16p<true>), grid, dim3(gx::NTHR), lds_bytes, s, a);
  else if (y) {
    if (accumulate) hipLaunchKernelGGL((gx::k_gemm16y<true>), grid, dim3(gx::NTHR), lds_bytes, s, a);
    else hipLaunchKernelGGL((gx::k_gemm16y<false>), grid, dim3(gx::NTHR), lds_bytes, s, a);
  } else if (accumulate) {
#ifdef KPRN_PERSIST_VARIANTS
    // measurement build (scripts/gpu_gemm16_knockouts.py): KPRN_GEMM16_DBG = knock-out mask of the split-K launch
    if (const char* e = KPRN_DEV_ENV("KPRN_GEMM16_DBG")) {
      const int dbg = atoi(e);
      bool found = true;
#define KV(D) case D: HIP_TRY(hipFuncSetAttribute((const void*)gx::k_gemm16x<true, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
                      hipLaunchKernelGGL((gx::k_gemm16x<true, D>), grid, dim3(gx::NTHR), lds_bytes, s, a); break;
      switch (dbg) { KV(0) KV(1) KV(3) KV(4) KV(5) KV(7) KV(8) KV(9) KV(12) KV(16) KV(19) KV(32) KV(35) default: found = false; }
#undef KV
      KPRN_REQUIRE(found, KPRN_E_ARG, "this knock-out of k_gemm16x is not compiled in");
      HIP_TRY(hipGetLastError());
      return true;
    }
#endif
    hipLaunchKernelGGL((gx::k_gemm16x<true>), grid, dim3(gx::NTHR), lds_bytes, s, a);
  } else hipLaunchKernelGGL((gx::k_gemm16x<false>), grid, dim3(gx::NTHR), lds_bytes, s, a);
  HIP_TRY(hipGetLastError());
  return true;
}

// C[(step, path)][N] (fp32) = sum_k AT[k][step Np + path] B[n][k]: the k-major A operand (gx::k_gemm16xt); false = shape not covered
static bool dx_t_off() {   // KPRN_BF16_DX_T=0: dx from a row-major dA (tests/test_gpu_persist.py A/B)
  static const bool off = [] { const char* e = getenv("KPRN_BF16_DX_T"); return e && e[0] == '0'; }();
  return off;
}
// the ONE predicate of the k-major-A product: what gemm16xt takes is what dx_from_transposed_ok promises the BPTT launch
static bool xt_shape_ok(int64_t Mp, int N, int64_t K, int64_t Np, int64_t ldat, int64_t ldb) {
  return !dx_t_off() && Mp >= gx::BM && Mp < ((int64_t)1 << 32) && N >= gx::BN && !(K & 7) && !(ldat & 7) && !(ldb & 7) && !(Np & 7);
}
static bool gemm16xt(hipStream_t s, const bf16* AT, int64_t ldat, const bf16* B, int64_t ldb, float* C, int64_t ldc, int64_t Mp, int N, int64_t K, int64_t Np, int64_t Nv) {
  if (!xt_shape_ok(Mp, N, K, Np, ldat, ldb)) return false;
  gx::XTArgs a;
  memset(&a, 0, sizeof(a));
  a.AT = AT; a.ldat = ldat; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.Mp = Mp; a.N = N; a.K = K; a.Np = Np; a.Nv = Nv; a.zero = zero16();
  a.mtiles = (Mp + gx::BM - 1) / gx::BM; a.ntiles = (N + gx::BN - 1) / gx::BN;
  const size_t lds_bytes = (size_t)gx::NSTAGE * gx::STAGE_BYTES;
  static PerDeviceOnce attr_done;
  if (attr_done.need()) {
    HIP_TRY(hipFuncSetAttribute((const void*)gx::k_gemm16xt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  }
  hipLaunchKernelGGL(gx::k_gemm16xt, dim3((unsigned)(((a.mtiles + 7) / 8) * 8 * a.ntiles)), dim3(gx::NTHR), lds_bytes, s, a);
  HIP_TRY(hipGetLastError());
  return true;
}
bool dx_from_transposed_ok(int64_t Mp, int N, int64_t K, int64_t Np) {   // (the BPTT launch asks before it drops its row-major copy of dA)
  return xt_shape_ok(Mp, N, K, Np, /*ldat = the padded (step, path) extent*/ Mp, /*ldb = the gate columns*/ K);
}

// ---- element-wise / layout kernels --------------------------------------------------------------------------------------
__global__ void k_cvt(const float* __restrict__ x, bf16* __restrict__ y, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i + 3 < n) {
    const f32x4 v = *(const f32x4*)(x + i);
    bf16x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = tobf(v[q]);
    *(bf16x4*)(y + i) = o;
  } else {
    for (int64_t k = i; k < n; ++k) y[k] = tobf(x[k]);
  }
}
// y[c][r] = bf16(x[r][c]), x fp32 [R][C] (weights: small)
__global__ void k_cvt_T(const float* __restrict__ x, bf16* __restrict__ y, int R, int Cc) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)R * Cc) return;
  const int c = (int)(i / R), r = (int)(i - (int64_t)c * R);
  y[i] = tobf(x[(int64_t)r * Cc + c]);
}
// shadow rows of the entity table after a row update: rows[0 .. *count)
__global__ void k_rows_cvt(const float* __restrict__ W, bf16* __restrict__ W16, const int32_t* __restrict__ rows, const int32_t* __restrict__ count,
                           int d) {
  const int n = *count;
  const int per = d >> 2;  // float4 pieces per row
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)n * per; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows[i / per];
    const int c = (int)(i % per) * 4;
    const f32x4 v = *(const f32x4*)(W + r * d + c);
    bf16x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = tobf(v[q]);
    *(bf16x4*)(W16 + r * d + c) = o;
  }
}
// FeatureEmbedding (net/FeatureEmbedding.lua:112-121) on the bf16 shadows: X16[t][n][:] = [ sum_k Wt[type_k] | We[ent] | Wr[rel] ], 8 columns
// per thread; ids are int32 (never through a float: 20 M entities exceed 2^24).  Several type slots: summed in fp32, rounded once.
__global__ void k_gather16(const int32_t* __restrict__ idx, int64_t N, int T, int F, int nT, const bf16* __restrict__ Wt, const bf16* __restrict__ We,
                           const bf16* __restrict__ Wr, int dt, int de, int dr, bf16* __restrict__ X) {
  const int D = dt + de + dr, DV = D >> 3;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= N * T * DV) return;
  const int cv = (int)(gid % DV);
  const int64_t nt = gid / DV;
  const int64_t n = nt / T;
  const int t = (int)(nt - n * T);
  const int32_t* f = idx + nt * F;
  const int col = cv * 8;
  bf16x8 v;
  if (col < dt) {
    v = *(const bf16x8*)(Wt + (int64_t)(f[F - nT - 2] - 1) * dt + col);
    if (nT > 1) {
      float acc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = (float)v[q];
      for (int k = 1; k < nT; ++k) {
        const bf16x8 w = *(const bf16x8*)(Wt + (int64_t)(f[F - nT - 2 + k] - 1) * dt + col);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += (float)w[q];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = tobf(acc[q]);
    }
  } else if (col < dt + de) {
    v = *(const bf16x8*)(We + (int64_t)(f[F - 2] - 1) * de + (col - dt));
  } else {
    v = *(const bf16x8*)(Wr + (int64_t)(f[F - 1] - 1) * dr + (col - dt - de));
  }
  *(bf16x8*)(X + ((int64_t)t * N + n) * D + col) = v;
}
// The same gather written TRANSPOSED: XT[d][t][n] (pitch ldy = T Np, pad columns n >= N zero) -- the k-contiguous operand of the dW product.
// When the persistent layer kernel ran the training forward (it gathers for itself) the row-major plane X16 has no other reader, so the
// backward builds the transposed image straight from the shadow tables: 64 positions x 64 columns per workgroup through LDS, 16-byte accesses
// on both sides (k_gather16 + k_transpose16 of X: 0.13 + 0.16 ms on configs[3]).
__global__ __launch_bounds__(256) void k_gather16_T(const int32_t* __restrict__ idx, int64_t N, int64_t Np, int T, int F, int nT, const bf16* __restrict__ Wt,
                                                    const bf16* __restrict__ We, const bf16* __restrict__ Wr, int dt, int de, int dr, bf16* __restrict__ XT,
                                                    int64_t ldy) {
  __shared__ __attribute__((aligned(16))) bf16 tl[64][72];
  const int D = dt + de + dr;
  const int64_t n0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64, t = blockIdx.z;
  // Both of a thread's pieces: ids first, then rows, every load unconditional from a clamped address and masked afterwards (written as
  // `if (in range) load`, hipcc keeps the wait inside the branch: four dependent round trips per thread instead of two).
  const bf16* src[2]; const int32_t* idp[2]; bool ok[2], summed[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int f = threadIdx.x + 256 * e;
    const int rr = f >> 3, col = c0 + (f & 7) * 8;
    const int64_t n = n0 + rr;
    ok[e] = n < N && col < D;
    const int cl = col < D ? col : 0;
    const int32_t* id = idx + ((n < N ? n : N - 1) * T + t) * F;
    const int which = cl < dt ? 0 : (cl < dt + de ? 1 : 2);
    idp[e] = id;
    summed[e] = which == 0 && nT > 1;
    const int row = id[which == 0 ? F - nT - 2 : (which == 1 ? F - 2 : F - 1)] - 1;
    src[e] = which == 0 ? Wt + (int64_t)row * dt + cl : (which == 1 ? We + (int64_t)row * de + (cl - dt) : Wr + (int64_t)row * dr + (cl - dt - de));
  }
  bf16x8 ld[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) ld[e] = *(const bf16x8*)src[e];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int f = threadIdx.x + 256 * e;
    const int rr = f >> 3, col = c0 + (f & 7) * 8;
    bf16x8 v = ld[e];
    if (summed[e] && ok[e]) {   // several type slots per step: their rows summed (FeatureEmbedding.lua:55)
      float acc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = (float)v[q];
      for (int k = 1; k < nT; ++k) {
        const bf16x8 w = *(const bf16x8*)(Wt + (int64_t)(idp[e][F - nT - 2 + k] - 1) * dt + col);
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] += (float)w[q];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = tobf(acc[q]);
    }
    if (!ok[e]) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = (bf16)0.f;
    }
    *(bf16x8*)(&tl[rr][(f & 7) * 8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int f = threadIdx.x + 256 * e;
    const int cc = f >> 3, rq = (f & 7) * 8;
    if (c0 + cc < D && n0 + rq < Np) {
      bf16x8 v;
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = tl[rq + q][cc];
      *(bf16x8*)(XT + (int64_t)(c0 + cc) * ldy + (int64_t)t * Np + n0 + rq) = v;
    }
  }
}
// cell backward of one step on the saved bf16 gate values (kernels_basic.hip k_gates_bwd, with dA written in bf16)
__global__ void k_gates_bwd16(const bf16* __restrict__ act, const float* __restrict__ c, const float* __restrict__ c_prev, const float* __restrict__ dH_up,
                              float* __restrict__ dH, float* __restrict__ dC, bf16* __restrict__ dA, int64_t N, int H) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= N * H) return;
  const int jx = (int)(gid % H);
  const int64_t n = gid / H;
  const bf16* a = act + n * 4 * H;
  const float ig = (float)a[jx], gg = (float)a[H + jx], fg = (float)a[2 * H + jx], og = (float)a[3 * H + jx];
  const float tc = tanhf(c[gid]);
  const float dh = dH[gid] + (dH_up ? dH_up[gid] : 0.f);
  const float dO = dh * tc;
  const float dc = dC[gid] + dh * og * (1.f - tc * tc);
  const float cp = c_prev ? c_prev[gid] : 0.f;
  bf16* d = dA + n * 4 * H;
  d[jx] = tobf(dc * gg * ig * (1.f - ig));
  d[H + jx] = tobf(dc * ig * (1.f - gg * gg));
  d[2 * H + jx] = tobf(dc * cp * fg * (1.f - fg));
  d[3 * H + jx] = tobf(dO * og * (1.f - og));
  dC[gid] = dc * fg;
  dH[gid] = 0.f;
}
// The same cell backward on the persistent layer kernel's FRAGMENT-order saves (lstm_bf16_persist.hip Cell::store): record ((unit NCH + chunk) NW + wave) 64 +
// lane holds, for path 32 unit + (lane & 31) and hidden units HC chunk + 8 wave + 4 (lane >> 5) .. + 3, c (4 bf16: a copy of the fp32 cell state the recurrence itself runs on) and the gates as [i4 g4] / [f4 o4].
// One workgroup = one (unit, chunk): 32 rows x HC hidden units.  The records are read coalesced; everything row-major (dH, dC in / out, dA out) goes
// through LDS tiles so that global accesses are whole 128 / 256-byte row segments.
template <int NW>
__global__ __launch_bounds__(256) void k_gates_bwd16_frag(const bf16x8* __restrict__ A0, const bf16x8* __restrict__ A1, const bf16x4* __restrict__ cF,
                                                          const bf16x4* __restrict__ cprevF, const float* __restrict__ dH_up, const float* __restrict__ dH,
                                                          float* __restrict__ dC, bf16* __restrict__ dA, int64_t N, int H, bf16* __restrict__ dAT /* nullable: this
                                                          step's column block of dA^T [4H][T Np] */, int64_t ldT, int64_t Np, int64_t NU, int units_per_wg,
                                                          float* __restrict__ gbias /* nullable: [4H] += column sums of dA (the bias gradient) */) {
  constexpr int HC = 8 * NW, Q = HC / 4;
  __shared__ __attribute__((aligned(16))) float sH[32][HC + 4];
  __shared__ __attribute__((aligned(16))) float sC[32][HC + 4];
  __shared__ __attribute__((aligned(16))) bf16 sA[32][4][HC + 8];
  const int tid = threadIdx.x;
  const int c = blockIdx.y, NCH = gridDim.y, ub = HC * c;
  // A workgroup walks units_per_wg 32-row units of its chunk: the bias gradient (column sums of dA, formerly a second sweep over dA^T: 1.2 GB
  // per step) is summed per thread -- thread (g, u) owns gate row g H + ub + u -- across the units and leaves as ONE atomic per thread.
  float bsum = 0.f;
  const int bg = tid / HC, bu = tid - bg * HC;   // (tid < 4 HC)
  for (int ui = 0; ui < units_per_wg; ++ui) {
    const int64_t unit = (int64_t)blockIdx.x * units_per_wg + ui;
    if (unit >= NU) break;   // (workgroup-uniform)
    const int64_t row0 = unit * 32;
    if (ui > 0) __syncthreads();   // (the previous unit's tiles have been read)
    for (int i = tid; i < 32 * Q; i += 256) {
      const int r = i / Q, q = i - r * Q;
      const int64_t n = row0 + r;
      f32x4 vh = f32x4{0.f, 0.f, 0.f, 0.f}, vc = vh;
      if (n < N) {
        vh = *(const f32x4*)(dH + n * H + ub + 4 * q);
        vc = *(const f32x4*)(dC + n * H + ub + 4 * q);
        if (dH_up) { const f32x4 u = *(const f32x4*)(dH_up + n * H + ub + 4 * q); vh += u; }
      }
      *(f32x4*)&sH[r][4 * q] = vh;
      *(f32x4*)&sC[r][4 * q] = vc;
    }
    __syncthreads();
    for (int rl = tid; rl < NW * 64; rl += 256) {
      const int w = rl >> 6, lane = rl & 63, ln = lane & 31, ul = 8 * w + 4 * (lane >> 5);
      const int64_t rec = ((unit * NCH + c) * NW + w) * 64 + lane;
      const bf16x8 a0 = A0[rec], a1 = A1[rec];
      const bf16x4 cc16 = cF[rec];
      bf16x4 cp16;
#pragma unroll
      for (int j = 0; j < 4; ++j) cp16[j] = (bf16)0.f;
      if (cprevF) cp16 = cprevF[rec];
      f32x4 cc, cp;
#pragma unroll
      for (int j = 0; j < 4; ++j) { cc[j] = (float)cc16[j]; cp[j] = (float)cp16[j]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ig = (float)a0[j], gg = (float)a0[4 + j], fg = (float)a1[j], og = (float)a1[4 + j];
        const float tc = tanhf(cc[j]);
        const float dh = sH[ln][ul + j];
        const float dO = dh * tc;
        const float dc = sC[ln][ul + j] + dh * og * (1.f - tc * tc);
        sA[ln][0][ul + j] = tobf(dc * gg * ig * (1.f - ig));
        sA[ln][1][ul + j] = tobf(dc * ig * (1.f - gg * gg));
        sA[ln][2][ul + j] = tobf(dc * cp[j] * fg * (1.f - fg));
        sA[ln][3][ul + j] = tobf(dO * og * (1.f - og));
        sC[ln][ul + j] = dc * fg;
      }
    }
    __syncthreads();
    // (dH is not cleared: the recurrent product of this step -- dh_{t-1} = dA_t W_o2g, plain stores -- or the next head backward overwrites it)
    for (int i = tid; i < 32 * Q; i += 256) {
      const int r = i / Q, q = i - r * Q;
      const int64_t n = row0 + r;
      if (n < N) *(f32x4*)(dC + n * H + ub + 4 * q) = *(const f32x4*)&sC[r][4 * q];
    }
    constexpr int P8 = HC / 8;   // 16-byte pieces of a (row, gate) segment
    for (int i = tid; i < 32 * 4 * P8; i += 256) {
      const int p = i % P8, g = (i / P8) & 3, r = i / (4 * P8);
      const int64_t n = row0 + r;
      if (n < N) *(bf16x8*)(dA + n * (int64_t)4 * H + (int64_t)g * H + ub + 8 * p) = *(const bf16x8*)&sA[r][g][8 * p];
    }
    if (dAT) {   // the transposed image the dW products contract over (k-contiguous operands): this tile's 4 HC gate rows x 32 paths, pad columns zero
      for (int i = tid; i < 4 * HC * 4; i += 256) {
        const int r8 = i & 3, u = (i >> 2) % HC, g = i / (4 * HC);
        const int64_t n0 = row0 + 8 * r8;
        if (n0 >= Np) continue;
        bf16x8 v;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (n0 + q < N) ? sA[8 * r8 + q][g][u] : (bf16)0.f;
        *(bf16x8*)(dAT + ((int64_t)g * H + ub + u) * ldT + n0) = v;
      }
    }
    if (gbias && tid < 4 * HC) {   // the bf16-rounded values, as the separate row sum over dA^T read them
      float sacc = 0.f;
#pragma unroll 8
      for (int r = 0; r < 32; ++r) sacc += (row0 + r < N) ? (float)sA[r][bg][bu] : 0.f;
      bsum += sacc;
    }
  }
  if (gbias && tid < 4 * HC && bsum != 0.f) unsafeAtomicAdd(gbias + (int64_t)bg * H + ub + bu, bsum);
}

// y[c][r] = x[r][c] for r < R, 0 for R <= r < Rp (the padded row count: 16-byte rows of y); x bf16 [R][C] (C a multiple of 8), y pitch
// ldy (a multiple of 8); 64 x 64 tiles through LDS, 16-byte global accesses on both sides
__global__ __launch_bounds__(256) void k_transpose16(const bf16* __restrict__ x, bf16* __restrict__ y, int64_t R, int64_t Rp, int64_t Cc, int64_t ldy) {
  __shared__ __attribute__((aligned(16))) bf16 t[64][72];
  const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int f = threadIdx.x + 256 * e;
    const int rr = f >> 3, cq = (f & 7) * 8;
    bf16x8 v;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (bf16)0.f;
    if (r0 + rr < R && c0 + cq < Cc) v = *(const bf16x8*)(x + (r0 + rr) * Cc + c0 + cq);
    *(bf16x8*)(&t[rr][cq]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int f = threadIdx.x + 256 * e;
    const int cc = f >> 3, rq = (f & 7) * 8;
    if (c0 + cc < Cc && r0 + rq < Rp) {
      bf16x8 v;
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = t[rq + q][cc];
      *(bf16x8*)(y + (c0 + cc) * ldy + r0 + rq) = v;
    }
  }
}
// h^T [H][T Np] (the dW_o2g product's k-contiguous operand) from the persistent forward's FRAGMENT-order h records (lstm_bf16_persist.hip Cell::store:
// record (((t NU + unit) NCHF + chunk) 8 + wave) 64 + lane = hidden units 64 chunk + 8 wave + 4 (lane >> 5) .. + 3 of path 32 unit + (lane & 31)).
// One workgroup = (two units = 64 paths, one chunk of 64 hidden units, one step): 8 KB of records in, coalesced; transposed through LDS by two-byte
// writes; 128-byte rows out.  Paths past N are written as zeros (their records hold a repeated row).
__global__ __launch_bounds__(256) void k_hfrag_T(const bf16x4* __restrict__ HsF, bf16* __restrict__ HT, int64_t NU, int64_t step_recs, int64_t N, int64_t Np, int64_t ldT, int H) {
  constexpr int TP = 64 + 8;   // elements: 144-byte rows (the two lane halves of a record wave write rows 4 apart: other banks)
  __shared__ __attribute__((aligned(16))) bf16 tl[64][TP];
  const int tid = threadIdx.x, cf = blockIdx.y, t = blockIdx.z, NCHF = H / 64;
  const int64_t u0 = (int64_t)blockIdx.x * 2, n0 = u0 * 32;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int r = tid + 256 * e;              // record of the pair of units: unit r >> 9, wave (r >> 6) & 7, lane r & 63
    const int uu = r >> 9, wf = (r >> 6) & 7, lane = r & 63, ln = lane & 31, half = lane >> 5;
    bf16x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (bf16)0.f;
    if (u0 + uu < NU && n0 + 32 * uu + ln < N) v = HsF[(int64_t)t * step_recs + (((u0 + uu) * NCHF + cf) * 8 + wf) * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) tl[8 * wf + 4 * half + j][32 * uu + ln] = v[j];
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int pc = tid + 256 * e, hl = pc >> 3, oct = pc & 7;
    if (n0 + 8 * oct < Np) *(bf16x8*)(HT + (int64_t)(64 * cf + hl) * ldT + (int64_t)t * Np + n0 + 8 * oct) = *(const bf16x8*)&tl[hl][8 * oct];
  }
}
// out[r] += sum of row r of x (bf16 [R][n], n a multiple of 8): the bias gradient from dA^T
__global__ __launch_bounds__(256) void k_rowsum16(const bf16* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[256];
  const bf16* row = x + (int64_t)blockIdx.x * n;
  float acc = 0.f;
  for (int64_t i = (int64_t)threadIdx.x * 8; i < n; i += 256 * 8) {
    const bf16x8 v = *(const bf16x8*)(row + i);
#pragma unroll
    for (int q = 0; q < 8; ++q) acc += (float)v[q];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int sft = 128; sft > 0; sft >>= 1) {
    if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] += red[0];
}

// ---- small tables: their gradients through ONE extra column block of the dW product (round 5) -------------------------------------------------
// The step input is x = [Wt[type] | We[entity] | Wr[relation]] (FeatureEmbedding.lua:112-121), so with the one-hot selectors S_t, S_r of a batch
//   x_r = S_r Wr                          =>   dW_i2g[:, relation columns] = dA^T x_r = (dA^T S_r) Wr        = G_r Wr
//   dWr = S_r^T dx_r = S_r^T (dA W_i2g_r)  =>   dWr                         = (S_r^T dA) W_i2g_r              = G_r^T W_i2g[:, relation columns]
// and the same for the type table: the relation / type thirds of the dx product and of the dW_i2g product, and both table-gradient launches, collapse
// into G = dA^T [S_r | S_t] -- 128 more columns of the split-K dW product (Vr + Vt <= 128 one-hot rows, exact in bf16) -- plus two tiny fp32 products.
// configs[3] (100 relations, 6 types, d = 128): dx shrinks from N = 384 to the entity slice (N = 128: a third of the flops, one column tile, dA^T
// read once instead of three times), the two dW products become ONE over the operand Z^T = [x_e^T | S^T | h_{t-1}^T] (N = 640 instead of 768, dA^T read
// by one launch instead of two), the one-hot table-gradient launch (0.27 ms, 400 MB of dx) and two thirds of the X^T gather disappear.
// G is a sum of bf16 dA values in fp32 and the small products run in fp32 on the same bf16 shadows the big ones read: the same gradient, fewer roundings.
//
// S^T rows of the operand: row r < Vr is relation r + 1, row Vr + y is type y + 1 (num_types = 1); [128][T][Np], pad columns zero.
__global__ __launch_bounds__(256) void k_onehot_T(const int32_t* __restrict__ idx, int64_t N, int64_t Np, int T, int F, int Vr, int Vt, bf16* __restrict__ ST, int64_t ldT) {
  __shared__ int rid[64], tyid[64];
  const int64_t n0 = (int64_t)blockIdx.x * 64;
  const int t = blockIdx.y;
  if (threadIdx.x < 64) {
    const int64_t n = n0 + threadIdx.x;
    const int32_t* id = idx + ((n < N ? n : N - 1) * T + t) * F;
    const int r = id[F - 1] - 1, y = id[F - 3] - 1;
    rid[threadIdx.x] = n < N ? r : -1;
    tyid[threadIdx.x] = n < N ? Vr + y : -1;
  }
  __syncthreads();
  const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
  typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p0 = 32 * half + 8 * q;
    if (n0 + p0 >= Np) continue;
    u16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (rid[p0 + e] == row || tyid[p0 + e] == row) ? (unsigned short)0x3f80 : (unsigned short)0;   // bf16 1.0
    *(u16x8*)(ST + (int64_t)row * ldT + (int64_t)t * Np + n0 + p0) = v;
  }
}
// What the merged product leaves in Ct [4H][NZ] (NZ = de + 128 + H: entity columns | G | recurrent columns) goes where the optimiser reads it (all +=):
// workgroups [0, 4H): one gate row each -- gW_i2g[k][:] = [G_t Wt | Ct entity block | G_r Wr], gW_o2g[k][:]; workgroups behind them: one table row
// each -- gWr[r][:] += sum_k G[k][r] W_i2g[k][relation columns] (no atomics: a fixed summation order), likewise gWt.
struct FinArgs {
  const float* Ct; int NZ, G4, Din, H, dt, de, dr, Vt, Vr;
  const bf16* Wt16; const bf16* Wr16; const bf16* Wi16;
  float* gWi; float* gWo; float* gWt; float* gWr;
};
__global__ __launch_bounds__(256) void k_small_tables_finish(FinArgs a) {
  __shared__ float g[128];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < a.G4) {
    const int k = blockIdx.x;
    const float* row = a.Ct + (int64_t)k * a.NZ;
    if (tid < 128) g[tid] = row[a.de + tid];
    __syncthreads();
    for (int j = tid; j < a.Din; j += 256) {
      float v = 0.f;
      if (j < a.dt) {
        for (int y = 0; y < a.Vt; ++y) v += g[a.Vr + y] * (float)a.Wt16[y * a.dt + j];
      } else if (j < a.dt + a.de) {
        v = row[j - a.dt];
      } else {
        const int jj = j - a.dt - a.de;
#pragma unroll 4
        for (int r = 0; r < a.Vr; ++r) v += g[r] * (float)a.Wr16[r * a.dr + jj];
      }
      a.gWi[(int64_t)k * a.Din + j] += v;
    }
    for (int j = tid; j < a.H; j += 256) a.gWo[(int64_t)k * a.H + j] += row[a.de + 128 + j];
    return;
  }
  // one workgroup per one-hot row r (relation r, or type r - Vr), no atomics: two interleaved K sub-sums per column added in a fixed order
  __shared__ float red[256];
  const int r = (int)blockIdx.x - a.G4;
  const bool rel = r < a.Vr;
  const int w = rel ? a.dr : a.dt, col0 = rel ? a.dt + a.de : 0;
  const int j = tid & 127, sb = tid >> 7;
  const int jc = j < w ? j : w - 1;   // (w <= 128; idle lanes re-read the last column: loads stay unconditional)
  float acc = 0.f;
#pragma unroll 8
  for (int k = sb; k < a.G4; k += 2) acc += a.Ct[(int64_t)k * a.NZ + a.de + r] * (float)a.Wi16[(int64_t)k * a.Din + col0 + jc];
  red[tid] = acc;
  __syncthreads();
  if (sb == 0 && j < w) (rel ? a.gWr + (int64_t)r * a.dr : a.gWt + (int64_t)(r - a.Vr) * a.dt)[j] += red[tid] + red[tid + 128];
}

// ---- state + orchestration --------------------------------------------------------------------------------------------------
// lstm_bf16_persist.hip: fragment-order saves of the persistent layer kernel's training launch
struct PersistSaves { const bf16* CsF; const bf16* ActF0; const bf16* ActF1; const bf16* HsF; int64_t NU, step_recs; int NW; };

struct State {
  bf16* We16 = nullptr; bool we_all_dirty = true;
  bf16* dense16 = nullptr;      // bf16 image of the dense arena (same offsets)
  bf16* WT16 = nullptr;         // per layer: W_i2g^T [Din][4H] | W_o2g^T [H][4H]
  bool dense_dirty = true;
  int64_t cap_N = 0; int cap_T = 0;
  bf16 *X16 = nullptr, *XT16 = nullptr, *H16 = nullptr, *HT16 = nullptr, *ACT16 = nullptr, *dA16 = nullptr, *dAT16 = nullptr;
  bf16* ZT16 = nullptr; int64_t z_cap = 0;   // small-table backward: the merged dW product's operand [x_e^T | S^T | h_{t-1}^T], [de + 128 + H][T][Np]
  float* Ctmp = nullptr; int64_t ct_cap = 0; //   ... and its result [4H][de + 128 + H]
  void* persist = nullptr;      // lstm_bf16_persist.hip: packed weights + scratch slabs of the persistent layer kernel
  bool pack_dirty = true;       // its packed weights are stale
  void* persist_bwd = nullptr;  // lstm_bf16_bwd_persist.hip: packed W_o2g^T fragments of the persistent BPTT kernel
  bool packb_dirty = true;
  bool bias_in_gates = false;   // k_gates_bwd16_frag also summed the bias gradient (measurement switch)
  bool act_frag = false;        // the last training forward wrote c and the gate planes in fragment order (persistent kernel; k_gates_bwd16_frag)
  PersistSaves sv{};
  // the backward's side stream (backward(): the memory-bound helpers run beside the matrix-core launches they do not depend on)
  int side_probes = 0;   // candidates make_concurrent_stream tried for it
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_operands = nullptr, ev_dx = nullptr, ev_join = nullptr;
};
// lstm_bf16_persist.hip: the whole layer (gather + T steps) as one persistent launch
bool persist_shape_ok(const kprn_handle* h, const kprn_batch* b);
void persist_forward(kprn_handle* h, const kprn_batch* b, bool save, void*& st, bool repack, const bf16* Wt16, const bf16* We16, const bf16* Wr16, bf16* H16,
                     PersistSaves* sv);
void persist_release(void*& st);
// lstm_bf16_bwd_persist.hip: BPTT through the layer (cell backward + recurrent product of all T steps) as one persistent launch
bool persist_bwd_shape_ok(const kprn_handle* h, const PersistSaves& sv, int64_t N, int T);
void persist_backward(kprn_handle* h, int64_t N, int T, int cid, const PersistSaves& sv, void*& st, bool repack, bf16* dA16 /* nullable: no row-major copy */, bf16* dAT16, int64_t Np,
                      float* dXe /* nullable: the launch also forms the entity slice of dx, [T][N][de] */, int64_t ldT /* row pitch of dA^T, T Np .. T Np + 512 */);
bool persist_bwd_dxe_ok(const kprn_handle* h);
void persist_bwd_release(void*& st);
static State* st(kprn_handle* h) {
  if (!h->bf16_state) h->bf16_state = new State();
  return (State*)h->bf16_state;
}
template <typename Tp> static Tp* dal(int64_t n) {
  void* p = nullptr;
  hipError_t e = kprn_dev_malloc(&p, (size_t)std::max<int64_t>(n, 1) * sizeof(Tp));
  if (e != hipSuccess) throw KprnError{KPRN_E_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e)};
  return (Tp*)p;
}
static int64_t wt_off(const kprn_handle* h, int l) {   // offset of layer l's transposed pair inside WT16
  int64_t o = 0;
  for (int k = 0; k < l; ++k) o += (int64_t)4 * h->cfg.H * (h->layer[k].Din + h->cfg.H);
  return o;
}

bool supported(const kprn_handle* h, const kprn_batch* b) {
  const kprn_config& c = h->cfg;
  const int64_t N = (int64_t)b->B * b->P;
  return c.compute_dtype == 1 && c.rnn_type == 0 && h->impl == 0 && N >= 256 && c.dt > 0 && c.de > 0 && (c.dt % 8) == 0 && (c.de % 8) == 0 && (c.dr % 8) == 0 && (c.H % 8) == 0;
}

void params_changed(kprn_handle* h, bool entity_rows_only) {
  if (!h->bf16_state) return;
  State* s = (State*)h->bf16_state;
  s->dense_dirty = true;
  s->pack_dirty = true;
  s->packb_dirty = true;
  if (!entity_rows_only) s->we_all_dirty = true;
}

// the optimiser / the catch-up replay rewrote these rows of entity_emb: refresh their shadow
void rows_updated(kprn_handle* h, const int32_t* rows, const int32_t* count, int64_t max_rows) {
  if (!h->bf16_state) return;
  State* s = (State*)h->bf16_state;
  if (s->we_all_dirty || !s->We16 || max_rows <= 0) return;   // (a full conversion is pending anyway)
  const int de = h->cfg.de;
  const int64_t work = max_rows * (de >> 2);
  hipLaunchKernelGGL(k_rows_cvt, dim3((unsigned)std::min<int64_t>((work + 255) / 256, 4096)), dim3(256), 0, h->stream, h->We, s->We16, rows, count, de);
  HIP_TRY(hipGetLastError());
}

void release(kprn_handle* h) {
  State* s = (State*)h->bf16_state;
  if (!s) return;
  for (bf16* p : {s->We16, s->dense16, s->WT16, s->X16, s->XT16, s->H16, s->HT16, s->ACT16, s->dA16, s->dAT16, s->ZT16}) if (p) hipFree(p);
  if (s->Ctmp) hipFree(s->Ctmp);
  persist_release(s->persist);
  persist_bwd_release(s->persist_bwd);
  if (s->side) {
    hipStreamSynchronize(s->side); hipStreamDestroy(s->side);
    for (hipEvent_t e : {s->ev_fork, s->ev_operands, s->ev_dx, s->ev_join}) if (e) hipEventDestroy(e);
  }
  delete s;
  h->bf16_state = nullptr;
}

static void refresh_shadows(kprn_handle* h) {
  State* s = st(h);
  const kprn_config& c = h->cfg;
  hipStream_t strm = h->stream;
  if (!s->We16) { s->We16 = dal<bf16>(h->n_ent); s->we_all_dirty = true; }
  if (!s->dense16) { s->dense16 = dal<bf16>(h->n_dense); s->WT16 = dal<bf16>(wt_off(h, c.L)); s->dense_dirty = true; }
  if (s->we_all_dirty) {
    ProfScope ps(h, "bf16_shadow_table");
    hipLaunchKernelGGL(k_cvt, dim3((unsigned)((h->n_ent / 4 + 256) / 256)), dim3(256), 0, strm, h->We, s->We16, h->n_ent);
    s->we_all_dirty = false;
  }
  if (s->dense_dirty) {
    ProfScope ps(h, "bf16_shadow_weights");
    hipLaunchKernelGGL(k_cvt, dim3((unsigned)((h->n_dense / 4 + 256) / 256)), dim3(256), 0, strm, h->dense, s->dense16, h->n_dense);
    for (int l = 0; l < c.L; ++l) {
      const int Din = h->layer[l].Din, G4 = 4 * c.H;
      bf16* wt = s->WT16 + wt_off(h, l);
      hipLaunchKernelGGL(k_cvt_T, dim3((unsigned)(((int64_t)G4 * Din + 255) / 256)), dim3(256), 0, strm, h->dense + h->layer[l].Wi, wt, G4, Din);
      hipLaunchKernelGGL(k_cvt_T, dim3((unsigned)(((int64_t)G4 * c.H + 255) / 256)), dim3(256), 0, strm, h->dense + h->layer[l].Wo, wt + (int64_t)G4 * Din, G4, c.H);
    }
    s->dense_dirty = false;
  }
  HIP_TRY(hipGetLastError());
}

static void ensure_buffers(kprn_handle* h, int64_t N, int T) {
  State* s = st(h);
  if (N <= s->cap_N && T <= s->cap_T) return;
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (bf16** p : {&s->X16, &s->XT16, &s->H16, &s->HT16, &s->ACT16, &s->dA16, &s->dAT16}) if (*p) { hipFree(*p); *p = nullptr; }
  const kprn_config& c = h->cfg;
  const int64_t cn = std::max(N, s->cap_N);
  const int ct = std::max(T, s->cap_T);
  const int64_t rows = cn * ct, rows_p = ((cn + 7) & ~(int64_t)7) * ct;
  const int Dm = std::max(h->D, c.H);
  s->X16 = dal<bf16>(rows * h->D); s->XT16 = dal<bf16>(rows_p * Dm);
  s->H16 = dal<bf16>((int64_t)c.L * rows * c.H); s->HT16 = dal<bf16>(rows_p * c.H);
  s->ACT16 = dal<bf16>((int64_t)c.L * rows * 4 * c.H);
  s->dA16 = dal<bf16>(rows * 4 * c.H); s->dAT16 = dal<bf16>((rows_p + 512) * 4 * c.H);   // (+ 512: room for a padded row pitch, t_pitch)
  s->cap_N = cn; s->cap_T = ct;
}

// nn.Linear(H, C) on the last step's h for the bf16 pipeline (OneModel.lua:275): S[n][c] = sum_k bf16(h_T[n][k]) bf16(W_out[c][k]) + b[c], fp32 accumulation --
// the arithmetic the generic product did for this call (fp32 operands rounded to bf16 as fragments are formed), as its own launch: that product
// spent 0.082 ms per call on configs[3] (twice per step: 100 MB of h_T through 64 x 64 tiles of 4-byte loads); this is one pass over h_T.
// A wave owns 16 paths x all classes (NT column tiles of v_mfma_f32_16x16x32_bf16); a lane reads 64 contiguous bytes of its path's row per 64-k block
// -- the MFMA's k index is permuted to make that so (k = 64 j + 16 kg + 8 i + q), and the weights sit in LDS in the same permuted fragment order.
// The launch is a handful of round trips (weight fragments, then per block of 16 paths the row and the stores), so the fragment fill and a block's
// row are each requested in one go (first build: 9 + 6 dependent round trips, 0.039 ms).
template <int NT, int NJ>   // NT: 16-class column tiles; NJ = H / 64
__global__ __launch_bounds__(256) void k_head_fwd16(const float* __restrict__ hT, const bf16* __restrict__ W16, const float* __restrict__ bias,
                                                       float* __restrict__ S, int64_t N, int C) {
  constexpr int H = 64 * NJ, E = NJ * 2 * NT * 64, PE = (E + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) bf16x8 wf[];   // [NJ][2][NT][64 lanes]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  {
    // the weight fragments: all of a thread's pieces requested together, from clamped addresses; rows past C are zeroed by a mask the optimiser
    // cannot see through (a select here would put each load and its wait inside a branch: PE dependent round trips before the first row is read)
    u32x4 v[PE];
#pragma unroll
    for (int i = 0; i < PE; ++i) {
      const int e0 = threadIdx.x + 256 * i, e = e0 < E ? e0 : E - 1;
      const int l = e & 63, nt = (e >> 6) % NT, ji = (e >> 6) / NT;
      const int cls = nt * 16 + (l & 15);
      v[i] = *(const u32x4*)(W16 + (int64_t)(cls < C ? cls : C - 1) * H + 64 * (ji >> 1) + 16 * (l >> 4) + 8 * (ji & 1));
    }
#pragma unroll
    for (int i = 0; i < PE; ++i) {
      const int e = threadIdx.x + 256 * i;
      unsigned m = ((e >> 6) % NT) * 16 + (e & 15) < C ? 0xffffffffu : 0u;
      asm("" : "+v"(m));
      if (E % 256 == 0 || e < E) *(u32x4*)(wf + e) = v[i] & m;
    }
  }
  float bj[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { const int col = nt * 16 + (lane & 15); bj[nt] = bias[col < C ? col : C - 1]; }
  __syncthreads();
  for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk * 16 < N; blk += (int64_t)gridDim.x * 4) {
    const int64_t row = blk * 16 + (lane & 15);
    const float* src = hT + (row < N ? row : N - 1) * H + 16 * (lane >> 4);
    asm volatile("" ::: "memory");   // (keeps the weight fragments in LDS: hoisted out of this loop they take 48 NT registers and the row loads get serialized)
    f32x4 x[NJ][4];   // this lane's share of its path's row, requested at once: one round trip per 16 paths
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) x[j][q] = *(const f32x4*)(src + 64 * j + 4 * q);
    __builtin_amdgcn_sched_barrier(0);   // all 4 NJ requests are issued before the first is used (left alone, hipcc sinks each load down to its MFMA)
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      bf16x8 a0, a1;
#pragma unroll
      for (int q = 0; q < 4; ++q) { a0[q] = tobf(x[j][0][q]); a0[4 + q] = tobf(x[j][1][q]); a1[q] = tobf(x[j][2][q]); a1[4 + q] = tobf(x[j][3][q]); }
      const bf16x8* w0 = wf + (size_t)(j * 2) * NT * 64 + lane;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, w0[nt * 64], acc[nt], 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, w0[(NT + nt) * 64], acc[nt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = nt * 16 + (lane & 15);
      if (col >= C) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t ro = blk * 16 + 4 * (lane >> 4) + r;   // D: lane (col, rg) register r <-> row 4 rg + r
        if (ro < N) S[ro * C + col] = acc[nt][r] + bj[nt];
      }
    }
  }
}
// false: a shape this launch does not take (the caller falls back to the generic product).  KPRN_BF16_HEAD=0: always false (A/B)
static bool head_fwd16(hipStream_t s, const float* hT, const bf16* W16, const float* bias, float* S, int64_t N, int H, int C) {
  static const bool off = KPRN_DEV_ENV("KPRN_BF16_HEAD") && KPRN_DEV_ENV("KPRN_BF16_HEAD")[0] == '0';
  const int NT = (C + 15) / 16, NJ = H >> 6;
  const size_t lds = (size_t)NJ * 2 * NT * 64 * sizeof(bf16x8);
  if (off || N <= 0 || (H & 63) || C < 1 || NT > 4 || !(NJ == 2 || NJ == 3 || NJ == 4 || NJ == 6) || ((uintptr_t)W16 & 15) || ((uintptr_t)hT & 15)) return false;
  const dim3 grid((unsigned)std::min<int64_t>((N + 63) / 64, 2 * 256));   // two workgroups per CU (<= 256 registers per lane), all resident: a wave walks its blocks
#define KPRN_HF(NTv, NJv) hipLaunchKernelGGL((k_head_fwd16<NTv, NJv>), grid, dim3(256), lds, s, hT, W16, bias, S, N, C)
#define KPRN_HF_NJ(NTv) do { if (NJ == 2) KPRN_HF(NTv, 2); else if (NJ == 3) KPRN_HF(NTv, 3); else if (NJ == 4) KPRN_HF(NTv, 4); else KPRN_HF(NTv, 6); } while (0)
  if (NT == 1) KPRN_HF_NJ(1); else if (NT == 2) KPRN_HF_NJ(2); else if (NT == 3) KPRN_HF_NJ(3); else KPRN_HF_NJ(4);
#undef KPRN_HF_NJ
#undef KPRN_HF
  HIP_TRY(hipGetLastError());
  return true;
}

// forward of the whole stack; the head runs on the fp32 h_T of the top layer (written by its last step) through the generic bf16-product GEMM
void forward(kprn_handle* h, const kprn_batch* b, bool save) {
  const kprn_config& c = h->cfg;
  State* s = st(h);
  Workspace& w = h->ws;
  hipStream_t strm = h->stream;
  const int H = c.H, L = c.L, T = b->T, D = h->D;
  const int64_t N = (int64_t)b->B * b->P;
  refresh_shadows(h);
  ensure_buffers(h, N, T);
  const bool persist = persist_shape_ok(h, b);
  if (!persist) {   // (the persistent kernel gathers for itself, and the backward then builds the dW product's operand X^T from the tables: k_gather16_T)
    ProfScope ps(h, "embed_gather_bf16");
    const int64_t work = N * T * (D >> 3);
    hipLaunchKernelGGL(k_gather16, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, strm, b->idx, N, T, b->F, c.num_types, s->dense16 + h->off_Wt, s->We16,
                       s->dense16 + h->off_Wr, c.dt, c.de, c.dr, s->X16);
    HIP_TRY(hipGetLastError());
  }
  if (persist) {
    persist_forward(h, b, save, s->persist, s->pack_dirty, s->dense16 + h->off_Wt, s->We16, s->dense16 + h->off_Wr, /*row-major h plane: not needed, h^T is built from the fragment-order records*/ nullptr, &s->sv);
    s->pack_dirty = false;
    if (save) s->act_frag = true;
  }
  if (!persist && save) s->act_frag = false;
  for (int l = 0; l < L && !persist; ++l) {
    const int Din = h->layer[l].Din;
    const bf16* in = (l == 0) ? s->X16 : s->H16 + (int64_t)(l - 1) * T * N * H;
    bf16* hs = s->H16 + (int64_t)l * T * N * H;
    float* cs = w.Cs + (int64_t)l * T * N * H;
    bf16* act = s->ACT16 + (int64_t)l * T * N * 4 * H;
    ProfScope ps(h, "lstm_step_bf16");
    ps.launches = T;
    for (int t = 0; t < T; ++t) {
      GArgs a;
      memset(&a, 0, sizeof(a));
      a.A = in + (int64_t)t * N * Din; a.lda = Din; a.B = s->dense16 + h->layer[l].Wi; a.ldb = Din; a.K = Din;
      if (t > 0) { a.A2 = hs + (int64_t)(t - 1) * N * H; a.lda2 = H; a.B2 = s->dense16 + h->layer[l].Wo; a.ldb2 = H; a.K2 = H; }
      a.M = N; a.N = 4 * H; a.H = H; a.bias = h->dense + h->layer[l].bi;
      a.cprev = t > 0 ? cs + (int64_t)(t - 1) * N * H : nullptr; a.cout = cs + (int64_t)t * N * H;
      a.hout = hs + (int64_t)t * N * H; a.ldh = H; a.act = save ? act + (int64_t)t * N * 4 * H : nullptr;
      a.hout_f32 = (l == L - 1 && t == T - 1) ? w.Hs + ((int64_t)(L - 1) * T + (T - 1)) * N * H : nullptr;
      a.kchunk = ((Din + TK - 1) / TK) * TK;
      if (big_tiles(N, 4 * (int64_t)H)) {
        a.mtiles = (N + 255) / 256; a.ntiles = (H + 63) / 64;
        launch16<EPI_LSTM, 1>(strm, a, 1);
      } else {
        a.mtiles = (N + 127) / 128; a.ntiles = (H + 31) / 32;
        launch16<EPI_LSTM, 0>(strm, a, 1);
      }
    }
  }
  {
    ProfScope ps(h, "gemm_head_fwd");
    const float* hT = w.Hs + ((int64_t)(L - 1) * T + (T - 1)) * N * H;
    if (!head_fwd16(strm, hT, s->dense16 + h->off_outW, h->dense + h->off_outb, w.S, N, H, c.C))
      gemm::run(strm, hT, H, 1, h->dense + h->off_outW, 1, H, w.S, c.C, N, c.C, H, false, h->dense + h->off_outb, 1, true);
  }
}

// time-major x [T][N][C] -> y [C][T][Np] (Np = N rounded up to 8: every step's block of a row starts on 16 bytes, pad columns zero)
static void transpose_steps(hipStream_t s, const bf16* x, bf16* y, int T, int64_t N, int64_t Np, int64_t Cc) {
  for (int t = 0; t < T; ++t)
    hipLaunchKernelGGL(k_transpose16, dim3((unsigned)((Np + 63) / 64), (unsigned)((Cc + 63) / 64)), dim3(256), 0, s, x + (int64_t)t * N * Cc, y + (int64_t)t * Np, N, Np,
                       Cc, (int64_t)T * Np);
  HIP_TRY(hipGetLastError());
}

// needs forward(save = true) of the same batch; ws.dS holds d loss / d S[:, cid]
void backward(kprn_handle* h, const kprn_batch* b, int cid) {
  const kprn_config& c = h->cfg;
  State* s = st(h);
  Workspace& w = h->ws;
  hipStream_t strm = h->stream;
  const int H = c.H, L = c.L, T = b->T, D = h->D, G4 = 4 * c.H;
  const int64_t N = (int64_t)b->B * b->P, TN = (int64_t)T * N;
  float* gd = h->g_dense;
  // the persistent BPTT launch forms dh_T = dS W_out[cid] itself and keeps dh / dc on the chip: no dH plane, no dC plane
  const bool bptt_persist = s->act_frag && L == 1 && persist_bwd_shape_ok(h, s->sv, N, T);
  // Side stream (with the persistent BPTT launch): what is left of this backward is a chain of matrix-core launches -- BPTT, the dx product,
  // the two dW products -- and memory-bound helpers that each depend on only part of it.  The helpers run on a second stream beside the
  // launches they do not depend on: the head's backward and the two transposed operands of the dW products (in^T gathered from the tables,
  // h^T from the forward's records) beside BPTT; the three table gradients (they need dx only) beside the dW products.  The main stream
  // waits for the side stream before this function returns: nothing outside sees two streams.  KPRN_BF16_BWD_OVERLAP=0: one stream.
  static const bool overlap_env = [] { const char* e = getenv("KPRN_BF16_BWD_OVERLAP"); return !(e && e[0] == '0'); }();
  const bool overlap = bptt_persist && overlap_env;
  if (overlap && !s->side) {
    s->side = make_concurrent_stream(h, &s->side_probes);
    for (hipEvent_t* e : {&s->ev_fork, &s->ev_operands, &s->ev_dx, &s->ev_join}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  hipStream_t side = overlap ? s->side : strm;
  // an error thrown between the fork and the join below must not leave helpers running behind the caller's back: drain the side stream on the way out
  struct SideGuard {
    hipStream_t st; bool joined;
    ~SideGuard() { if (st && !joined) (void)hipStreamSynchronize(st); }
  } side_guard{overlap ? side : nullptr, false};
  if (overlap) {
    HIP_TRY(hipEventRecord(s->ev_fork, strm));
    HIP_TRY(hipStreamWaitEvent(side, s->ev_fork, 0));
  }
  {
    ProfScope ps(h, "head_bwd", side);
    const float* hT = w.Hs + ((int64_t)(L - 1) * T + (T - 1)) * N * H;
    kk::head_bwd(side, w.dS, hT, h->dense + h->off_outW, N, H, cid, bptt_persist ? nullptr : w.dH, gd + h->off_outW, gd + h->off_outb);
  }
  if (!bptt_persist) HIP_TRY(hipMemsetAsync(w.dC, 0, (size_t)N * H * sizeof(float), strm));
  for (int l = L - 1; l >= 0; --l) {
    const int Din = h->layer[l].Din;
    const bf16* act = s->ACT16 + (int64_t)l * TN * G4;
    const bf16* hs = s->H16 + (int64_t)l * TN * H;
    const float* cs = w.Cs + (int64_t)l * TN * H;
    const bf16* wt = s->WT16 + wt_off(h, l);          // W_i2g^T [Din][4H] | W_o2g^T [H][4H]
    const bool has_up = (l < L - 1);
    if (has_up) {
      HIP_TRY(hipMemsetAsync(w.dH, 0, (size_t)N * H * sizeof(float), strm));
      HIP_TRY(hipMemsetAsync(w.dC, 0, (size_t)N * H * sizeof(float), strm));
    }
    // dx straight from the transposed dA the BPTT launch writes for the dW products (gx::k_gemm16xt): that launch then writes no row-major copy
    const int64_t Np_ = (N + 7) & ~(int64_t)7;
    const bool dx_t = bptt_persist && dx_from_transposed_ok((int64_t)T * Np_, Din, G4, Np_);
    // Small tables (see k_onehot_T): dx for the entity slice only, ONE dW product over [x_e^T | S^T | h_{t-1}^T], the type / relation gradients from G.
    const bool tabs = bptt_persist && dx_t && l == 0 && h->bf16_small_tables && c.num_types == 1 && c.Vt + c.Vr <= 128 && c.dt <= 128 && c.dr <= 128 &&
                      c.de >= 128 && T > 1 && b->key_sorted != nullptr && !b->tile_k && s->sv.HsF && (H % 64) == 0;
    // ... and that slice of dx formed inside the BPTT launch itself (a fourth result tile per wave): no dx product launch, dA^T read once less
    const bool dxe_in_bptt = tabs && h->bf16_bptt_dxe != 0 && persist_bwd_dxe_ok(h);
    // Row pitch of the transposed images of the small-table route: T Np elements is 3 x 2^18 bytes at the bench's size -- the 1 536 rows of dA^T (and the 640 of the
    // merged product's other operand) that a K range touches then start at addresses 786 432 bytes apart, i.e. on the same few HBM channels.  A pad of g_t_pad
    // elements (kprn_set_option "bf16_t_pad", default 64 = one 128-byte line) walks consecutive rows over consecutive channels.
    const int64_t ldz = (int64_t)T * Np_ + (tabs ? g_t_pad : 0);
    if (bptt_persist) {
      persist_backward(h, N, T, cid, s->sv, s->persist_bwd, s->packb_dirty, dx_t ? nullptr : s->dA16, s->dAT16, Np_, dxe_in_bptt ? w.dIn : nullptr, ldz);
      s->packb_dirty = false;
      s->bias_in_gates = true;   // (the launch sums the bias gradient from the dA^T pieces it writes)
    }
    for (int t = T - 1; t >= 0 && !bptt_persist; --t) {
      bf16* dA_t = s->dA16 + (int64_t)t * N * G4;
      {
        ProfScope ps(h, "lstm_gates_bwd_bf16");
        if (s->act_frag && l == 0) {   // the persistent layer kernel's fragment-order saves
          const PersistSaves& v = s->sv;
          const int NCH = H / (8 * v.NW);
          const bf16x8* a0 = (const bf16x8*)v.ActF0 + (int64_t)t * v.step_recs;
          const bf16x8* a1 = (const bf16x8*)v.ActF1 + (int64_t)t * v.step_recs;
          const bf16x4* cf = (const bf16x4*)v.CsF + (int64_t)t * v.step_recs;
          const bf16x4* cpf = t > 0 ? (const bf16x4*)v.CsF + (int64_t)(t - 1) * v.step_recs : nullptr;
          const float* up = has_up ? w.dIn + (int64_t)t * N * H : nullptr;
          const int64_t Np_ = (N + 7) & ~(int64_t)7;
          bf16* dat = s->dAT16 + (int64_t)t * Np_;   // (this kernel has the tile in LDS: it writes the transposed image too, no transpose pass for dA)
          // Measured (configs[3], ms per launch): one unit per workgroup, bias sums left to k_rowsum16 0.224; the bias gradient summed here
          // 0.26-0.29 (units per workgroup 4 / 2 / 1 / 8) -- 6 x 0.04-0.06 costs what the separate sweep over dA^T costs (0.20 ms), so it stays
          // a separate launch; KPRN_GATES_FUSED_BIAS=<units per workgroup> switches the fused form on.
          static const int fb_env = KPRN_DEV_ENV("KPRN_GATES_FUSED_BIAS") ? atoi(KPRN_DEV_ENV("KPRN_GATES_FUSED_BIAS")) : 0;
          s->bias_in_gates = fb_env > 0;
          const int upw = fb_env > 0 ? fb_env : 1;
          const dim3 grid((unsigned)((v.NU + upw - 1) / upw), (unsigned)NCH);
          float* gb = fb_env > 0 ? gd + h->layer[l].bi : nullptr;
          if (v.NW == 8) hipLaunchKernelGGL((k_gates_bwd16_frag<8>), grid, dim3(256), 0, strm, a0, a1, cf, cpf, up, w.dH, w.dC, dA_t, N, H, dat, (int64_t)T * Np_, Np_, (int64_t)v.NU, upw, gb);
          else hipLaunchKernelGGL((k_gates_bwd16_frag<4>), grid, dim3(256), 0, strm, a0, a1, cf, cpf, up, w.dH, w.dC, dA_t, N, H, dat, (int64_t)T * Np_, Np_, (int64_t)v.NU, upw, gb);
        } else {
          hipLaunchKernelGGL(k_gates_bwd16, dim3((unsigned)((N * H + 255) / 256)), dim3(256), 0, strm, act + (int64_t)t * N * G4, cs + (int64_t)t * N * H,
                             t > 0 ? cs + (int64_t)(t - 1) * N * H : nullptr, has_up ? w.dIn + (int64_t)t * N * H : nullptr, w.dH, w.dC, dA_t, N, H);
        }
        HIP_TRY(hipGetLastError());
      }
      if (t > 0) {
        ProfScope ps(h, "gemm_o2g_bwd_dh");   // dh_{t-1} = dA_t W_o2g
        gemm16(strm, dA_t, G4, wt + (int64_t)G4 * Din, G4, w.dH, H, N, H, G4, false, nullptr, 1);
      }
    }
    const int64_t Np = (N + 7) & ~(int64_t)7, TNp = (int64_t)T * Np;   // padded step blocks of the transposed images (pads are zero)
    if (tabs) {
      const int NZ = c.de + 128 + H;
      if (ldz * NZ > s->z_cap) {
        HIP_TRY(hipStreamSynchronize(strm));
        if (overlap) HIP_TRY(hipStreamSynchronize(side));
        if (s->ZT16) hipFree(s->ZT16);
        s->ZT16 = nullptr; s->ZT16 = dal<bf16>(ldz * NZ); s->z_cap = ldz * NZ;
      }
      if ((int64_t)G4 * NZ > s->ct_cap) {
        HIP_TRY(hipStreamSynchronize(strm));
        if (s->Ctmp) hipFree(s->Ctmp);
        s->Ctmp = nullptr; s->Ctmp = dal<float>((int64_t)G4 * NZ); s->ct_cap = (int64_t)G4 * NZ;
      }
      bf16* zt_e = s->ZT16;                              // rows [0, de): x_e^T, the entity slice of the step input
      bf16* zt_s = s->ZT16 + (int64_t)c.de * ldz;        // rows [de, de + 128): the one-hot selectors
      bf16* zt_h = s->ZT16 + (int64_t)(c.de + 128) * ldz;   // rows behind them: h_{t-1}^T (step block 0 zero)
      {
        ProfScope ps(h, "bf16_transposes", side);   // (side stream: beside the BPTT launch queued above -- nothing here reads what it writes)
        hipLaunchKernelGGL(k_gather16_T, dim3((unsigned)((Np + 63) / 64), (unsigned)((c.de + 63) / 64), (unsigned)T), dim3(256), 0, side, b->idx, N, Np, T, b->F, 1,
                           s->We16, s->We16, s->We16, 0, c.de, 0, zt_e, ldz);
        hipLaunchKernelGGL(k_onehot_T, dim3((unsigned)((Np + 63) / 64), (unsigned)T), dim3(256), 0, side, b->idx, N, Np, T, b->F, c.Vr, c.Vt, zt_s, ldz);
        HIP_TRY(hipMemset2DAsync(zt_h, (size_t)ldz * sizeof(bf16), 0, (size_t)Np * sizeof(bf16), (size_t)H, side));
        hipLaunchKernelGGL(k_hfrag_T, dim3((unsigned)((s->sv.NU + 1) / 2), (unsigned)(H / 64), (unsigned)(T - 1)), dim3(256), 0, side, (const bf16x4*)s->sv.HsF, zt_h + Np,
                           (int64_t)s->sv.NU, (int64_t)s->sv.step_recs, N, Np, ldz, H);
        HIP_TRY(hipGetLastError());
      }
      if (overlap) HIP_TRY(hipEventRecord(s->ev_operands, side));
      if (!dxe_in_bptt) {
        ProfScope ps(h, "gemm_i2g_bwd_dx_e");   // dx_e [T N][de] = dA W_i2g[:, entity columns] (compact: what the entity gather-reduce reads)
        const bool ran = gemm16xt(strm, s->dAT16, ldz, wt + (int64_t)c.dt * G4, G4, w.dIn, c.de, TNp, c.de, G4, Np, N);
        KPRN_REQUIRE(ran, KPRN_E_ARG, "bf16 backward: the transposed-dA dx product does not cover this shape (dx_from_transposed_ok said it would)");
      }
      if (overlap) {
        HIP_TRY(hipEventRecord(s->ev_dx, strm));
        HIP_TRY(hipStreamWaitEvent(side, s->ev_dx, 0));
        HIP_TRY(hipStreamWaitEvent(strm, s->ev_operands, 0));
      }
      {
        ProfScope ps(h, "entity_grad", side);   // (beside the merged dW product)
        bidx::entity_grad(side, w.dIn, /*frag_order=*/0, b->key_sorted, b->pos_sorted, b->n_index, N, T, c.de, 0, c.de, c.Ve, h->g_We);
      }
      {
        ProfScope ps(h, "gemm_bwd_dw_merged");   // Ct [4H][NZ] = dA^T [x_e^T | S^T | h_{t-1}^T]^T, split-K
        HIP_TRY(hipMemsetAsync(s->Ctmp, 0, (size_t)G4 * NZ * sizeof(float), strm));
        const int split = (int)std::min<int64_t>(1024, std::max<int64_t>(1, TN / 4096));
        // (h_{t-1}^T has no step -1: its column tiles skip the K ranges of step block 0 -- a tenth of the launch's workgroups; 8 K ranges per XCD so that
        //  whole workgroups fall into that block)
        if (!gemm16x(strm, s->dAT16, ldz, s->ZT16, ldz, s->Ctmp, NZ, G4, NZ, TNp, true, split, c.de + 128, Np, 8))
          gemm16(strm, s->dAT16, ldz, s->ZT16, ldz, s->Ctmp, NZ, G4, NZ, TNp, true, nullptr, split);
      }
      {
        ProfScope ps(h, "small_tables_finish");
        FinArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.Ct = s->Ctmp; fa.NZ = NZ; fa.G4 = G4; fa.Din = Din; fa.H = H; fa.dt = c.dt; fa.de = c.de; fa.dr = c.dr; fa.Vt = c.Vt; fa.Vr = c.Vr;
        fa.Wt16 = s->dense16 + h->off_Wt; fa.Wr16 = s->dense16 + h->off_Wr; fa.Wi16 = s->dense16 + h->layer[l].Wi;
        fa.gWi = gd + h->layer[l].Wi; fa.gWo = gd + h->layer[l].Wo; fa.gWt = gd + h->off_Wt; fa.gWr = gd + h->off_Wr;
        hipLaunchKernelGGL(k_small_tables_finish, dim3((unsigned)(G4 + c.Vr + c.Vt)), dim3(256), 0, strm, fa);
        HIP_TRY(hipGetLastError());
      }
      if (overlap) {
        HIP_TRY(hipEventRecord(s->ev_join, side));
        HIP_TRY(hipStreamWaitEvent(strm, s->ev_join, 0));
        side_guard.joined = true;
      }
      return;   // (L = 1: nothing below this layer)
    }
    {
      // (with the side stream: in^T and h^T are built there, beside the BPTT launch queued above -- neither reads anything it writes)
      ProfScope ps(h, "bf16_transposes", side);
      if (!(s->act_frag && l == 0)) transpose_steps(strm, s->dA16, s->dAT16, T, N, Np, G4);                       // dA^T [4H][T][Np] (the fragment-order gate backward wrote it)
      if (l == 0 && s->act_frag) {   // in^T gathered straight into the transposed layout (the forward was the persistent kernel: no X16 plane)
        hipLaunchKernelGGL(k_gather16_T, dim3((unsigned)((Np + 63) / 64), (unsigned)((Din + 63) / 64), (unsigned)T), dim3(256), 0, side, b->idx, N, Np, T, b->F,
                           c.num_types, s->dense16 + h->off_Wt, s->We16, s->dense16 + h->off_Wr, c.dt, c.de, c.dr, s->XT16, (int64_t)T * Np);
        HIP_TRY(hipGetLastError());
      } else transpose_steps(strm, (l == 0) ? s->X16 : s->H16 + (int64_t)(l - 1) * TN * H, s->XT16, T, N, Np, Din);      // in^T [Din][T][Np]
      if (T > 1 && s->act_frag && l == 0 && s->sv.HsF && s->sv.NW == 8 && (H % 64) == 0) {   // h^T straight from the persistent forward's fragment-order records (steps 0 .. T-2)
        hipLaunchKernelGGL(k_hfrag_T, dim3((unsigned)((s->sv.NU + 1) / 2), (unsigned)(H / 64), (unsigned)(T - 1)), dim3(256), 0, side, (const bf16x4*)s->sv.HsF, s->HT16,
                           (int64_t)s->sv.NU, (int64_t)s->sv.step_recs, N, Np, (int64_t)T * Np, H);
        HIP_TRY(hipGetLastError());
      } else if (T > 1) {
        // (the persistent forward writes no row-major h plane: its records are the only copy, and k_hfrag_T above is their only reader)
        KPRN_REQUIRE(!(s->act_frag && l == 0), KPRN_E_ARG, "bf16 backward: h^T of a persistent forward needs its fragment-order records (NW = 8, H a multiple of 64)");
        transpose_steps(strm, hs, s->HT16, T, N, Np, H);                                                        // h^T  [H][T][Np]
      }
    }
    if (overlap) {
      HIP_TRY(hipEventRecord(s->ev_operands, side));
      // dx first: the table gradients (side stream) then run beside the two dW products
      {
        ProfScope ps(h, "gemm_i2g_bwd_dx");   // dx [T N][Din] = dA W_i2g
        if (dx_t) {   // (the BPTT launch wrote no row-major dA: there is nothing to fall back to)
          const bool ran = gemm16xt(strm, s->dAT16, TNp, wt, G4, w.dIn, Din, TNp, Din, G4, Np, N);
          KPRN_REQUIRE(ran, KPRN_E_ARG, "bf16 backward: the transposed-dA dx product does not cover this shape (dx_from_transposed_ok said it would)");
        } else gemm16(strm, s->dA16, G4, wt, G4, w.dIn, Din, TN, Din, G4, false, nullptr, 1);
      }
      HIP_TRY(hipEventRecord(s->ev_dx, strm));
      HIP_TRY(hipStreamWaitEvent(side, s->ev_dx, 0));
      HIP_TRY(hipStreamWaitEvent(strm, s->ev_operands, 0));
    }
    const int split = (int)std::min<int64_t>(1024, std::max<int64_t>(1, TN / 4096));
    {
      ProfScope ps(h, "gemm_i2g_bwd_dw");   // gW_i2g [4H][Din] += dA^T in
      gemm16(strm, s->dAT16, TNp, s->XT16, TNp, gd + h->layer[l].Wi, Din, G4, Din, TNp, true, nullptr, split);
    }
    if (T > 1) {
      ProfScope ps(h, "gemm_o2g_bwd_dw");   // gW_o2g [4H][H] += dA[1..T-1]^T h[0..T-2]
      gemm16(strm, s->dAT16 + Np, TNp, s->HT16, TNp, gd + h->layer[l].Wo, H, G4, H, (int64_t)(T - 1) * Np, true, nullptr, split);
    }
    if (!(s->act_frag && l == 0 && s->bias_in_gates)) {   // (else the fragment-order gate backward summed the bias gradient as it produced dA)
      ProfScope ps(h, "bias_colsum");
      hipLaunchKernelGGL(k_rowsum16, dim3((unsigned)G4), dim3(256), 0, strm, s->dAT16, TNp, gd + h->layer[l].bi);
      HIP_TRY(hipGetLastError());
    }
    if (!overlap) {
      ProfScope ps(h, "gemm_i2g_bwd_dx");   // dx [T N][Din] = dA W_i2g
      if (dx_t) {
        const bool ran = gemm16xt(strm, s->dAT16, TNp, wt, G4, w.dIn, Din, TNp, Din, G4, Np, N);
        KPRN_REQUIRE(ran, KPRN_E_ARG, "bf16 backward: the transposed-dA dx product does not cover this shape (dx_from_transposed_ok said it would)");
      } else gemm16(strm, s->dA16, G4, wt, G4, w.dIn, Din, TN, Din, G4, false, nullptr, 1);
    }
  }
  // (with the side stream these were queued behind ev_dx, i.e. they run beside the dW products issued above)
  {
    ProfScope ps(h, "embed_scatter", side);
    const bool have_index = b->key_sorted != nullptr && !b->tile_k;
    kk::embed_scatter(side, b->idx, N, T, b->F, c.num_types, w.dIn, c.dt, c.de, c.dr, c.Vt, c.Vr, gd + h->off_Wt, h->g_We, gd + h->off_Wr, have_index);
  }
  if (b->key_sorted != nullptr && !b->tile_k) {
    ProfScope ps(h, "entity_grad", side);
    bidx::entity_grad(side, w.dIn, /*frag_order=*/0, b->key_sorted, b->pos_sorted, b->n_index, N, T, D, c.dt, c.de, c.Ve, h->g_We);
  }
  if (overlap) {
    HIP_TRY(hipEventRecord(s->ev_join, side));
    HIP_TRY(hipStreamWaitEvent(strm, s->ev_join, 0));
    side_guard.joined = true;
  }
}

// measurement hook (kprn_debug_gemm what = 5 / 6): C[M][N] = A[M][K] B[N][K]^T on random bf16 operands; split_k > 1: the split-K accumulate form
float debug_gemm16(hipStream_t strm, int64_t M, int N, int64_t K, int split_k, int iters) {
  float* f = dal<float>(std::max<int64_t>(M * K, (int64_t)N * K));
  bf16 *A = dal<bf16>(M * K), *B = dal<bf16>((int64_t)N * K);
  float* C = dal<float>(M * N);
  kk::fill_uniform(strm, f, M * K, 0.1f, 1, 0);
  hipLaunchKernelGGL(k_cvt, dim3((unsigned)((M * K / 4 + 256) / 256)), dim3(256), 0, strm, f, A, M * K);
  kk::fill_uniform(strm, f, (int64_t)N * K, 0.1f, 2, 0);
  hipLaunchKernelGGL(k_cvt, dim3((unsigned)(((int64_t)N * K / 4 + 256) / 256)), dim3(256), 0, strm, f, B, (int64_t)N * K);
  hipMemsetAsync(C, 0, (size_t)(M * N) * sizeof(float), strm);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = -2; it < iters; ++it) {
    if (it == 0) hipEventRecord(e0, strm);
    gemm16(strm, A, K, B, K, C, N, M, N, K, split_k > 1, nullptr, split_k);
  }
  hipEventRecord(e1, strm);
  hipEventSynchronize(e1);
  float t = 0.f;
  hipEventElapsedTime(&t, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  hipFree(f); hipFree(A); hipFree(B); hipFree(C);
  return t / (float)iters;
}

}  // namespace bf16p
