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
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "kprn_internal.h"

namespace fused {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DH = 64;        // D == H == 64 in this variant
constexpr int MT = 64;        // paths per tile
constexpr int LDA = DH + 4;   // LDS row stride (floats): 16-byte aligned, spreads ds_read_b128 slots

struct FwdArgs {
  const int32_t* idx;  // [N][T][F] 1-based
  int64_t N;
  int T, F, nT;
  const float *Wt, *We, *Wr;
  int dt, de, dr;
  const float* Wi[2];
  const float* bi[2];
  const float* Wo[2];
  const float* Wout;
  const float* bout;
  int C;
  float* S;          // [N][C]
  float* save_frag;  // training: [(N/16)][T][L][4 waves][5: i,g,f,o,c][64 lanes][4]   (nullable)
  float* save_h;     // training: [T][L][N][H] row-major                               (nullable)
  int64_t n_tiles;
};

// v_exp_f32 + v_rcp_f32 (1 ulp each): ~1e-7 absolute error on the gate values, far inside the 1e-4 score bar
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits for
// every global store / atomic / prefetch load in flight (the activation saves of the training forward,
// the embedding-gradient atomics, the next step's gather) -- none of which the other waves depend on.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// float add on an LDS address as ds_add_f32.  (atomicAdd() through a generic pointer was emitted as
// flat_atomic_add_f32, the slow aperture path.)
__device__ __forceinline__ void lds_atomic_add(float* p, float v) {
  typedef __attribute__((address_space(3))) float lds_float;
  __hip_atomic_fetch_add((lds_float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

constexpr int MAXT_LDS = 16;  // steps whose ids are staged in LDS per tile

// all the tile's ids -> LDS: ids[(row*T + t)*4 + {0: first type, 1: entity, 2: relation}] (0-based).
// Rows past N repeat row N-1 (their results are never stored).  Removes the dependent id -> row load
// chain from every step's gather.
template <int NTHREADS>
__device__ __forceinline__ void ids_stage(const int32_t* idx, int64_t N, int T, int F, int nT, int64_t tile, int32_t* ids) {
  for (int c = threadIdx.x; c < MT * T; c += NTHREADS) {
    const int row = c / T, t = c - row * T;
    int64_t n = tile * MT + row;
    if (n >= N) n = N - 1;
    const int32_t* f = idx + (n * T + t) * F;
    ids[c * 4 + 0] = f[F - nT - 2] - 1;
    ids[c * 4 + 1] = f[F - 2] - 1;
    ids[c * 4 + 2] = f[F - 1] - 1;
  }
}

// slot 3 of the id tile <- precomputed tile leader (fused backward, bottom layer)
__device__ __forceinline__ void lead_stage(const int32_t* lead, int T, int64_t tile, int32_t* ids) {
  for (int c = threadIdx.x; c < MT * T; c += 256) ids[c * 4 + 3] = lead[tile * MT * T + c];
}

// gather this thread's share of one step's x rows for `tile` into registers (ids from the LDS id tile)
template <int NTHREADS>
__device__ __forceinline__ void gather_load(const FwdArgs& a, int64_t tile, int t, const int32_t* ids, f32x4 (&v)[1024 / NTHREADS]) {
  constexpr int PER = 1024 / NTHREADS;  // 64 rows x 16 float4 chunks
  const int c_t = a.dt >> 2, c_e = (a.dt + a.de) >> 2;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int c = threadIdx.x + k * NTHREADS;
    const int row = c >> 4, ch = c & 15;
    const int32_t* id = ids + (row * a.T + t) * 4;
    f32x4 out;
    if (ch < c_t) {
      out = *(const f32x4*)(a.Wt + (int64_t)id[0] * a.dt + ch * 4);
      if (a.nT > 1) {
        int64_t n = tile * MT + row;
        if (n >= a.N) n = a.N - 1;
        const int32_t* f = a.idx + (n * a.T + t) * a.F;
        for (int q = 1; q < a.nT; ++q) out += *(const f32x4*)(a.Wt + (int64_t)(f[a.F - a.nT - 2 + q] - 1) * a.dt + ch * 4);
      }
    } else if (ch < c_e) {
      out = *(const f32x4*)(a.We + (int64_t)id[1] * a.de + (ch - c_t) * 4);
    } else {
      out = *(const f32x4*)(a.Wr + (int64_t)id[2] * a.dr + (ch - c_e) * 4);
    }
    v[k] = out;
  }
}

template <int NTHREADS>
__device__ __forceinline__ void gather_store(float* xbuf, const f32x4 (&v)[1024 / NTHREADS]) {
  constexpr int PER = 1024 / NTHREADS;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int c = threadIdx.x + k * NTHREADS;
    const int row = c >> 4, ch = c & 15;
    *(f32x4*)(xbuf + row * LDA + ch * 4) = v[k];
  }
}

constexpr int NPL = 6;  // saved planes per (16-row m-tile, t, layer, wave): i, g, f, o, c, h -- 1 KiB each, MFMA C-fragment order

// One quarter (accumulator register r) of the LSTM cell of one 16-row m-tile, lane-local on the MFMA
// accumulators: acc[q][r] <-> row 4*ag + r of the m-tile, hidden col 16j + arow.
template <bool SAVE, int R>
__device__ __forceinline__ void cell_q(const f32x4 (&acc)[4], float (&cst)[4], bool first, float* out_row, f32x4 (&sv)[NPL]) {
  const float ig = fast_sigmoid(acc[0][R]);
  const float gg = fast_tanh(acc[1][R]);
  const float fg = fast_sigmoid(acc[2][R]);
  const float og = fast_sigmoid(acc[3][R]);
  const float cp = first ? 0.f : cst[R];
  const float cc = fg * cp + ig * gg;
  const float hh = og * fast_tanh(cc);
  cst[R] = cc;
  out_row[R * LDA] = hh;
  if (SAVE) { sv[0][R] = ig; sv[1][R] = gg; sv[2][R] = fg; sv[3][R] = og; sv[4][R] = cc; sv[5][R] = hh; }
}

// Half of a unit's 4-gate GEMM: 4 k-groups (S) of 16 MFMAs over one LDS tile (the recurrent h_{t-1} tile or
// the step-input tile).  The A fragment of the NEXT group is always in flight (apre) while a group's MFMAs
// issue; with CELL the LSTM cell of the PREVIOUS unit (pacc) is interleaved under the MFMAs, one accumulator
// register per group, so the transcendental / VALU work hides in the MFMA shadow instead of following it.
template <bool SAVE, bool CELL, bool PF>
__device__ __forceinline__ void half_unit(const float* abase, const f32x4 (&w)[4][4], f32x4 (&acc)[4], f32x4& apre, const float* next_abase,
                                          const f32x4 (&pacc)[4], float (&pc)[4], bool pfirst, float* pout_row, f32x4 (&sv)[NPL]) {
#pragma unroll
  for (int S = 0; S < 4; ++S) {
    const f32x4 a4 = apre;
    if (S < 3) apre = *(const f32x4*)(abase + (S + 1) * 16);
    else if (PF) apre = *(const f32x4*)(next_abase);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], w[q][S][jj], acc[q], 0, 0, 0);
    if (CELL) {
      if (S == 0) cell_q<SAVE, 0>(pacc, pc, pfirst, pout_row, sv);
      if (S == 1) cell_q<SAVE, 1>(pacc, pc, pfirst, pout_row, sv);
      if (S == 2) cell_q<SAVE, 2>(pacc, pc, pfirst, pout_row, sv);
      if (S == 3) cell_q<SAVE, 3>(pacc, pc, pfirst, pout_row, sv);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // the A-fragment prefetch first
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x402, 2, 0);  // 2 VALU / transcendental of the cell
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool SAVE>
__device__ __forceinline__ void cell_all(const f32x4 (&pacc)[4], float (&pc)[4], bool pfirst, float* pout_row, f32x4 (&sv)[NPL]) {
  cell_q<SAVE, 0>(pacc, pc, pfirst, pout_row, sv);
  cell_q<SAVE, 1>(pacc, pc, pfirst, pout_row, sv);
  cell_q<SAVE, 2>(pacc, pc, pfirst, pout_row, sv);
  cell_q<SAVE, 3>(pacc, pc, pfirst, pout_row, sv);
}

// nn.Linear(H, C) on the tile's h_T (LDS) -> S[n][0..C)
__device__ __forceinline__ void head_tile(const FwdArgs& a, const float* hbuf, int64_t tile, int j, int lane) {
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
        const f32x4 a4 = *(const f32x4*)(hbuf + (mt * 16 + arow) * LDA + S * 16 + ag * 4);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], w4[S][jj], acc, 0, 0, 0);
      }
      if (cv) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t n = tile * MT + mt * 16 + ag * 4 + r;
          if (n < a.N) a.S[n * a.C + col] = acc[r];
        }
      }
    }
  }
}

template <int L, bool SAVE>
__global__ __launch_bounds__(256, 1) void k_lstm_fwd(FwdArgs a) {
  constexpr int NT = 256;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // LDS carve (floats): x double buffer | h(layer l) double buffer, l = 0..L-1
  auto xbuf = [&](int i) -> float* { return lds + i * (MT * LDA); };
  auto hbuf = [&](int g, int i) -> float* { return lds + (2 + 2 * g + i) * (MT * LDA); };
  // id tiles (double-buffered by tile parity): [64][T][4] ints each
  auto idbuf = [&](int i) -> int32_t* { return (int32_t*)(lds + (2 + 2 * L) * (MT * LDA)) + i * (MT * MAXT_LDS * 4); };

  const int lane = threadIdx.x & 63;
  const int j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // hidden tile owned by this wave
  const int arow = lane & 15, ag = lane >> 4;

  // ---- register-stationary weights of EVERY layer: rows (q*H + 16j + arow), 16-byte pieces at k = 16S + 4ag
  f32x4 wi[L][4][4], wo[L][4][4];
  float bias[L][4];
#pragma unroll
  for (int l = 0; l < L; ++l) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t row = (int64_t)q * DH + j * 16 + arow;
      bias[l][q] = a.bi[l][row];
#pragma unroll
      for (int S = 0; S < 4; ++S) {
        wi[l][q][S] = *(const f32x4*)(a.Wi[l] + row * DH + S * 16 + ag * 4);
        wo[l][q][S] = *(const f32x4*)(a.Wo[l] + row * DH + S * 16 + ag * 4);
      }
    }
  }
  float c[L][4][4];
#pragma unroll
  for (int l = 0; l < L; ++l)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) c[l][m][r] = 0.f;

  const int T = a.T;
  const int64_t my_tiles = (a.n_tiles > blockIdx.x) ? (a.n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const int64_t total_slots = my_tiles * T;
  if (my_tiles == 0) return;

  f32x4 gv[1024 / NT];
  ids_stage<NT>(a.idx, a.N, T, a.F, a.nT, blockIdx.x, idbuf(0));
  lds_barrier();
  gather_load<NT>(a, blockIdx.x, 0, idbuf(0), gv);
  gather_store<NT>(xbuf(0), gv);

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

  auto save_unit = [&](int64_t p_tile, int p_t, int pl, int pm) {
    if (!SAVE) return;
    float* fb = a.save_frag + (p_tile * 4 + pm) * frag_mt_stride + ((int64_t)(p_t * L + pl) * 4 + j) * frag_unit + lane * 4;
#pragma unroll
    for (int k = 0; k < NPL; ++k) *(f32x4*)(fb + k * 256) = sv[k];
  };
  // training: the complete h tile of (tile, t, layer l) -> save_h[t][l][n][:] row-major, coalesced 16-byte stores
  auto copy_h = [&](int64_t f_tile, int f_t, int l, const float* hb) {
    if (!SAVE) return;
    float* dst = a.save_h + (((int64_t)f_t * L + l) * a.N + f_tile * MT) * DH;
    const int64_t rows_valid = a.N - f_tile * MT;
#pragma unroll
    for (int k = 0; k < 1024 / NT; ++k) {
      const int cch = threadIdx.x + k * NT;
      const int row = cch >> 4, ch = cch & 15;
      if (row < rows_valid) *(f32x4*)(dst + row * DH + ch * 4) = *(const f32x4*)(hb + row * LDA + ch * 4);
    }
  };

  auto slot = [&](auto first_tag, const int64_t tile, const int t, const int par, const bool has_prev, const int64_t p_tile, const int p_t,
                  const bool p_first) {
    constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float* in_buf = (l == 0) ? xbuf(par) : hbuf(l - 1, par);
      const float* hp_buf = hbuf(l, par ^ 1);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        // the unit whose cell is still outstanding
        const int pl = (mt > 0) ? l : ((l > 0) ? l - 1 : L - 1);
        const int pm = (mt > 0) ? mt - 1 : 3;
        const bool cross = (l == 0 && mt == 0);             // it belongs to the previous slot
        const int64_t q_tile = cross ? p_tile : tile;
        const int q_t = cross ? p_t : t;
        const int q_par = cross ? (par ^ 1) : par;
        const bool q_first = cross ? p_first : FIRST;
        float* pout = hbuf(pl, q_par) + pm * 16 * LDA + o_off;
        f32x4(&acc)[4] = accs[mt & 1];
        f32x4(&pacc)[4] = accs[(mt & 1) ^ 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{bias[l][q], bias[l][q], bias[l][q], bias[l][q]};
        const float* in_base = in_buf + mt * 16 * LDA + a_off;
        // first A fragment of the unit that follows this one (always a readable LDS address; unused when that
        // unit starts behind a barrier)
        const float* nxt;
        if (mt < 3) nxt = (FIRST ? in_buf : hp_buf) + (mt + 1) * 16 * LDA + a_off;
        else if (l + 1 < L) nxt = hbuf(l + 1, par ^ 1) + a_off;
        else nxt = hbuf(0, par) + a_off;
        if (!FIRST) {
          half_unit<SAVE, true, true>(hp_buf + mt * 16 * LDA + a_off, wo[l], acc, apre, in_base, pacc, c[pl][pm], q_first, pout, sv);
          save_unit(q_tile, q_t, pl, pm);
          if (mt == 0) {
            lds_barrier();
            copy_h(q_tile, q_t, pl, hbuf(pl, q_par));
            apre = *(const f32x4*)(in_base);
          }
          half_unit<SAVE, false, true>(in_base, wi[l], acc, apre, nxt, pacc, c[pl][pm], q_first, pout, sv);
        } else if (mt == 0) {
          if (!cross || has_prev) {
            cell_all<SAVE>(pacc, c[pl][pm], q_first, pout, sv);
            save_unit(q_tile, q_t, pl, pm);
          }
          lds_barrier();
          if (!cross || has_prev) copy_h(q_tile, q_t, pl, hbuf(pl, q_par));
          if (cross && has_prev) head_tile(a, hbuf(L - 1, q_par), p_tile, j, lane);
          apre = *(const f32x4*)(in_base);
          half_unit<SAVE, false, true>(in_base, wi[l], acc, apre, nxt, pacc, c[pl][pm], q_first, pout, sv);
        } else {
          half_unit<SAVE, true, true>(in_base, wi[l], acc, apre, nxt, pacc, c[pl][pm], q_first, pout, sv);
          save_unit(q_tile, q_t, pl, pm);
        }
      }
    }
  };

  int64_t tile = blockIdx.x;
  int t = 0;
  int tpar = 0;  // parity of the tile's id buffer
  int64_t p_tile = tile;
  int p_t = 0;
  for (int64_t s = 0; s < total_slots; ++s) {
    const int par = (int)(s & 1);
    // (1) issue the gather for the NEXT slot (latency hidden under this slot's MFMAs)
    int tn = t + 1;
    int64_t tile_n = tile;
    int tpar_n = tpar;
    if (tn == T) { tn = 0; tile_n += gridDim.x; tpar_n ^= 1; }
    const bool have_next = (s + 1) < total_slots;
    // the next tile's ids are staged while this tile's first step computes (visible after >= 1 barrier)
    if (t == 0 && tile + gridDim.x < a.n_tiles) ids_stage<NT>(a.idx, a.N, T, a.F, a.nT, tile + gridDim.x, idbuf(tpar ^ 1));
    if (have_next) gather_load<NT>(a, tile_n, tn, idbuf(tpar_n), gv);
    // (2) the units of this slot
    if (t == 0) slot(std::true_type{}, tile, t, par, s > 0, p_tile, p_t, p_t == 0);
    else slot(std::false_type{}, tile, t, par, true, p_tile, p_t, p_t == 0);
    // (3) land the gathered rows of the next slot (xbuf[par^1] was last read one slot ago); visible to the
    //     other waves after the next slot's first barrier
    if (have_next) gather_store<NT>(xbuf(par ^ 1), gv);
    p_tile = tile; p_t = t;
    t = tn; tile = tile_n; tpar = tpar_n;
  }
  // drain: the cell of the very last unit, then the last tile's head
  {
    const int par = (int)((total_slots - 1) & 1);
    cell_all<SAVE>(accs[1], c[L - 1][3], p_t == 0, hbuf(L - 1, par) + 3 * 16 * LDA + o_off, sv);
    save_unit(p_tile, p_t, L - 1, 3);
    lds_barrier();
    copy_h(p_tile, p_t, L - 1, hbuf(L - 1, par));
    head_tile(a, hbuf(L - 1, par), p_tile, j, lane);
  }
}

// ============================================================================================
// Fused backward (BPTT) -- one launch per layer, top layer first.
//
// Stands in for the backward of nn.Sequencer(nn.FastLSTM) x L + FeatureEmbedding
// (model/OneModel.lua:223-274 via MyOptimizer.lua:195 model:backward): exact BPTT over all T
// steps, weight gradients accumulated over steps, LookupTable scatter-add for layer 1.
//
// Per 64-path tile, t = T-1 .. 0:
//   A. stage the step's input tile (x_t re-gathered, or h^{l-1}_t) and h^l_{t-1} row-major in LDS
//   C. per 16-row m-tile: cell backward, lane-local on the saved gate fragments (same wave <-> hidden
//      tile ownership as the forward) -> dA (pre-activation grads) in the MFMA C layout.  Those
//      registers ARE the A operand of dW = dA^T [x | h]  (k-slot = lane>>4 <-> row 4*(lane>>4)+r), so
//      dW += is issued straight away with B fragments read from the LDS tiles; dW (128 VGPRs) and db
//      stay in registers for the WHOLE launch and are flushed with atomics once per workgroup.
//      dA is also written row-major to LDS.
//   E. [dx | dh_{t-1}] = dA [W_i2g | W_o2g]: A from LDS (ds_read_b128), B = one 16-byte load of the
//      TRANSPOSED weights (WT[n][k], rebuilt when parameters change) streamed from L2.  dh_{t-1} goes
//      to an LDS tile for the next (earlier) step; dx goes to HBM for the layer below, or -- bottom
//      layer -- is scattered to the embedding gradients directly from the accumulators (types /
//      relations via LDS partial sums, entities via L2 atomics).
struct BwdArgs {
  const int32_t* idx; int64_t N; int T, F, nT;
  const float *Wt, *We, *Wr; int dt, de, dr; int Vt, Vr;
  int L, layer;
  const float* WiT;        // [64][256]  = W_i2g^T of this layer
  const float* WoT;        // [64][256]
  const float* save_frag;  // forward's fragment-order gates
  const float* save_h;     // [T][L][N][64]
  const float* dHhead;     // [N][64] (top layer) or null
  float* DX;               // [T][Npad][64]: in = dx of the layer above (not top), out = dx of this layer (not bottom)
  int64_t Npad;            // N rounded up to the 64-row tile
  float* gWi; float* gbi; float* gWo;   // [256][64], [256], [256][64]
  float* gWt; float* gWe; float* gWr;   // bottom layer only
  float* part;             // [grid][2*256*64 + 256] per-workgroup partial dW_i2g | dW_o2g | db
  unsigned long long* timing;  // optional [grid][8] cycle counters (KPRN_TIMING=1)
  int dbg;                 // KPRN_DBG bit 0: skip the embedding scatter (measurement only)
  const int32_t* lead;     // [Npad][T] tile leaders (bottom layer)
  int mfma_scatter;        // 1: one-hot MFMA scatter (dims multiples of 16, one type slot, tables <= 16 rows)
  int64_t n_tiles;
};

constexpr int PART = 2 * 256 * 64 + 256;  // floats per workgroup partial slab
constexpr int LDD = 4 * DH + 4;  // dA tile row stride

// stage-A helper: this thread's 4 float4 chunks of the step's input tile and h_{t-1} tile
template <bool BOTTOM>
__device__ __forceinline__ void bwd_tile_load(const BwdArgs& a, int64_t n0, int t, int tid, const int32_t* ids, f32x4 (&vin)[4], f32x4 (&vhp)[4]) {
  const int c_t = a.dt >> 2, c_e = (a.dt + a.de) >> 2;
  const int T = a.T, L = a.L, ly = a.layer;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = tid + k * 256;
    const int row = c >> 4, ch = c & 15;
    const int64_t n = n0 + row;
    vin[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    vhp[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n < a.N) {
      if (BOTTOM) {
        const int32_t* id = ids + (row * T + t) * 4;
        if (ch < c_t) {
          vin[k] = *(const f32x4*)(a.Wt + (int64_t)id[0] * a.dt + ch * 4);
          if (a.nT > 1) {
            const int32_t* f = a.idx + (n * T + t) * a.F;
            for (int q = 1; q < a.nT; ++q) vin[k] += *(const f32x4*)(a.Wt + (int64_t)(f[a.F - a.nT - 2 + q] - 1) * a.dt + ch * 4);
          }
        } else if (ch < c_e) {
          vin[k] = *(const f32x4*)(a.We + (int64_t)id[1] * a.de + (ch - c_t) * 4);
        } else {
          vin[k] = *(const f32x4*)(a.Wr + (int64_t)id[2] * a.dr + (ch - c_e) * 4);
        }
      } else {
        vin[k] = *(const f32x4*)(a.save_h + (((int64_t)t * L + (ly - 1)) * a.N + n) * DH + ch * 4);
      }
      if (t > 0) vhp[k] = *(const f32x4*)(a.save_h + (((int64_t)(t - 1) * L + ly) * a.N + n) * DH + ch * 4);
    }
  }
}

__device__ __forceinline__ void bwd_tile_store(float* in_t, float* hp_t, int tid, const f32x4 (&vin)[4], const f32x4 (&vhp)[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = tid + k * 256;
    const int row = c >> 4, ch = c & 15;
    *(f32x4*)(in_t + row * LDA + ch * 4) = vin[k];
    *(f32x4*)(hp_t + row * LDA + ch * 4) = vhp[k];
  }
}

struct Frag6 { f32x4 i, g, f, o, c, cp, up; };
template <bool REC>
__device__ __forceinline__ void frag_load(Frag6& fr, const float* fb, const float* fbp) {
  fr.i = *(const f32x4*)(fb + 0 * 256);
  fr.g = *(const f32x4*)(fb + 1 * 256);
  fr.f = *(const f32x4*)(fb + 2 * 256);
  fr.o = *(const f32x4*)(fb + 3 * 256);
  fr.c = *(const f32x4*)(fb + 4 * 256);
  if (REC) fr.cp = *(const f32x4*)(fbp + 4 * 256);
  else fr.cp = f32x4{0.f, 0.f, 0.f, 0.f};
}

#define TPROBE(slot)                                                                 \
  if (a.timing) {                                                                     \
    const unsigned long long now__ = __builtin_amdgcn_s_memtime();                    \
    tacc[slot] += now__ - tlast;                                                      \
    tlast = now__;                                                                    \
  }

template <bool BOTTOM, bool TOP, bool MSCAT>
__global__ __launch_bounds__(256, 1) void k_lstm_bwd(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = a.timing ? __builtin_amdgcn_s_memtime() : 0ull;
  float* in_t = lds;                     // [64][LDA]  x_t or h^{l-1}_t
  float* hp_t = lds + MT * LDA;          // [64][LDA]  h^l_{t-1}
  float* dhr = lds + 2 * MT * LDA;       // [64][LDA]  recurrent dh for the step being processed
  float* dA_t = lds + 3 * MT * LDA;      // [64][LDD]
  int32_t* ids = (int32_t*)(dA_t + MT * LDD);        // [64][T][4] the tile's ids (bottom layer), 0-based
  int32_t* lead = ids + MT * MAXT_LDS * 4;           // [64] leader row of each row's entity id inside the tile
  float* dxt = (float*)(lead + MT);                  // [64][LDA] bottom layer: dx tile staged for the combine + scatter
  float* small_g = dxt + (BOTTOM ? MT * LDA : 0);    // [Vt*dt + Vr*dr] bottom layer partial sums (if they fit)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int arow = lane & 15, ag = lane >> 4;
  const int T = a.T, L = a.L, ly = a.layer;
  const int n_small = BOTTOM ? (a.Vt * a.dt + a.Vr * a.dr) : 0;
  const bool small_in_lds = BOTTOM && !MSCAT && n_small <= 4096;
  if (small_in_lds) for (int i = tid; i < n_small; i += 256) small_g[i] = 0.f;

  // launch-persistent accumulators: dW_i2g / dW_o2g rows (q*64 + 16j + 4ag + r), cols 16nt + arow
  f32x4 dwi[4][4], dwo[4][4];
  float dbias[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    dbias[q] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { dwi[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dwo[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }

  const int64_t frag_mt_stride = (int64_t)T * L * 4 * NPL * 256;
  // one-hot MFMA scatter: class of this wave's 16 columns (0 type, 1 entity, 2 relation) and the launch-
  // persistent accumulator of its small table: acc_s[r] <-> table row 4ag + r, column 16j + arow
  constexpr bool mscat = BOTTOM && MSCAT;  // compile-time: the two scatter forms never share a register allocation
  const int wcls = (j * 16 < a.dt) ? 0 : ((j * 16 < a.dt + a.de) ? 1 : 2);
  f32x4 acc_s = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
    const int64_t n0 = tile * MT;
    float dc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) dc[m][r] = 0.f;
    lds_barrier();  // previous tile fully consumed before its LDS tiles are overwritten
    if (BOTTOM) { ids_stage<256>(a.idx, a.N, T, a.F, a.nT, tile, ids); if (mscat) lead_stage(a.lead, T, tile, ids); lds_barrier(); }
    // recurrent dh starts at 0 (or at the head gradient for the top layer)
    for (int c = tid; c < MT * 16; c += 256) {
      const int row = c >> 4, ch = c & 15;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (TOP && n0 + row < a.N) v = *(const f32x4*)(a.dHhead + (n0 + row) * DH + ch * 4);
      *(f32x4*)(dhr + row * LDA + ch * 4) = v;
    }
    {
      f32x4 vin[4], vhp[4];
      bwd_tile_load<BOTTOM>(a, n0, T - 1, tid, ids, vin, vhp);
      bwd_tile_store(in_t, hp_t, tid, vin, vhp);
    }
    lds_barrier();
    TPROBE(0)  // tile prologue

    // the step body is compiled twice (REC: t > 0, there is an h_{t-1} / c_{t-1}) so that no MFMA sits
    // under a run-time condition: a branch inside the MFMA loops made hipcc emit a jump plus dozens of
    // accumulator copies around every MFMA (stage E measured 4x over its MFMA time)
    auto step = [&](auto rec_tag, const int t) {
      constexpr bool REC = decltype(rec_tag)::value;
      TPROBE(1)

      // ---- C. cell backward + dW, one m-tile at a time; saved gate fragments one m-tile ahead --------
      const float* fr_base = a.save_frag + (((tile * 4) * T + t) * L + ly) * (4 * NPL * 256) + (int64_t)j * (NPL * 256) + lane * 4;
      const float* frp_base = REC ? a.save_frag + (((tile * 4) * T + (t - 1)) * L + ly) * (4 * NPL * 256) + (int64_t)j * (NPL * 256) + lane * 4 : nullptr;
      // dx of the layer above for this wave's rows / columns (C layout) rides along with the fragments
      // (DX rows past N are written as exact zeros by the layer above, so no tail handling here)
      const float* dxp = TOP ? nullptr : a.DX + ((int64_t)t * a.Npad + n0 + ag * 4) * DH + j * 16 + arow;
      auto up_load = [&](Frag6& f6, int mt) {
        if (TOP) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) f6.up[r] = dxp[(mt * 16 + r) * DH];
      };
      Frag6 fr[2];
      frag_load<REC>(fr[0], fr_base, frp_base);
      up_load(fr[0], 0);
      // B operands of the dW product: [in | h_prev][row = mt*16 + 4ag + r][16nt + arow], one group ahead
      float bq[8];
      auto load_b = [&](float (&b)[8], int mt2, int r2) {
        const int row = mt2 * 16 + ag * 4 + r2;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          b[nt] = in_t[row * LDA + nt * 16 + arow];
          if constexpr (REC) b[4 + nt] = hp_t[row * LDA + nt * 16 + arow];
          else b[4 + nt] = 0.f;
        }
      };
      load_b(bq, 0, 0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        if (mt < 3) {
          frag_load<REC>(fr[(mt + 1) & 1], fr_base + (int64_t)(mt + 1) * frag_mt_stride, frp_base + (int64_t)(mt + 1) * frag_mt_stride);
          up_load(fr[(mt + 1) & 1], mt + 1);
        }
        const Frag6& F = fr[mt & 1];
        f32x4 dA[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mt * 16 + ag * 4 + r;
          float dh = dhr[row * LDA + j * 16 + arow];
          if (!TOP) dh += F.up[r];
          const float tc = fast_tanh(F.c[r]);
          const float dO = dh * tc;
          const float dC = dc[mt][r] + dh * F.o[r] * (1.f - tc * tc);
          dA[0][r] = dC * F.g[r] * F.i[r] * (1.f - F.i[r]);
          dA[1][r] = dC * F.i[r] * (1.f - F.g[r] * F.g[r]);
          dA[2][r] = dC * F.cp[r] * F.f[r] * (1.f - F.f[r]);
          dA[3][r] = dO * F.o[r] * (1.f - F.o[r]);
          dc[mt][r] = dC * F.f[r];
#pragma unroll
          for (int q = 0; q < 4; ++q) dA_t[row * LDD + q * DH + j * 16 + arow] = dA[q][r];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) dbias[q] += (dA[q][0] + dA[q][1]) + (dA[q][2] + dA[q][3]);
        // dW += dA^T [in | h_prev]: MFMA #r uses k-slot ag <-> row mt*16 + 4ag + r.
        // The B values of group (mt, r) were read from LDS one group earlier (bq); the reads for the
        // next group are issued before this group's 32 MFMAs.  (Left to itself hipcc emitted
        // read -> s_waitcnt -> 8 MFMAs, exposing the LDS latency on every group.)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float bc[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) bc[u] = bq[u];
          {
            const int g2 = mt * 4 + r + 1;  // next group (static after unrolling)
            if (g2 < 16) load_b(bq, g2 >> 2, g2 & 3);
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the next group's LDS reads ahead of this group's MFMAs
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dwi[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dA[q][r], bc[nt], dwi[q][nt], 0, 0, 0);
          }
          if constexpr (REC) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
              for (int q = 0; q < 4; ++q) dwo[q][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dA[q][r], bc[4 + nt], dwo[q][nt], 0, 0, 0);
            }
          }
        }
      }
      TPROBE(2)  // stage C (cell backward + dW MFMAs)
      lds_barrier();
      TPROBE(3)  // mid barrier wait
      // tiles of step t are dead now (stage C is the only reader): fetch the NEXT step's tiles into
      // registers here (their latency hides under stage E's MFMAs) and land them after those MFMAs --
      // keeping them out of stage C, which is the register-pressure peak of the kernel
      f32x4 nin[4], nhp[4];
      if (REC) bwd_tile_load<BOTTOM>(a, n0, t - 1, tid, ids, nin, nhp);

      // ---- E. [dx | dh_prev] = dA * [W_i2g | W_o2g]; this wave: columns 16j..16j+15 of each ----------
      {
        f32x4 ax[4], ah[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { ax[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; ah[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const float* wi_row = a.WiT + (int64_t)(j * 16 + arow) * (4 * DH) + ag * 4;
        const float* wo_row = a.WoT + (int64_t)(j * 16 + arow) * (4 * DH) + ag * 4;
#pragma unroll 4
        for (int S = 0; S < 16; ++S) {
          const f32x4 bi4 = *(const f32x4*)(wi_row + S * 16);
          f32x4 bo4 = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (REC) bo4 = *(const f32x4*)(wo_row + S * 16);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 a4 = *(const f32x4*)(dA_t + (mt * 16 + arow) * LDD + S * 16 + ag * 4);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              ax[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], bi4[jj], ax[mt], 0, 0, 0);
              if constexpr (REC) ah[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], bo4[jj], ah[mt], 0, 0, 0);
            }
          }
        }
        if (REC) bwd_tile_store(in_t, hp_t, tid, nin, nhp);
        const int col = j * 16 + arow;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = mt * 16 + ag * 4 + r;
            dhr[row * LDA + col] = ah[mt][r];  // read by step t-1 after the end-of-step barrier
            if (BOTTOM) { if (!mscat) dxt[row * LDA + col] = ax[mt][r]; }
            else a.DX[((int64_t)t * a.Npad + n0 + row) * DH + col] = ax[mt][r];  // rows past N are exact zeros (dA = 0 there)
          }
        }
        if constexpr (mscat) if (!(a.dbg & 1)) {
          // nn.LookupTable backward as a matrix product: grad_table[v][:] += sum_rows onehot(id[row] == v) dx[row][:].
          // The dx accumulators ax[mt][r] already sit in the MFMA B layout (k-slot ag <-> row mt*16+4ag+r),
          // the one-hot A operand is built from the LDS id tile; exact (products by 1.0 / 0.0).
          if (wcls != 1) {
            const int which = (wcls == 0) ? 0 : 2;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int32_t* id = ids + ((mt * 16 + ag * 4 + r) * T + t) * 4;
                const float oh = (id[which] == arow && id[3] >= 0) ? 1.f : 0.f;
                acc_s = __builtin_amdgcn_mfma_f32_16x16x4f32(oh, ax[mt][r], acc_s, 0, 0, 0);
              }
          } else {
            // entity rows: fold the tile's duplicate ids onto their leader row (every pad step hits ONE row,
            // a pair's user / item repeat across its paths), then one L2 atomic per distinct id
#pragma unroll
            for (int m = 0; m < 4; ++m) {  // one 16-leader block at a time: 4 live accumulator registers
              f32x4 comb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int ld = ids[((mt * 16 + ag * 4 + r) * T + t) * 4 + 3];
                  comb = __builtin_amdgcn_mfma_f32_16x16x4f32((ld == m * 16 + arow) ? 1.f : 0.f, ax[mt][r], comb, 0, 0, 0);
                }
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = m * 16 + ag * 4 + r;
                const int32_t* id = ids + (row * T + t) * 4;
                if (id[3] == row) unsafeAtomicAdd(a.gWe + (int64_t)id[1] * a.de + (col - a.dt), comb[r]);
              }
            }
          }
        }
      }
      TPROBE(4)  // landing + stage E (dX MFMAs)
      if constexpr (BOTTOM && !mscat) if (!(a.dbg & 1)) {
        // nn.LookupTable backward = scatter-add with duplicates accumulating (FeatureEmbedding.lua:29,41-49,86).
        // Pad steps all hit ONE entity row and a pair's user / item repeat across its paths, so rows
        // with the same entity id are first summed inside the tile (LDS); one L2 atomic row per distinct id.
        {
          // leader = first row of the tile with the same entity id; 4 threads per row, 16 candidates each,
          // all 16 id reads issued together (branch-free: one wave per SIMD hides no LDS latency for us)
          const int row = tid >> 2, part = tid & 3;
          const int e = ids[(row * T + t) * 4 + 1];
          int cand[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) cand[u] = ids[((part * 16 + u) * T + t) * 4 + 1];
          int ld = row;
#pragma unroll
          for (int u = 15; u >= 0; --u) {
            const int r2 = part * 16 + u;
            ld = (r2 < row && cand[u] == e) ? r2 : ld;
          }
          ld = min(ld, __shfl_xor(ld, 1, 64));
          ld = min(ld, __shfl_xor(ld, 2, 64));
          if (part == 0) lead[row] = (n0 + row < a.N) ? ld : -1;
        }
        lds_barrier();
        const int e0 = a.dt, e1 = a.dt + a.de;
        const int col = tid & 63, rg = tid >> 6;  // this thread: one column, rows rg, rg+4, ...
        const bool is_ent = col >= e0 && col < e1;
        int ldk[16];
        if (a.dbg & 4) {
        } else if (small_in_lds && a.nT == 1) {
          // (1) fold follower rows into their leader (entity slice); types / relations into the LDS small tables.
          //     One predicated ds_add_f32 per element, all operand reads batched up front.
          float vk[16];
          int idk[16];
          const int which = (col < e0) ? 0 : 2;
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int row = rg + 4 * k;
            vk[k] = dxt[row * LDA + col];
            ldk[k] = lead[row];
            idk[k] = ids[(row * T + t) * 4 + which];
          }
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int row = rg + 4 * k;
            float* p;
            bool go = ldk[k] >= 0;
            if (is_ent) { p = dxt + ldk[k] * LDA + col; go = go && (ldk[k] != row); }
            else if (col < e0) p = small_g + idk[k] * a.dt + col;
            else p = small_g + a.Vt * a.dt + idk[k] * a.dr + (col - e1);
            if (go) lds_atomic_add(p, vk[k]);
          }
        } else {
          // general form: several type slots per step and / or small tables too big for LDS
          for (int k = 0; k < 16; ++k) {
            const int row = rg + 4 * k;
            const int ld = lead[row];
            ldk[k] = ld;
            if (ld < 0) continue;
            const float v = dxt[row * LDA + col];
            const int32_t* id = ids + (row * T + t) * 4;
            if (col < e0) {
              const int32_t* f = a.idx + ((n0 + row) * T + t) * a.F;
              for (int kk = 0; kk < a.nT; ++kk) {
                const int rr = (kk == 0) ? id[0] : f[a.F - a.nT - 2 + kk] - 1;
                if (small_in_lds) lds_atomic_add(&small_g[rr * a.dt + col], v);
                else unsafeAtomicAdd(a.gWt + (int64_t)rr * a.dt + col, v);
              }
            } else if (col < e1) {
              if (ld != row) lds_atomic_add(&dxt[ld * LDA + col], v);
            } else {
              if (small_in_lds) lds_atomic_add(&small_g[a.Vt * a.dt + id[2] * a.dr + (col - e1)], v);
              else unsafeAtomicAdd(a.gWr + (int64_t)id[2] * a.dr + (col - e1), v);
            }
          }
        }
        lds_barrier();
        // (2) leaders add their (combined) entity slice to the gradient table (fire-and-forget atomics:
        //     the LDS-only barriers do not wait for them)
        if (is_ent && !(a.dbg & 2)) {
          float vk[16];
          int ek[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int row = rg + 4 * k;
            vk[k] = dxt[row * LDA + col];
            ek[k] = ids[(row * T + t) * 4 + 1];
          }
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int row = rg + 4 * k;
            if (ldk[k] == row) unsafeAtomicAdd(a.gWe + (int64_t)ek[k] * a.de + (col - e0), vk[k]);
          }
        }
      }
      lds_barrier();  // dhr / staged tiles visible to step t-1; dA_t, dxt, lead free for reuse
      TPROBE(5)  // scatter + end barrier
    };
    for (int t = T - 1; t > 0; --t) step(std::true_type{}, t);
    step(std::false_type{}, 0);
  }

  // ---- flush the launch-persistent accumulators: plain coalesced stores into this workgroup's slab;
  // k_reduce_partials sums the slabs (device-scope atomics from 256 workgroups onto the same 128 KB
  // were measured slower: they execute at the memory side, the per-XCD L2s are not coherent)
  {
    float* pw = a.part + (int64_t)blockIdx.x * PART;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t grow = (int64_t)q * DH + j * 16 + ag * 4 + r;
          pw[grow * DH + nt * 16 + arow] = dwi[q][nt][r];
          pw[256 * 64 + grow * DH + nt * 16 + arow] = dwo[q][nt][r];
        }
      }
      float v = dbias[q];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (ag == 0) pw[2 * 256 * 64 + q * DH + j * 16 + arow] = v;
    }
  }
  if constexpr (mscat) if (wcls != 1) {
    // acc_s[r] <-> table row 4ag + r, column 16j + arow of the type (wcls 0) / relation (wcls 2) gradient
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int v = ag * 4 + r;
      const int col = j * 16 + arow;
      if (wcls == 0) { if (v < a.Vt) unsafeAtomicAdd(a.gWt + (int64_t)v * a.dt + col, acc_s[r]); }
      else { if (v < a.Vr) unsafeAtomicAdd(a.gWr + (int64_t)v * a.dr + (col - a.dt - a.de), acc_s[r]); }
    }
  }
  if constexpr (!mscat) if (small_in_lds) {
    lds_barrier();
    const int nt_small = a.Vt * a.dt;
    for (int i = tid; i < n_small; i += 256) {
      const float v = small_g[i];
      if (v != 0.f) { if (i < nt_small) unsafeAtomicAdd(a.gWt + i, v); else unsafeAtomicAdd(a.gWr + (i - nt_small), v); }
    }
  }
  TPROBE(6)  // flush
  if (a.timing && tid == 0) {
    for (int k = 0; k < 8; ++k) a.timing[(int64_t)blockIdx.x * 8 + k] = tacc[k];
  }
}

// gW_i2g / gW_o2g / gb += sum over workgroup slabs
__global__ void k_reduce_partials(const float* __restrict__ part, int nslab, float* __restrict__ gWi, float* __restrict__ gWo, float* __restrict__ gbi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= PART) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = blockIdx.y * 4; s < nslab; s += gridDim.y * 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) if (s + u < nslab) acc[u] += part[(int64_t)(s + u) * PART + i];
  }
  const float v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  if (i < 256 * 64) unsafeAtomicAdd(gWi + i, v);
  else if (i < 2 * 256 * 64) unsafeAtomicAdd(gWo + (i - 256 * 64), v);
  else unsafeAtomicAdd(gbi + (i - 2 * 256 * 64), v);
}

// WT[n][k] = W[k][n] for a [256][64] weight
__global__ void k_transpose_256x64(const float* __restrict__ W, float* __restrict__ WT) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over 64*256 outputs
  if (i >= 64 * 256) return;
  const int n = i >> 8, k = i & 255;
  WT[i] = W[k * 64 + n];
}

// ---- host side ----
struct State {
  float* save_frag = nullptr;
  float* save_h = nullptr;
  int64_t cap_N = 0;
  int cap_T = 0;
  int num_cu = 0;
  float* WT = nullptr;      // [L][2][64][256]
  bool wt_dirty = true;
  float* dHhead = nullptr;  // [N][64]
  float* DX = nullptr;      // [T][N][64]
  float* part = nullptr;    // [num_cu][PART]
  unsigned long long* timing = nullptr;  // [num_cu][8] when KPRN_TIMING=1
  int64_t cap_Nb = 0; int cap_Tb = 0;
};

static State* st(kprn_handle* h) {
  if (!h->fused_state) {
    State* s = new State();
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, h->cfg.device_id) == hipSuccess) s->num_cu = p.multiProcessorCount;
    if (s->num_cu <= 0) s->num_cu = 256;
    h->fused_state = s;
  }
  return (State*)h->fused_state;
}

bool fwd_supported(const kprn_handle* h, int T) {
  const kprn_config& c = h->cfg;
  return (h->D == DH && c.H == DH && c.L >= 1 && c.L <= 2 && (c.dt % 4) == 0 && (c.de % 4) == 0 && (c.dr % 4) == 0 && T >= 2 && T <= MAXT_LDS);
}

bool bwd_supported(const kprn_handle* h, int T) { return fwd_supported(h, T); }

template <int L, bool SAVE>
static void launch_fwd(kprn_handle* h, const FwdArgs& a, int grid) {
  const size_t lds_bytes = (size_t)(2 + 2 * L) * MT * LDA * sizeof(float) + 2 * MT * MAXT_LDS * 4 * sizeof(int32_t);
  static bool attr_done = false;  // one per template instantiation
  if (!attr_done) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_lstm_fwd<L, SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_done = true;
  }
  hipLaunchKernelGGL((k_lstm_fwd<L, SAVE>), dim3(grid), dim3(256), lds_bytes, h->stream, a);
  HIP_TRY(hipGetLastError());
}

void forward(kprn_handle* h, const kprn_batch* b, bool save) {
  const kprn_config& c = h->cfg;
  State* s = st(h);
  const int64_t N = (int64_t)b->B * b->P;
  FwdArgs a;
  a.idx = b->idx; a.N = N; a.T = b->T; a.F = b->F; a.nT = c.num_types;
  a.Wt = h->dense + h->off_Wt; a.We = h->We; a.Wr = h->dense + h->off_Wr;
  a.dt = c.dt; a.de = c.de; a.dr = c.dr;
  for (int l = 0; l < 2; ++l) {
    const int ll = l < c.L ? l : 0;
    a.Wi[l] = h->dense + h->layer[ll].Wi; a.bi[l] = h->dense + h->layer[ll].bi; a.Wo[l] = h->dense + h->layer[ll].Wo;
  }
  a.Wout = h->dense + h->off_outW; a.bout = h->dense + h->off_outb; a.C = c.C;
  a.S = h->ws.S;
  a.n_tiles = (N + MT - 1) / MT;
  a.save_frag = nullptr; a.save_h = nullptr;
  if (save) {
    if (N > s->cap_N || b->T > s->cap_T) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (s->save_frag) hipFree(s->save_frag);
      if (s->save_h) hipFree(s->save_h);
      const int64_t cn = std::max<int64_t>(N, s->cap_N);
      const int ct = std::max(b->T, s->cap_T);
      const int64_t mts = (cn + 15) / 16 + 4;
      HIP_TRY(hipMalloc((void**)&s->save_frag, (size_t)mts * ct * c.L * 4 * NPL * 256 * sizeof(float)));
      HIP_TRY(hipMalloc((void**)&s->save_h, (size_t)ct * c.L * (cn + 64) * DH * sizeof(float)));
      s->cap_N = cn; s->cap_T = ct;
    }
    a.save_frag = s->save_frag; a.save_h = s->save_h;
  }
  const int grid = (int)std::min<int64_t>(a.n_tiles, (int64_t)s->num_cu);
  ProfScope ps(h, save ? "lstm_fused_fwd_train" : "lstm_fused_fwd");
  if (c.L == 1) { if (save) launch_fwd<1, true>(h, a, grid); else launch_fwd<1, false>(h, a, grid); }
  else { if (save) launch_fwd<2, true>(h, a, grid); else launch_fwd<2, false>(h, a, grid); }
}

template <bool BOTTOM, bool TOP, bool MSCAT>
static void launch_bwd(kprn_handle* h, const BwdArgs& a, int grid) {
  const int n_small = BOTTOM ? (a.Vt * a.dt + a.Vr * a.dr) : 0;
  const size_t lds_bytes = (size_t)(3 * MT * LDA + MT * LDD) * sizeof(float) + (MT * MAXT_LDS * 4 + MT) * sizeof(int32_t) +
                           (size_t)(BOTTOM ? MT * LDA : 0) * sizeof(float) + (size_t)(n_small <= 4096 ? n_small : 0) * sizeof(float);
  HIP_TRY(hipFuncSetAttribute((const void*)k_lstm_bwd<BOTTOM, TOP, MSCAT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL((k_lstm_bwd<BOTTOM, TOP, MSCAT>), dim3(grid), dim3(256), lds_bytes, h->stream, a);
  HIP_TRY(hipGetLastError());
}

// needs: forward(save=true) of the same batch just ran; ws.dS holds d loss / d S[:, cid]
void backward(kprn_handle* h, const kprn_batch* b, int cid) {
  const kprn_config& c = h->cfg;
  State* s = st(h);
  const int64_t N = (int64_t)b->B * b->P;
  const int T = b->T, L = c.L;
  hipStream_t strm = h->stream;
  if (N > s->cap_Nb || T > s->cap_Tb) {
    HIP_TRY(hipStreamSynchronize(strm));
    if (s->dHhead) hipFree(s->dHhead);
    if (s->DX) hipFree(s->DX);
    const int64_t cn = std::max<int64_t>(N, s->cap_Nb);
    const int ct = std::max(T, s->cap_Tb);
    HIP_TRY(hipMalloc((void**)&s->dHhead, (size_t)(cn + 64) * DH * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&s->DX, (size_t)ct * (cn + 2 * MT) * DH * sizeof(float)));
    s->cap_Nb = cn; s->cap_Tb = ct;
  }
  if (!s->WT) HIP_TRY(hipMalloc((void**)&s->WT, (size_t)2 * 2 * 64 * 256 * sizeof(float)));
  if (!s->part) HIP_TRY(hipMalloc((void**)&s->part, (size_t)s->num_cu * PART * sizeof(float)));
  static const bool want_timing = getenv("KPRN_TIMING") != nullptr;
  if (want_timing && !s->timing) HIP_TRY(hipMalloc((void**)&s->timing, (size_t)s->num_cu * 8 * sizeof(unsigned long long)));
  if (s->wt_dirty) {
    ProfScope ps(h, "weight_transpose");
    for (int l = 0; l < L; ++l) {
      hipLaunchKernelGGL(k_transpose_256x64, dim3(64), dim3(256), 0, strm, h->dense + h->layer[l].Wi, s->WT + (size_t)(l * 2 + 0) * 64 * 256);
      hipLaunchKernelGGL(k_transpose_256x64, dim3(64), dim3(256), 0, strm, h->dense + h->layer[l].Wo, s->WT + (size_t)(l * 2 + 1) * 64 * 256);
    }
    HIP_TRY(hipGetLastError());
    s->wt_dirty = false;
  }
  float* gd = h->g_dense;
  {
    ProfScope ps(h, "head_bwd");
    const float* hT = s->save_h + ((int64_t)(T - 1) * L + (L - 1)) * N * DH;
    kk::head_bwd(strm, h->ws.dS, hT, h->dense + h->off_outW, N, DH, cid, s->dHhead, gd + h->off_outW, gd + h->off_outb);
  }
  const int64_t n_tiles = (N + MT - 1) / MT;
  const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)s->num_cu);
  for (int l = L - 1; l >= 0; --l) {
    BwdArgs a;
    a.idx = b->idx; a.N = N; a.T = T; a.F = b->F; a.nT = c.num_types;
    a.Wt = h->dense + h->off_Wt; a.We = h->We; a.Wr = h->dense + h->off_Wr;
    a.dt = c.dt; a.de = c.de; a.dr = c.dr; a.Vt = c.Vt; a.Vr = c.Vr;
    a.L = L; a.layer = l;
    a.WiT = s->WT + (size_t)(l * 2 + 0) * 64 * 256; a.WoT = s->WT + (size_t)(l * 2 + 1) * 64 * 256;
    a.save_frag = s->save_frag; a.save_h = s->save_h; a.dHhead = s->dHhead; a.DX = s->DX; a.Npad = n_tiles * MT;
    a.gWi = gd + h->layer[l].Wi; a.gbi = gd + h->layer[l].bi; a.gWo = gd + h->layer[l].Wo;
    a.gWt = gd + h->off_Wt; a.gWe = h->g_We; a.gWr = gd + h->off_Wr;
    a.n_tiles = n_tiles;
    a.part = s->part; a.timing = s->timing;
    { static const char* d = getenv("KPRN_DBG"); a.dbg = d ? atoi(d) : 0; }
    a.lead = b->lead;
    a.mfma_scatter = (b->lead && c.num_types == 1 && (c.dt % 16) == 0 && (c.de % 16) == 0 && (c.dr % 16) == 0 && c.Vt <= 16 && c.Vr <= 16 &&
                      !(a.dbg & 8)) ? 1 : 0;
    const bool bottom = (l == 0), top = (l == L - 1);
    {
      ProfScope ps(h, "lstm_fused_bwd");
      if (bottom && top) { if (a.mfma_scatter) launch_bwd<true, true, true>(h, a, grid); else launch_bwd<true, true, false>(h, a, grid); }
      else if (bottom) { if (a.mfma_scatter) launch_bwd<true, false, true>(h, a, grid); else launch_bwd<true, false, false>(h, a, grid); }
      else if (top) launch_bwd<false, true, false>(h, a, grid);
      else launch_bwd<false, false, false>(h, a, grid);
    }
    {
      ProfScope ps(h, "dw_reduce");
      hipLaunchKernelGGL(k_reduce_partials, dim3((PART + 255) / 256, 16), dim3(256), 0, strm, s->part, grid, a.gWi, a.gWo, a.gbi);
      HIP_TRY(hipGetLastError());
    }
    if (s->timing) {
      HIP_TRY(hipStreamSynchronize(strm));
      std::vector<unsigned long long> tb((size_t)grid * 8);
      HIP_TRY(hipMemcpy(tb.data(), s->timing, tb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double sum[8] = {0};
      for (int g = 0; g < grid; ++g) for (int k = 0; k < 8; ++k) sum[k] += (double)tb[(size_t)g * 8 + k];
      fprintf(stderr, "[kprn timing] bwd layer %d N=%lld grid=%d avg cycles/WG: prologue %.0f prefetch %.0f stageC %.0f midbar %.0f stageE %.0f scatter+bar %.0f flush %.0f\n",
              l, (long long)N, grid, sum[0] / grid, sum[1] / grid, sum[2] / grid, sum[3] / grid, sum[4] / grid, sum[5] / grid, sum[6] / grid);
    }
  }
}

void params_changed(kprn_handle* h) { if (h->fused_state) ((State*)h->fused_state)->wt_dirty = true; }

void release(kprn_handle* h) {
  State* s = (State*)h->fused_state;
  if (!s) return;
  for (float* p : {s->save_frag, s->save_h, s->WT, s->dHhead, s->DX, s->part}) if (p) hipFree(p);
  if (s->timing) hipFree(s->timing);
  delete s;
  h->fused_state = nullptr;
}

}  // namespace fused
