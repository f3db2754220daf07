// Shared device helpers + host-side state of the fused LSTM path kernels (gfx950, D = H = 64, L <= 2).
// Forward: lstm_fused_fwd.hip, backward: lstm_fused_bwd.hip.
#pragma once
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

// time-split tile hand-over (described further down, at ho_plan)
struct HoArgs {
  float* state = nullptr;      // [gridDim.x][HO_STATE] hand-over slots, indexed by the HEAVY workgroup of a pair
  unsigned* flag = nullptr;    // [gridDim.x] epoch of the slot's contents
  unsigned epoch = 0;          // this launch's serial; 0: no hand-over (every workgroup runs whole tiles)
  int mode = 1;                // pairing: 1 = b with G - 1 - b, 2 = b with b + G / 2
  int* fault = nullptr;        // host-visible: set when a wait timed out
};

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
  const int32_t* perm;    // identical-prefix plan (nullable): tile slot n holds original path perm[n] (S is written in the original order)
  const int32_t* tile_k;  // [n_tiles] leading steps of the tile that are replaced by the prefix state (nullable: 0)
  const int32_t* pmeta;   // plan header: [0] longest prefix in the batch (nullable with tile_k)
  const float* pfb;       // [KCAP+1][L][PFB] per prefix class: W_o2g h_prefix | c_prefix   (class 0: zeros)
  float* save_frag;  // training: [(N/16)][T][L][4 waves][NPL][64 lanes][4]   (nullable)
  int64_t n_tiles;
  unsigned long long* timing;  // optional [grid][8] cycle counters (KPRN_TIMING=1)
  HoArgs ho;         // time-split tile hand-over (below)
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

// Identical-prefix skipping (batch_index.hip prefix_plan, lstm_fused_prefix.hip): a tile whose 64 paths all start with
// k copies of the batch's reference step starts at step k from the state the prefix kernel computed once.
constexpr int R1 = 2 * 4 * DH + 2 * DH;  // floats per (layer, prefix step) of the rank-1 term buffer
constexpr int PFB = 4 * DH + DH; // floats per (class, layer) of the prefix table: recurrent half of the first step [256] | c[64]

// all the tile's ids -> LDS: ids[(row*T + t)*4 + {0: first type, 1: entity, 2: relation}] (0-based).
// Rows past N repeat row N-1 (their results are never stored).  Removes the dependent id -> row load
// chain from every step's gather.
// MTR: path rows of a tile -- MT (64), or 16 in the small-batch instantiations (one 16-row m-tile per tile: lstm_fused_fwd.hip NMT)
template <int NTHREADS, int MTR = MT>
__device__ __forceinline__ void ids_stage(const int32_t* idx, int64_t N, int T, int F, int nT, int64_t tile, int32_t* ids) {
  for (int c = threadIdx.x; c < MTR * T; c += NTHREADS) {
    const int row = c / T, t = c - row * T;
    int64_t n = tile * MTR + row;
    if (n >= N) n = N - 1;
    const int32_t* f = idx + (n * T + t) * F;
    ids[c * 4 + 0] = f[F - nT - 2] - 1;
    ids[c * 4 + 1] = f[F - 2] - 1;
    ids[c * 4 + 2] = f[F - 1] - 1;
  }
}

// Per-thread source of the x-row gather.  A thread always serves the same 16-byte chunk column (ch = tid & 15)
// of rows (tid >> 4) + 16k, so the table it reads from is fixed for the whole launch: no branches per load.
struct GatherSrc {
  const float* base;  // table + chunk offset inside the row
  int width;          // table row stride (floats)
  int slot;           // which id of the LDS id tile: 0 type, 1 entity, 2 relation
};
template <class Args>
__device__ __forceinline__ GatherSrc gather_src(const Args& a) {
  const int ch = threadIdx.x & 15;
  const int c_t = a.dt >> 2, c_e = (a.dt + a.de) >> 2;
  GatherSrc g;
  if (ch < c_t) { g.base = a.Wt + ch * 4; g.width = a.dt; g.slot = 0; }
  else if (ch < c_e) { g.base = a.We + (ch - c_t) * 4; g.width = a.de; g.slot = 1; }
  else { g.base = a.Wr + (ch - c_e) * 4; g.width = a.dr; g.slot = 2; }
  return g;
}

// gather this thread's share of one step's x rows for `tile` into registers (ids from the LDS id tile)
template <int NTHREADS, int MTR = MT, class Args>
// t indexes the LDS id tile handed in; tg = the step's index in the batch (differs when `ids` points behind a skipped prefix)
__device__ __forceinline__ void gather_load(const Args& a, const GatherSrc& g, int64_t tile, int t, const int32_t* ids, f32x4 (&v)[MTR * 16 / NTHREADS],
                                            int tg = -1) {
  constexpr int PER = MTR * 16 / NTHREADS;  // MTR rows x 16 float4 chunks
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int row = (threadIdx.x >> 4) + k * (NTHREADS >> 4);
    const int id = ids[(row * a.T + t) * 4 + g.slot];
    v[k] = *(const f32x4*)(g.base + (int64_t)id * g.width);
  }
  if (a.nT > 1 && g.slot == 0) {  // several type slots per step: CAddTable over them (FeatureEmbedding.lua:55)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int row = (threadIdx.x >> 4) + k * (NTHREADS >> 4);
      int64_t n = tile * MTR + row;
      if (n >= a.N) n = a.N - 1;
      const int32_t* f = a.idx + (n * a.T + (tg >= 0 ? tg : t)) * a.F;
      for (int q = 1; q < a.nT; ++q) v[k] += *(const f32x4*)(g.base + (int64_t)(f[a.F - a.nT - 2 + q] - 1) * g.width);
    }
  }
}

// ---- the same through LDS-DMA (round 6: the fused forward) ------------------------------------------------
// ids_stage loads a tile's ids into registers, converts and writes them to LDS: a load round trip in front of every tile's first slot (2.5 k cycles a
// tile, per-phase counters) and a division by T per entry.  Here the ids go global -> LDS without passing a register (global_load_lds_dword: lane l's
// dword lands at dst + 4 l), RAW (1-based: the gather's table base is shifted by one row instead), in three planes [kind][t][row] -- a wave instruction
// is one step's 64 rows (or four steps' 16), no division -- and nothing waits for them: a wave's pieces have landed when its next counted vmcnt wait has
// passed (gather_store, or the explicit wait in the prologue), the other waves see them behind the barrier after that.
template <int MTR = MT> constexpr int IDS_PLANE = MAXT_LDS * MTR;   // ints per plane (3 planes fit the [MT][MAXT_LDS][4] buffer of ids_stage)
__device__ __forceinline__ unsigned lds_offset_of(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }
template <int NTHREADS, int MTR = MT>
__device__ __forceinline__ void ids_stage_dma(const int32_t* idx, int64_t N, int T, int F, int nT, int64_t tile, int32_t* ids) {
  static_assert(NTHREADS == 256 && (MTR == 64 || MTR == 16), "4 waves; a wave instruction covers 64 / MTR steps");
  constexpr int PER = 64 / MTR;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int row = lane % MTR;
  int64_t n = tile * MTR + row;
  if (n >= N) n = N - 1;
  const unsigned base = lds_offset_of(ids);
  for (int tb = wave * PER; tb < T; tb += 4 * PER) {
    int t = tb + lane / MTR;
    if (t >= T) t = T - 1;   // (lanes past T fill plane entries nothing reads: MAXT_LDS is a multiple of PER)
    const int32_t* f = idx + (n * T + t) * F;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int32_t* src = f + (k == 0 ? F - nT - 2 : (k == 1 ? F - 2 : F - 1));
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(base + (unsigned)(k * IDS_PLANE<MTR> + tb * MTR) * 4u));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
  }
}
// (its gather: ids from the planes, the table base one row down)
template <int NTHREADS, int MTR = MT, class Args>
__device__ __forceinline__ void gather_load_planes(const Args& a, const GatherSrc& g, int64_t tile, int t, const int32_t* ids, f32x4 (&v)[MTR * 16 / NTHREADS]) {
  constexpr int PER = MTR * 16 / NTHREADS;
  const int32_t* p = ids + g.slot * IDS_PLANE<MTR> + t * MTR + (threadIdx.x >> 4);
  const float* base1 = g.base - g.width;
#pragma unroll
  for (int k = 0; k < PER; ++k) v[k] = *(const f32x4*)(base1 + (int64_t)p[k * (NTHREADS >> 4)] * g.width);
  if (a.nT > 1 && g.slot == 0) {  // several type slots per step: CAddTable over them (FeatureEmbedding.lua:55)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int row = (threadIdx.x >> 4) + k * (NTHREADS >> 4);
      int64_t n = tile * MTR + row;
      if (n >= a.N) n = a.N - 1;
      const int32_t* f = a.idx + (n * a.T + t) * a.F;
      for (int q = 1; q < a.nT; ++q) v[k] += *(const f32x4*)(base1 + (int64_t)f[a.F - a.nT - 2 + q] * g.width);
    }
  }
}

template <int NTHREADS, int MTR = MT>
__device__ __forceinline__ void gather_store(float* xbuf, const f32x4 (&v)[MTR * 16 / NTHREADS]) {
  constexpr int PER = MTR * 16 / NTHREADS;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int c = threadIdx.x + k * NTHREADS;
    const int row = c >> 4, ch = c & 15;
    *(f32x4*)(xbuf + row * LDA + ch * 4) = v[k];
  }
}

// Training saves, per (16-row m-tile, t, layer, wave): NPL planes of 1 KiB in MFMA C-fragment order (lane-major
// float4: lane (ag, arow), register r <-> row 4 ag + r, hidden col 16 wave + arow).  The forward stores the
// factors the backward multiplies by, not the raw gates, so the cell backward is 7 VALU ops per element:
//   P1 = g i (1-i)   P2 = i (1-g^2)   P3 = c_{t-1} f (1-f)   P4 = tanh(c) o (1-o)   P5 = o (1-tanh(c)^2)   P6 = f
//   dC = dc + dh P5;  dA_i = dC P1;  dA_g = dC P2;  dA_f = dC P3;  dA_o = dh P4;  dc' = dC P6
// plane 6 = h: the B operand of dW = dA^T [x | h] in exactly the register layout the MFMA wants.
constexpr int NPL = 7;
// Explicitly GLOBAL pointers for the training forward's plane stores: a tile's region is wave-uniform, its address is rebuilt from two scalar registers
// (readfirstlane) and everything below it fits a 32-bit offset -- the stores are then `global_store voffset, data, sbase` and the per-unit 64-bit VALU
// multiplies by run-time strides are gone (training forward 783 k -> 774 k cycles per workgroup).  A pointer rebuilt from integers without the address space
// would be a FLAT one (flat_store).  The same form on the BACKWARD's plane loads was measured and taken out again: stage C's MFMA phase went from 299 k to
// 358 k cycles per workgroup (each load's offset then comes from a v_add right in front of it, inside the dW MFMA stream; profiles/r06 README, call r7o).
typedef __attribute__((address_space(1))) char gchar;
typedef __attribute__((address_space(1))) f32x4 gf32x4;
__device__ __forceinline__ gchar* uniform_global(const void* p) {
  const uint64_t v = (uint64_t)(size_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return (gchar*)(((uint64_t)hi << 32) | lo);
}

// ---- MFMA issue, hand-placed ---------------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 as inline asm with the register FILES chosen here.  Values that only ever feed the
// matrix pipe (register-stationary weights as B operand, the launch-persistent dW accumulators) live in the
// accumulation half of the unified register file ("a"); everything the VALU touches lives in architectural
// VGPRs ("v").  Left to hipcc, such values were parked in AGPRs but staged through v_accvgpr_read/write around
// most MFMAs; a staging copy that feeds the very next MFMA costs +16 cycles per 32-cycle MFMA
// (scripts/ubench/mfma_rate.hip: 48.0 vs 32.07 ticks/MFMA).
// The asm MFMAs are opaque to hipcc's hazard recogniser: every place where an MFMA RESULT is consumed by a
// non-MFMA instruction (or as an A/B operand) less than ~20 issue slots later carries an explicit KPRN_MFMA_DRAIN.
#define KPRN_MFMA(ACC, A_, B_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A_), "a"(B_))
// first MFMA of an accumulation chain, srcC = another register (bias vector)
#define KPRN_MFMA_C(ACC, A_, B_, C_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %3" : "=&v"(ACC) : "v"(A_), "a"(B_), "v"(C_))
// first MFMA of an accumulation chain, srcC = 0
#define KPRN_MFMA_Z(ACC, A_, B_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(ACC) : "v"(A_), "a"(B_))
// accumulator in AGPRs, both operands in VGPRs (dW += dA^T [x | h])
#define KPRN_MFMA_ACC_A(ACC, A_, B_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(ACC) : "v"(A_), "v"(B_))
// everything in VGPRs; the A operand is a VALU result of the instruction right before (one-hot scatter): a VALU
// write needs 2 wait states before an MFMA may read it, which hipcc inserts for its own MFMAs only
#define KPRN_MFMA_VV(ACC, A_, B_) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A_), "v"(B_))
#define KPRN_MFMA_VVZ(ACC, A_, B_) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(ACC) : "v"(A_), "v"(B_))
#define KPRN_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")
// A drain orders nothing for hipcc's scheduler by itself: an operand-less asm statement has no data dependency on the values it
// protects, and a VALU read of an MFMA result may be scheduled above it (seen on the matrix-core forward: the first two
// registers of the last cell read accumulators that had not landed).  Every drain is therefore followed by a PIN of the
// accumulators about to be read: an empty asm that takes them as read-write operands.  asm volatile statements keep their
// order (MFMAs -> drain -> pin), and every later use of the values depends on the pin.
#define KPRN_PIN_V4(A) asm volatile("" : "+v"((A)[0]), "+v"((A)[1]), "+v"((A)[2]), "+v"((A)[3]))
#define KPRN_PIN_V2(X, Y) asm volatile("" : "+v"(X), "+v"(Y))
#define KPRN_PIN_A8(A, B) \
  asm volatile("" : "+a"((A)[0]), "+a"((A)[1]), "+a"((A)[2]), "+a"((A)[3]), "+a"((B)[0]), "+a"((B)[1]), "+a"((B)[2]), "+a"((B)[3]))

// ---- time-split tile hand-over ----------------------------------------------------------
// The persistent kernels deal tiles round-robin (tile = blockIdx.x + i gridDim.x) and a tile's time is its executed steps, so with a
// left-padded path set (tiles of T and T - 2 steps) the workgroups' sums differ by whole steps: 21 of 256 draw 20 tile-steps, the others
// 18, and the launch lasts 20.  A tile is therefore allowed to change workgroups once, BETWEEN two of its steps: the recurrent state of
// its 64 rows ((h, c) per layer in the forward, (dh, dc) in the backward) goes through a per-tile slot in global memory.  Workgroup b is
// paired with workgroup G - 1 - b (tiles are sorted longest first, so loads fall with b: the pairing is heaviest with lightest); when
// their loads differ by at least one full step, the LAST d steps (in the kernel's own time order) of the heavy one's first tile move:
//   forward:  heavy runs steps k0 .. T-1-d of the tile FIRST and publishes (h, c); light resumes it at T-d as its LAST piece of work;
//   backward: light runs steps T-1 .. T-d of the tile FIRST and publishes (dh, dc); heavy resumes it at T-d-1 as its last piece.
// The piece that publishes runs at the very start of its workgroup, the piece that waits at the very end of the other: the wait is a
// formality (and bounded: a time-out raises the fault word instead of hanging the queue).  Both sides evaluate the same closed-form
// rule from tile_k -- no schedule array, nothing for the batch planners to build.  Cost unit: half a step (a tile's first executed step
// in the forward / last one in the backward has no recurrent half).
constexpr int HO_STATE = 4 * MT * DH;   // floats per slot: forward h [L][64][64] + c [L][NMT][256][4]; backward dh | dc [NMT][256][4] each
struct HoPlan { int role; int d; int64_t tile; int slot; };   // role: 0 whole tiles only, 1 heavy, 2 light; tile: the one that changes hands; d: steps moved
constexpr long long HO_TIMEOUT_TICKS = 200000000ll;   // 2 s of the 100 MHz wall clock

// the pair of workgroup bx among G: mode 1 = G - 1 - bx (heaviest with lightest), 2 = bx +- G / 2 (the heavy one is still the lower index, and the highest
// indices -- the workgroups the hardware places last when the launch starts under another kernel's tail -- keep their slack); bx itself: unpaired
__host__ __device__ __forceinline__ int ho_partner(int mode, int G, int bx) {
  const int half = G >> 1;
  return (mode == 2) ? (bx < half ? bx + half : (bx - half < half ? bx - half : bx)) : G - 1 - bx;
}
// what a pair with these loads (half steps) does; steps: executed steps of the heavy one's first tile; keep: steps of that tile that must stay in front of
// the moved ones (forward: 2 -- the ids of a workgroup's next tile are staged under a tile's first slot and read under its second; backward: 1)
// heavy_tiles: tiles of the heavy workgroup -- with a single one it has nothing to run while the other workgroup works on the piece it waits for
__host__ __device__ __forceinline__ HoPlan ho_decide(int bx, int px, int mine, int theirs, int steps, int keep, int64_t heavy_tiles) {
  HoPlan p{0, 0, -1, 0};
  const int diff = mine > theirs ? mine - theirs : theirs - mine;
  if (px == bx || diff < 4 || heavy_tiles < 2) return p;
  p.role = mine > theirs ? 1 : 2;
  p.tile = mine > theirs ? bx : px;
  p.slot = (int)p.tile;
  // d minimises max(heavy - 2 d, light + 2 d)
  int d = diff >> 2;
  if (2 * (d + 1) - diff < -2 * d) ++d;
  p.d = d < steps - keep ? d : steps - keep;
  if (p.d < 1) p.role = 0;
  return p;
}
// host form (kprn_batch_handover_stats evaluates the rule for every workgroup): tile_k0(t) = prefix length of tile t
template <class K0>
inline HoPlan ho_plan_host(bool on, int mode, int G, int bx, int64_t n_tiles, int T, K0 tile_k0, int* load, int keep = 1) {
  const int px = ho_partner(mode, G, bx);
  int mine = 0, theirs = 0;
  for (int64_t t = bx; t < n_tiles; t += G) mine += 2 * (T - tile_k0(t)) - 1;
  for (int64_t t = px; t < n_tiles; t += G) theirs += 2 * (T - tile_k0(t)) - 1;
  if (load) *load = mine;
  if (!on || (n_tiles + G - 1) / G > 32) return HoPlan{0, 0, -1, 0};
  const int64_t heavy = mine > theirs ? bx : px;
  return ho_decide(bx, px, mine, theirs, T - tile_k0(heavy), keep, (n_tiles - 1 - heavy) / G + 1);
}
// device form: ONE round trip -- lane i < 32 of every wave reads the prefix length of this workgroup's i-th tile, lane 32 + i of the pair's, and the two sums
// are wave reductions.  (Read tile after tile it was eight dependent loads in front of every workgroup's first tile: 4 us, 1.5 % of the launch.)
__device__ __forceinline__ HoPlan ho_plan(const HoArgs& ho, const int32_t* tile_k, int64_t n_tiles, int T, int keep, const int bx, const int G) {
  const int px = ho_partner(ho.mode, G, bx);
  if (!ho.epoch || px == bx || (n_tiles + G - 1) / G > 32) return HoPlan{0, 0, -1, 0};
  const int lane = threadIdx.x & 63, i = lane & 31;
  const int64_t t = (int64_t)(lane < 32 ? bx : px) + (int64_t)i * G;
  const int k = (tile_k && t < n_tiles) ? tile_k[t] : 0;
  int cost = t < n_tiles ? 2 * (T - k) - 1 : 0;
#pragma unroll
  for (int sh = 1; sh < 32; sh <<= 1) cost += __shfl_xor(cost, sh, 64);
  const int mine = __builtin_amdgcn_readlane(cost, 0), theirs = __builtin_amdgcn_readlane(cost, 32);
  const int k_mine = __builtin_amdgcn_readlane(k, 0), k_theirs = __builtin_amdgcn_readlane(k, 32);
  return ho_decide(bx, px, mine, theirs, T - (mine > theirs ? k_mine : k_theirs), keep, (n_tiles - 1 - (mine > theirs ? bx : px)) / G + 1);
}

// slot traffic at agent scope (sc1: written through to / read from where the eight XCDs' L2s agree), 16 bytes per call
__device__ __forceinline__ void ho_store4(float* p, const f32x4& v) {
  union { f32x4 f; unsigned long long u[2]; } c;
  c.f = v;
  __hip_atomic_store((unsigned long long*)p, c.u[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store((unsigned long long*)p + 1, c.u[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 ho_load4(const float* p) {
  union { f32x4 f; unsigned long long u[2]; } c;
  c.u[0] = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  c.u[1] = __hip_atomic_load((const unsigned long long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return c.f;
}
// ... 4 bytes per call (the forward's cell state: its registers are scalars of the hot loop -- handled as 16-byte tuples here they constrained that loop's
// allocation: 140 spilled registers)
__device__ __forceinline__ void ho_store1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ho_load1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every thread's slot stores have been acknowledged -> the flag
__device__ __forceinline__ void ho_publish(const HoArgs& ho, int slot) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(ho.flag + slot, ho.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ho_wait(const HoArgs& ho, int slot) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(ho.flag + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ho.epoch) {
    __builtin_amdgcn_s_sleep(16);
    if (wall_clock64() - t0 > HO_TIMEOUT_TICKS) {
      if (threadIdx.x == 0) __hip_atomic_store(ho.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
  asm volatile("" ::: "memory");   // the slot's loads stay behind the flag's
}

// ---- host-side state shared by forward() and backward() ----
struct State {
  float* save_frag = nullptr;
  unsigned* pipe_flag = nullptr;   // layer pipeline of the small-batch BPTT launch: [num_cu][MAXT_LDS] epochs
  int64_t cap_N = 0;
  int cap_T = 0;
  int num_cu = 0;
  float* WT = nullptr;      // [L][2][64][256]
  bool wt_dirty = true;
  float* DX = nullptr;      // [T][N][64]
  float* DXe = nullptr;     // compact entity slice of the bottom layer's dx: [(Npad T + KCAP)][de]
  bool DXe_on = false;      // the backward in progress writes / reads it
  // identical-prefix state (lstm_fused_prefix.hip)
  float* pfb = nullptr;     // [KCAP+1][L][PFB] W_o2g h_prefix | c_prefix per prefix class
  float* pfs = nullptr;     // [KCAP][L][NPL][64] backward factors + h of the prefix steps
  float* pfx = nullptr;     // [64] the reference step's input row
  float* PG = nullptr;      // [L][KCAP+1][PFB] per class: sum of dA at the first executed step | sum of dc handed to the prefix
  float* r1 = nullptr;      // [L][KCAP][R1] rank-1 weight-gradient terms of the prefix steps: dA_{t+1}[256] | dA_t[256] | h_t[64] | in_t[64]
  // matrix-core forward (lstm_fused_fwd_mc.hip): split weights in register order, scaled biases, h hand-over between layers
  void* mc_wsp = nullptr; float* mc_bias = nullptr; bool mc_dirty = true; int mc_ns = 0;
  float* mc_hseq[2] = {nullptr, nullptr}; int64_t mc_capN[2] = {0, 0}; int mc_capT[2] = {0, 0};  // [stream: main, scoring]
  int64_t pf_batch = -1;    // batch serial the table was computed for (-1: stale)

  float* part = nullptr;    // [L][num_cu][PART]
  // time-split tile hand-over: slots + flags per launch context (0: the engine's main stream, 1: the scoring stream -- the two run side by side),
  // one launch serial for all of them, the fault word in page-locked host memory
  float* ho_state[2] = {nullptr, nullptr}; unsigned* ho_flag[2] = {nullptr, nullptr};
  unsigned ho_epoch = 0;
  float* part_small = nullptr;  // [8 num_cu][Vt*dt + Vr*dr] small-table partials of the embedding scatter
  int part_small_n = 0;
  unsigned long long* timing = nullptr;  // [num_cu][8] when KPRN_TIMING=1
  int64_t cap_Nb = 0; int cap_Tb = 0;
};

static inline State* st(kprn_handle* h) {
  if (!h->fused_state) {
    State* s = new State();
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, h->cfg.device_id) == hipSuccess) s->num_cu = p.multiProcessorCount;
    if (s->num_cu <= 0) s->num_cu = 256;
    h->fused_state = s;
  }
  return (State*)h->fused_state;
}

bool fwd_supported(const kprn_handle* h, int T);
// small batches: tiles of ONE 16-row m-tile (four times as many workgroups, a quarter of the latency per tile); no identical-prefix plan
constexpr int64_t SMALL_TILES_MAX_PATHS = 8192;
bool small_tiles(const kprn_handle* h, int64_t N, bool has_plan);
void handover_stats(kprn_handle* h, const kprn_batch* b, int64_t* out /*[4]: pairs, steps moved, longest workgroup in half steps without / with*/);
HoArgs handover_args(kprn_handle* h, int grid);   // this launch's hand-over context (epoch 0: off)
void prefix_forward(kprn_handle* h, const kprn_batch* b);
bool forward_dual(kprn_handle* h, const kprn_batch* bt, const kprn_batch* bs, float* S_score);
bool catch_up_with_prefix(kprn_handle* h, const kprn_batch* b, float* W, float* g, float* m, float* v, int32_t* last, int32_t t_now, const float* step_tab,
                          float b1, float b2, float eps);
void forward_mc(kprn_handle* h, const kprn_batch* b, bool save);
void mc_prepare(kprn_handle* h);
bool prefix_backward(kprn_handle* h, const kprn_batch* b, int64_t n_tiles);
bool bwd_supported(const kprn_handle* h, int T);
bool transpose_job(kprn_handle* h, kk::TransposeJob* tj);

}  // namespace fused
