// hipcc-flags: -fno-slp-vectorize
// configs[3] (BASELINE.json: 20 M entities, d = 128 -> D = H = 384, "bf16 MFMA LSTM"): BPTT through the FastLSTM layer as ONE persistent launch.
//
// Stands for the backward of   nn.Sequencer(nn.FastLSTM(D, H)), all T steps     release/songPathRnn/model/OneModel.lua:236,268-274
// (gradInput of the T cells + the recurrent gradient dh_{t-1} = dA_t W_o2g) at the bf16 pipeline's precision, replacing per step one
// k_gates_bwd16_frag launch (1.2 GB of HBM traffic each: saves in, dH / dC in and out, dA and dA^T out) and one dh GEMM launch
// (lstm_bf16.hip backward(): 6 + 5 launches, 1.8 ms of the 6.6 ms step).  What those launches paid for was state through HBM: dH and dC
// (fp32, [N][H]) round-tripped every step, dA re-read by the recurrent product.  Here a workgroup owns a 64-path tile for t = T-1 .. 0:
//   * dc_t and the accumulators of dh_{t-1} never leave registers (2 x 96 per lane: one wave per SIMD, the 512-entry file); the finished
//     dh_t waits in LDS in lane-private 16-byte slots (96 KB: written once per step, read back quad by quad -- no barrier, no conflicts).
//   * The recurrent product is taken TRANSPOSED on v_mfma_f32_32x32x16_bf16, as in the forward (lstm_bf16_persist.hip): the weight
//     fragment (W_o2g^T: 32 hidden units x 16 gate columns) is the A operand, dA_t (16 gate columns x 32 paths) the B operand.  In the
//     C/D layout a lane then holds dh_{t-1} of FOUR CONSECUTIVE HIDDEN UNITS of one path per register quad -- exactly the (path, 4 units)
//     quad whose saved gates the forward wrote as one record of its fragment-order planes (Cell::store): the cell backward is lane-local
//     on the accumulators and every save is read with one coalesced 1 KiB wave load.
//   * K (the 4H = 1536 gate columns) is walked in 12 chunks of 32 hidden units x 4 gates.  Wave w's three 32-row result tiles are
//     composed so that register quad q of tile j is this lane's quad of chunk 4 j + q (rows 8 q + 4 half + r <-> hidden unit
//     32 (4 j + q) + 8 w + 4 half + r): per chunk EVERY lane of EVERY wave owns exactly one quad per path tile, so the cell backward of
//     chunk c + 1 is spread evenly over the four SIMDs' VALUs while the matrix cores run chunk c's product.
//   * dA of a chunk goes to LDS in B-FRAGMENT order (a lane's [di4 dg4] / [df4 do4] are two 16-byte pieces of k-step 2 w + half:
//     the store is one ds_write_b128 into the lane's own slot, the product's operand read one ds_read_b128), double buffered:
//     2 x 16 KB.  Each weight fragment (1 KiB, packed by k_pack_wb in exactly the order it is read) is fetched L2 -> registers by
//     exactly ONE wave and used for both path tiles.
//   * Outputs per chunk: dA row-major (the dx product's operand; v_permlane32_swap pairs the two lanes of a path into 16-byte stores), dA^T
//     in 16-byte pieces of 8 consecutive paths (the split-K dW products' k-contiguous operand) read from a second, TRANSPOSED LDS tile
//     that the lanes fill with two-byte writes (fire and forget; the first version gathered the pieces with 32 ds_read_u16 per thread and
//     chunk, each a round trip the single wave per SIMD waited for: 2.0 ms per launch), and the bias gradient = row sums of dA^T, kept in
//     an LDS table for the whole launch and flushed with one atomic per gate column per workgroup.
// HBM-bound by construction: per (path, step) it reads 4.6 KB of saves and writes 6 KB (dA + dA^T); the product and the cell ride under that.
// Index algebra replayed lane by lane in numpy: tests/test_persist_layout.py (backward model).
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "kprn_internal.h"

namespace bf16p {

typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace pb {

constexpr int H = 384;          // hidden units (instantiated shape)
constexpr int NW = 4;           // waves per workgroup, one per SIMD
constexpr int MJ = 3;           // 32-row result tiles per wave: NW * MJ * 32 = H
constexpr int NCH = 12;         // K chunks: 32 hidden units x 4 gates = 128 gate columns = 8 k-steps
constexpr int KSC = 8;          // k-steps per chunk
constexpr int FR = NCH * KSC * MJ;   // weight fragments per wave and step (288 KiB)
// With the entity slice of dx formed in this launch (DXE, round 5): a FOURTH result tile per wave -- rows = entity columns 32 w .. 32 w + 31 of the step
// input, dx_e[n][col] = sum_k dA_t[n][k] W_i2g[k][dt + col] -- fed by one more weight fragment per k-step (W_i2g^T entity rows, 96 KiB more per wave and step).
constexpr int DE = 128;         // entity columns (instantiated shape: 4 waves x 32)
constexpr int MJX = MJ + 1;
constexpr int FRX = NCH * KSC * MJX;
// ring depth of the DXE variant = the template argument itself (8 or 16: divides the 32 fragments of a chunk).  hipcc -S, gfx950: 8 -> 512 registers, 12 spilled
// dwords, all outside the step loop; 16 -> 34 spilled, 8 reloads inside the loop.
#ifndef KPRN_BPTT_PF
#define KPRN_BPTT_PF 12
#endif
#ifndef KPRN_BPTT_SD
#define KPRN_BPTT_SD 2
#endif
constexpr int PF = KPRN_BPTT_PF;   // fragments in flight per wave (12 = 4 k-steps ahead; divides the fragments of a chunk)
constexpr int SD = KPRN_BPTT_SD;   // chunks the saves are requested ahead of their cell backward (one register set of 24 per chunk in flight)
static_assert(NCH % SD == 0, "the save sets keep their phase across steps");
constexpr int UREC = H / 4 * 32;     // records (quads) of one unit of 32 rows and one step: [forward chunk 6][forward wave 8][lane 64]
// geometry of a work tile of NPT path tiles (32 rows each): NPT = 2 (64 rows, one workgroup per CU) or 1 (32 rows, two workgroups per CU)
template <int NPT> struct Geo {
  static constexpr int PW = 32 * NPT;             // paths of a tile
  static constexpr int BUF = NPT * KSC * 1024;    // bytes of one dA chunk tile in LDS
  static constexpr int TP = PW * 2 + 16;          // row pitch of the transposed chunk tile (bytes): the two halves of a wave write rows 4 apart -> other banks
  static constexpr int TT = 128 * TP;             // bytes of the transposed chunk tile: 4 gates x 32 hidden units rows
  static constexpr int NO = PW / 8;               // 16-byte pieces (8 paths) of a row
  static constexpr int RPP = 256 / NO;            // rows of the transposed tile the 256 threads cover per pass
  static constexpr int NI = 128 / RPP;            // passes = pieces per thread and chunk
  static constexpr int DHL = MJ * NPT * 4 * 4096; // bytes of dh_t in LDS
  static constexpr int LDS = 2 * BUF + TT + 4 * H * 4 + DHL;
};


struct BArgs {
  const bf16x8* A0; const bf16x8* A1; const bf16x4* cF;   // the forward's fragment-order saves: [i4 g4], [f4 o4], c (records of 4 hidden units)
  int64_t NU, step_recs;     // units of 32 rows; records per step (= NU UREC)
  const float* dS;           // [N] d loss / d S[n][cid]
  const float* Wc;           // [H] row cid of out.weight: dh_T[n] = dS[n] Wc
  const bf16* WpB;           // packed W_o2g^T fragments [NW][FR][64 lanes][8]
  bf16* dA;                  // [T][N][4H] row-major
  bf16* dAT;                 // [4H][ldT], this step's block at column t Np
  float* dXe;                // DXE: [T][N][DE] fp32, the entity slice of dx (what the entity gather-reduce reads)
  float* gbias;              // [4H] += column sums of dA (the bf16-rounded values)
  int64_t N, Np, ldT; int T;
  int64_t tiles;             // ceil(N / (32 NPT))
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// workgroup barrier ordering LDS traffic only (the save prefetches, the weight ring and the dA stores stay in flight across it)
__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000); }
template <class V> __device__ __forceinline__ V ldb(rsrc_t r, unsigned voff, unsigned soff) {
  if constexpr (sizeof(V) == 16) return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
  else { static_assert(sizeof(V) == 8, "8- or 16-byte pieces"); return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0)); }
}
template <class V> __device__ __forceinline__ void stb(rsrc_t r, unsigned voff, unsigned soff, V v) {
  static_assert(sizeof(V) == 8, "8-byte pieces");
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), r, (int)voff, (int)soff, 0);
}

__device__ __forceinline__ float tanh_fast(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f; }

// the saves of this lane's quad of one chunk, per path tile
template <int NPT> struct SvT { bf16x8 a0[NPT], a1[NPT]; bf16x4 c[NPT], cp[NPT]; };

// DBG: 2 = no row-major copy of dA (the product default: dx reads the transposed image, lstm_bf16.hip gx::k_gemm16xt; 0 keeps it for the row-major
// dx product).  Measurement builds (KPRN_PERSIST_VARIANTS + KPRN_PERSIST_BWD_DBG) add: 1 no dA^T / bias pass, 4 no product, 8 no save loads
template <int NPT, int DBG, int DXE = 0>
__global__ __launch_bounds__(64 * NW, NPT == 1 ? 2 : 1) void k_lstm16_bwd_persist(BArgs a) {
  static_assert(NW * MJ * 32 == H && NCH * 32 == H && MJ * 4 == NCH && (KSC * MJ) % PF == 0 && (DXE == 0 || (KSC * MJX) % DXE == 0) && NW * 32 == DE, "shape algebra of the backward tile");
  constexpr int MJP = DXE ? MJX : MJ;      // result tiles per wave in the product
  constexpr int FRP = DXE ? FRX : FR;      // weight fragments per wave and step
  constexpr int PFP = DXE ? DXE : PF;      // ring depth
  typedef Geo<NPT> GE;
  typedef SvT<NPT> Sv;
  constexpr int BUF = GE::BUF, TP = GE::TP, TT = GE::TT, NO = GE::NO, RPP = GE::RPP, NI = GE::NI, PW = GE::PW;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* const buf = smem;                                   // 2 x BUF: dA chunk tiles in B-fragment order [pt][k-step][slot][16 B]
  char* const tt = smem + 2 * BUF;                          // dA chunk tile TRANSPOSED: [gate 4][unit 32] rows of 64 paths (pitch TP bytes)
  float* const sdb = (float*)(smem + 2 * BUF + TT);         // [4H] bias-gradient sums of this workgroup
  char* const dhl = smem + 2 * BUF + TT + 4 * H * 4 + threadIdx.x * 16;   // dh_t of the tile: [(j NPT + pt) 4 + q][thread][16 B], this thread's slots
  const int tid = threadIdx.x, lane = tid & 63, ln = lane & 31, half = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t G = gridDim.x, b = blockIdx.x;
  const int64_t t_beg = a.tiles * b / G, t_end = a.tiles * (b + 1) / G;
  if (t_beg >= t_end) return;
  for (int i = tid; i < 4 * H; i += 64 * NW) sdb[i] = 0.f;
  const int T = a.T;
  const rsrc_t rW = make_rsrc(a.WpB + (int64_t)w * FRP * 512);
  const unsigned l16 = (unsigned)lane * 16u, l8 = (unsigned)lane * 8u;
  bf16x8 ring[PFP];
#pragma unroll
  for (int s = 0; s < PFP; ++s) ring[s] = ldb<bf16x8>(rW, l16, (unsigned)s * 1024u);
  // the dA^T pass of a chunk: thread -> (octet of 8 consecutive paths, row kc0 + 32 i of the transposed tile = gate i, unit kc0 of the chunk)
  const int oct = tid % NO, kc0 = tid / NO;   // row kc0 + RPP i of the transposed tile = gate (RPP i + kc0) / 32, unit (kc0 & 31)
  const unsigned et_voff = (unsigned)(((int64_t)((kc0 >> 5) * H + (kc0 & 31)) * a.ldT + 8 * oct) * 2);   // (host: H + 32 rows of dA^T span < 4 GB)
  bar();

  for (int64_t tile = t_beg; tile < t_end; ++tile) {
    const int64_t row0 = tile * PW;
    // units of 32 rows of the tile (the second may lie past the end: its lanes are masked, its loads stay in range)
    const int64_t u0 = tile * NPT;
    const unsigned d1_16 = (NPT > 1 && u0 + 1 < a.NU) ? (unsigned)UREC * 16u : 0u;   // byte distance of path tile 1's records (16-byte planes)
    bool valid[NPT];
    float ds[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
      const int64_t n = row0 + 32 * pt + ln;
      valid[pt] = n < a.N;
      ds[pt] = valid[pt] ? a.dS[n] : 0.f;
    }
    // saves of (step t, chunk c): forward group of 8 hidden units G8 = 4 c + w = (forward chunk G8 >> 3, forward wave G8 & 7), record
    // ((unit NCHF + chunk) 8 + wave) 64 + lane  ->  byte offset (4 c + w) 1024 (+ 16 lane) inside the unit's block of a 16-byte plane
    auto request = [&](int t, auto cc, Sv& s) {
      constexpr int c = decltype(cc)::value;
      if constexpr ((DBG & 8) != 0) {
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
          for (int q = 0; q < 8; ++q) { s.a0[pt][q] = (bf16)0.5f; s.a1[pt][q] = (bf16)0.5f; s.c[pt][q & 3] = (bf16)0.25f; s.cp[pt][q & 3] = (bf16)0.25f; }
        return;
      }
      const int64_t r0 = (int64_t)t * a.step_recs + u0 * UREC;
      const rsrc_t r_a0 = make_rsrc(a.A0 + r0), r_a1 = make_rsrc(a.A1 + r0), r_c = make_rsrc(a.cF + r0);
      const unsigned so = (unsigned)(4 * c + w) * 1024u;
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) {
        const unsigned sp = so + (pt ? d1_16 : 0u);
        s.a0[pt] = ldb<bf16x8>(r_a0, l16, sp);
        s.a1[pt] = ldb<bf16x8>(r_a1, l16, sp);
        s.c[pt] = ldb<bf16x4>(r_c, l8, sp >> 1);
      }
      if (t > 0) {
        const rsrc_t r_p = make_rsrc(a.cF + (r0 - a.step_recs));
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) s.cp[pt] = ldb<bf16x4>(r_p, l8, (so + (pt ? d1_16 : 0u)) >> 1);
      } else {
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
          for (int q = 0; q < 4; ++q) s.cp[pt][q] = (bf16)0.f;
      }
    };

    // ---- state of the tile: dc_t and the accumulators of dh_{t-1} (register 4 q + r of tile j <-> chunk 4 j + q, unit r of the lane's quad);
    // dh_t in LDS.  dh_T = dS[n] W_out[cid] (nn.Linear backward on the selected column), this lane's slice Wc[32 c + 8 w + 4 half + r]
    f32x16 acc[MJ][NPT], dcs[MJ][NPT];
    f32x16 accx[NPT];   // DXE: dx_e of the step being processed (rows = entity columns 32 w + (r & 3) + 8 (r >> 2) + 4 half, column = path)
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accx[pt][r] = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const f32x4 wq = *(const f32x4*)(a.Wc + 32 * c + 8 * w + 4 * half);
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt) *(f32x4*)(dhl + (((c >> 2) * NPT + pt) * 4 + (c & 3)) * 4096) = wq * ds[pt];
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dcs[j][pt][r] = 0.f; acc[j][pt][r] = 0.f; }

    Sv sv[SD];
    static_for<0, SD>([&](auto cc) __attribute__((always_inline)) { request(T - 1, cc, sv[decltype(cc)::value]); });

    for (int t = T - 1; t >= 0; --t) {
      const rsrc_t r_dA = make_rsrc(a.dA + ((int64_t)t * a.N + row0) * (4 * H));
      bf16* const dat_t = a.dAT + (int64_t)t * a.Np + row0;
      const bool et_ok = row0 + 8 * oct < a.Np;
      bf16x8 pc[NPT][2];    // the chunk's pieces [di4 dg4] / [df4 do4] of this lane, kept for the transposed tile (written behind the barrier)
      f32x4 dh4[NPT];       // dh_t of the quads about to be processed
      bf16x8 ev;            // the dA^T piece in flight

      auto read_dh = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) dh4[pt] = *(const f32x4*)(dhl + (((c >> 2) * NPT + pt) * 4 + (c & 3)) * 4096);
      };
      // ---- the cell backward of chunk c, cut into slices that ride behind the product's MFMAs (one wave per SIMD: nothing else fills an
      // MFMA's shadow, and hipcc, left alone, sinks every prefetch load down to its use: the first version waited vmcnt(0) in front of most MFMAs).
      // G = 0 .. 7: element (pt = G >> 2, r = G & 3);  8, 9: the path tile's pieces -> B-fragment tile (c & 1) + row-major plane;
      // 10: the saves of chunk c + 2 (or of the next step's first chunks) and dh of chunk c + 1 are requested.
      auto gslice = [&](auto cc, auto gg, Sv& s) {
        constexpr int c = decltype(cc)::value, Gs = decltype(gg)::value, j = c >> 2, q = c & 3;
        if constexpr (Gs < 8 && (Gs >> 2) < NPT) {
          constexpr int pt = Gs >> 2, r = Gs & 3;
          const float ig = (float)s.a0[pt][r], gv = (float)s.a0[pt][4 + r], fg = (float)s.a1[pt][r], og = (float)s.a1[pt][4 + r];
          const float tc = tanh_fast((float)s.c[pt][r]);
          const float cp = (float)s.cp[pt][r];
          const float dh = dh4[pt][r];
          const float dO = dh * tc;
          const float dc = dcs[j][pt][4 * q + r] + dh * og * (1.f - tc * tc);
          pc[pt][0][r] = (bf16)(dc * gv * ig * (1.f - ig));
          pc[pt][0][4 + r] = (bf16)(dc * ig * (1.f - gv * gv));
          pc[pt][1][r] = (bf16)(dc * cp * fg * (1.f - fg));
          pc[pt][1][4 + r] = (bf16)(dO * og * (1.f - og));
          dcs[j][pt][4 * q + r] = dc * fg;
        } else if constexpr (Gs >= 8 && Gs < 10 && (Gs - 8) < NPT) {
          constexpr int pt = Gs - 8;
          char* const dst = buf + (c & 1) * BUF;
          // B-fragment order: k-step 2 w + half, k-group ab (piece [di dg] -> 0, [df do] -> 1), slot ln + 32 ab
          *(bf16x8*)(dst + ((pt * KSC + 2 * w + half) * 64 + ln) * 16) = pc[pt][0];
          *(bf16x8*)(dst + ((pt * KSC + 2 * w + half) * 64 + 32 + ln) * 16) = pc[pt][1];
          if constexpr ((DBG & 2) == 0) {
            // row-major plane: the two lanes of a path (half 0 / 1) hold units 8 w .. + 3 / + 4 .. + 7 of every gate; one v_permlane32_swap per dword
            // hands lane half 0 the 16 bytes of gate i (f) and lane half 1 those of gate g (o): two 16-byte stores per lane instead of four of 8
            const u32x4 x0 = __builtin_bit_cast(u32x4, pc[pt][0]), x1 = __builtin_bit_cast(u32x4, pc[pt][1]);
            u32x4 s0, s1;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              const auto r0 = __builtin_amdgcn_permlane32_swap(x0[d], x0[2 + d], false, false);   // (di dword d, dg dword d)
              s0[d] = r0[0]; s0[2 + d] = r0[1];
              const auto r1 = __builtin_amdgcn_permlane32_swap(x1[d], x1[2 + d], false, false);   // (df, do)
              s1[d] = r1[0]; s1[2 + d] = r1[1];
            }
            if (valid[pt]) {
              const unsigned vo_ = (unsigned)(((32 * pt + ln) * (4 * H) + half * H) * 2);
              const unsigned so_ = (unsigned)((32 * c + 8 * w) * 2);
              __builtin_amdgcn_raw_buffer_store_b128(s0, r_dA, (int)vo_, (int)so_, 0);
              __builtin_amdgcn_raw_buffer_store_b128(s1, r_dA, (int)vo_, (int)(so_ + (unsigned)(2 * H * 2)), 0);
            }
          }
        } else if constexpr (Gs == 10) {
          if constexpr (c + SD < NCH) request(t, std::integral_constant<int, c + SD>{}, s);
          else if (t > 0) request(t - 1, std::integral_constant<int, c + SD - NCH>{}, s);
          if constexpr (c + 1 < NCH) read_dh(std::integral_constant<int, c + 1>{});
        }
      };
      // the chunk's pieces into the transposed tile: row gate 32 + 8 w + 4 half + r, column = path (16 two-byte LDS writes per path tile: fire and forget)
      auto write_T = [&]() {
        if constexpr ((DBG & 1) != 0) return;
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *(bf16*)(tt + (g4 * 32 + 8 * w + 4 * half + r) * TP + (32 * pt + ln) * 2) = pc[pt][g4 >> 1][4 * (g4 & 1) + r];
      };
      // ---- dA^T pieces and the bias sums of chunk c from the transposed tile, in slices G = 11 .. 15: thread (oct, kc0) owns 8 consecutive paths of
      // rows kc0 + 32 i (gate i, unit kc0); piece i is read in slice 11 + i and leaves in slice 12 + i
      auto eslice = [&](auto cc, auto gg) {
        constexpr int c = decltype(cc)::value, Gs = decltype(gg)::value;
        if constexpr ((DBG & 1) != 0) return;
        if constexpr (Gs >= 12 && Gs < 12 + NI) {
          constexpr int i = Gs - 12, gbase = (RPP * i) / 32;   // this pass's rows: gates gbase (+ kc0 >> 5)
          // address = uniform 64-bit base (scalar registers: plane + step block + row block of (gate, chunk c)) + this thread's 32-bit byte offset
          // (its row of the block, octet).  The row block is made opaque: otherwise hipcc precomputes the (chunk, pass) offsets of the whole
          // step outside the step loop and spills registers for them.
          int64_t sofs = (int64_t)(gbase * H + 32 * c) * a.ldT;
          asm volatile("" : "+s"(sofs));
          if (et_ok) *(bf16x8*)((char*)(dat_t + sofs) + et_voff) = ev;
          float sum = 0.f;
#pragma unroll
          for (int x = 0; x < 8; ++x) sum += (float)ev[x];
          // sum over the lanes of the row's pieces (DPP: quad xor 1, quad xor 2, and for 8 pieces the mirror of the half row)
          sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0xB1, 0xf, 0xf, true));
          sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x4E, 0xf, 0xf, true));
          if constexpr (NO == 8) sum += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sum), 0x141, 0xf, 0xf, true));
          if (oct == 0) sdb[(gbase + (kc0 >> 5)) * H + 32 * c + (kc0 & 31)] += sum;   // (one owner thread per gate column: no atomics)
        }
        if constexpr (Gs >= 11 && Gs < 11 + NI) ev = *(const bf16x8*)(tt + (RPP * (Gs - 11) + kc0) * TP + oct * 16);
      };

      // first chunk of the step: nothing to ride behind
      read_dh(std::integral_constant<int, 0>{});
      static_for<0, 10>([&](auto gg) __attribute__((always_inline)) { gslice(std::integral_constant<int, 0>{}, gg, sv[0]); });
      request(t, std::integral_constant<int, SD>{}, sv[0]);
      read_dh(std::integral_constant<int, 1>{});
      bar();
      write_T();
      bar();
      // Per chunk two regions: A = product(c): 8 k-steps x 3 result tiles x 2 path tiles (dh_{t-1} += W_o2g^T[:, chunk c] dA_t[chunk c]) with the
      // dA^T pass of chunk c and the cell backward of chunk c + 1 [writes the OTHER B-fragment tile] in its shadow, barrier, B = the transposed
      // tile of chunk c + 1 from the pieces kept in registers (every wave has finished reading chunk c's), barrier.
      // (no run-time branch around the MFMA groups: at every join hipcc's wait-count bookkeeping assumes that none of the later ring loads were
      //  issued -- it then waits vmcnt(0) in front of every MFMA and the prefetch ring is gone)
      auto chunks = [&](auto hp) __attribute__((always_inline)) {
        constexpr bool HP = decltype(hp)::value && !(DBG & 4);
        static_for<0, NCH>([&](auto cc) __attribute__((always_inline)) {
          constexpr int c = decltype(cc)::value;
          const char* const src = buf + (c & 1) * BUF + lane * 16;
          bf16x8 bfr[2][NPT];
          if constexpr (HP) {
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) bfr[0][pt] = *(const bf16x8*)(src + (pt * KSC) * 1024);
          }
          static_for<0, KSC * MJP>([&](auto gg) __attribute__((always_inline)) {
            constexpr int g = decltype(gg)::value, ks = g / MJP, j = g % MJP;
            if constexpr (HP) {
              constexpr int f = c * KSC * MJP + g, slot = f % PFP;
              if constexpr (j < MJ) {
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) acc[j][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[slot], bfr[ks & 1][pt], acc[j][pt], 0, 0, 0);
              } else {
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) accx[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[slot], bfr[ks & 1][pt], accx[pt], 0, 0, 0);
              }
              if constexpr (j == 0 && ks + 1 < KSC) {
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) bfr[(ks + 1) & 1][pt] = *(const bf16x8*)(src + (pt * KSC + ks + 1) * 1024);
              }
              constexpr int fn = (f + PFP) % FRP;   // (the next step walks the same fragments again)
              ring[slot] = ldb<bf16x8>(rW, l16, (unsigned)fn * 1024u);
            }
            if constexpr (c + 1 < NCH && g <= 10) gslice(std::integral_constant<int, c + 1>{}, gg, sv[(c + 1) % SD]);
            eslice(cc, gg);
            __builtin_amdgcn_sched_barrier(0);
          });
          if constexpr (c + 1 < NCH) {
            bar();
            write_T();
          }
          bar();
        });
      };
      // The product runs at t = 0 as well (its result, "dh_{-1}", is dropped): one sixth more MFMA work -- the matrix cores are not what bounds this
      // launch -- for ONE copy of the step body with no branches in it.  (Two copies, with and without, cost 500 spilled registers.)
      chunks(std::true_type{});
      // DXE: dx_e of step t is complete (this one is NOT dropped at t = 0): a lane holds four quads of 4 consecutive entity columns of its path
      if constexpr (DXE != 0) {
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
          if (valid[pt]) {
            float* const dst = a.dXe + ((int64_t)t * a.N + row0 + 32 * pt + ln) * DE + 32 * w + 4 * half;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(f32x4*)(dst + 8 * q) = f32x4{accx[pt][4 * q], accx[pt][4 * q + 1], accx[pt][4 * q + 2], accx[pt][4 * q + 3]};
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) accx[pt][r] = 0.f;
        }
      }
      // dh_{t-1} is complete: it becomes the step's dh (lane-private LDS slots), the accumulators start again from zero
      {
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
          for (int pt = 0; pt < NPT; ++pt) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *(f32x4*)(dhl + ((j * NPT + pt) * 4 + q) * 4096) = f32x4{acc[j][pt][4 * q], acc[j][pt][4 * q + 1], acc[j][pt][4 * q + 2], acc[j][pt][4 * q + 3]};
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][pt][r] = 0.f;
          }
      }
    }
  }
  bar();
  for (int i = tid; i < 4 * H; i += 64 * NW) {
    const float v = sdb[i];
    if (v != 0.f) unsafeAtomicAdd(a.gbias + i, v);
  }
}

// ---- weight packing (whenever the dense parameters change) -----------------------------------------------------------------------
// WpB[((w FR + f) 64 + lane)][e], f = (c KSC + ks) MJ + j: A fragment of result tile j of wave w for k-step ks of chunk c.  Lane = (m, kg):
// row m <-> hidden unit 32 (4 j + (m >> 3)) + 8 w + (m & 7); element e <-> gate column k = 16 ks + 8 kg + e of the chunk = gate
// 2 kg + (e >> 2) of hidden unit 32 c + 8 (ks >> 1) + 4 (ks & 1) + (e & 3) (the order the cell backward writes its pieces in).
// Value: W_o2g[gate H + unit_k][unit_m]  (dh_{t-1}[m] = sum_k dA_t[k] W_o2g[k][m]).  From the fp32 master, rounded once.
// mjp = MJ: the recurrent fragments only; mjp = MJX (DXE): fragment j = MJ of every k-step is W_i2g^T's entity slice -- row m <-> entity column 32 w + m,
// value W_i2g[gate H + unit_k][col0 + 32 w + m]  (dx_e[m] = sum_k dA_t[k] W_i2g[k][col0 + m]; col0 = dt, Din = the row pitch of W_i2g).
__global__ void k_pack_wb(const float* __restrict__ Wo, const float* __restrict__ Wi, int Din, int col0, int mjp, bf16* __restrict__ WpB) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int fr = NCH * KSC * mjp;
  if (i >= (int64_t)NW * fr * 64) return;
  const int lane = (int)(i & 63);
  const int f = (int)((i >> 6) % fr), w = (int)((i >> 6) / fr);
  const int j = f % mjp, ks = (f / mjp) % KSC, c = f / (mjp * KSC);
  const int m = lane & 31, kg = lane >> 5;
  const int um = 32 * (4 * j + (m >> 3)) + 8 * w + (m & 7);
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int gate = 2 * kg + (e >> 2), uk = 32 * c + 8 * (ks >> 1) + 4 * (ks & 1) + (e & 3);
    o[e] = (bf16)(j < MJ ? Wo[(int64_t)(gate * H + uk) * H + um] : Wi[(int64_t)(gate * H + uk) * Din + col0 + 32 * w + m]);
  }
  *(bf16x8*)(WpB + i * 8) = o;
}

}  // namespace pb

// ---- host side --------------------------------------------------------------------------------------------------------------
struct PersistSaves { const bf16* CsF; const bf16* ActF0; const bf16* ActF1; const bf16* HsF; int64_t NU, step_recs; int NW; };   // (also declared in lstm_bf16.hip)
struct PersistBwdState { bf16* WpB = nullptr; int grid = 0; int packed_mjp = 0; };

// Everything persist_backward() requires of a batch is decided HERE, before the backward has accumulated anything: a batch the launch cannot take
// (T x N beyond the 32-bit row-block offsets of dA^T: about 4.8 M (path, step) positions at H = 384) trains through the per-step loop instead.
static bool persist_bwd_offsets_ok(int64_t N, int T) {
  const int64_t Np = (N + 7) & ~(int64_t)7;
  return ((int64_t)T * Np + 512) * (pb::H + 64) * 2 < ((int64_t)1 << 32);   // (+ 512: the row pitch of dA^T may be padded, lstm_bf16.hip t_pitch)
}
bool persist_bwd_shape_ok(const kprn_handle* h, const PersistSaves& sv, int64_t N, int T) {
  static const bool off = [] { const char* e = getenv("KPRN_BF16_BWD_PERSIST"); return e && e[0] == '0'; }();
  return !off && h->cfg.L == 1 && h->cfg.H == pb::H && sv.NW == 8 && T >= 1 && persist_bwd_offsets_ok(N, T);
}

void persist_bwd_release(void*& st) {
  PersistBwdState* p = (PersistBwdState*)st;
  if (!p) return;
  if (p->WpB) (void)hipFree(p->WpB);
  delete p;
  st = nullptr;
}

// dA_t (row-major and transposed) for all steps + the bias gradient, from the persistent forward's saves; ws.dS holds d loss / d S[:, cid]
// dXe (nullable): the launch also forms the entity slice of dx, [T][N][de] fp32 (de = 128 columns from dt on) -- persist_bwd_dxe_ok() says whether it can
bool persist_bwd_dxe_ok(const kprn_handle* h) { return h->cfg.de == pb::DE && h->D >= h->cfg.dt + pb::DE; }
void persist_backward(kprn_handle* h, int64_t N, int T, int cid, const PersistSaves& sv, void*& st, bool repack, bf16* dA16 /* nullable: no row-major copy */, bf16* dAT16, int64_t Np,
                      float* dXe, int64_t ldT) {
  hipStream_t strm = h->stream;
  PersistBwdState* p = (PersistBwdState*)st;
  if (!p) {
    p = new PersistBwdState();
    st = p;
    int dev = 0, ncu = 256;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    p->grid = ncu;
    void* q = nullptr;
    hipError_t e = kprn_dev_malloc(&q, (size_t)pb::NW * pb::FRX * 1024 + 64);
    if (e != hipSuccess) throw KprnError{KPRN_E_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e)};
    p->WpB = (bf16*)q;
    repack = true;
  }
  const bool dxe = dXe != nullptr;
  KPRN_REQUIRE(!dxe || (persist_bwd_dxe_ok(h) && !dA16), KPRN_E_ARG, "persistent BPTT: dx_e in the launch needs de = 128 and no row-major dA");
  const int mjp = dxe ? pb::MJX : pb::MJ;
  if (repack || p->packed_mjp != mjp) {
    const int64_t total = (int64_t)pb::NW * pb::NCH * pb::KSC * mjp * 64;
    hipLaunchKernelGGL(pb::k_pack_wb, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, strm, h->dense + h->layer[0].Wo, h->dense + h->layer[0].Wi, h->D, h->cfg.dt, mjp, p->WpB);
    HIP_TRY(hipGetLastError());
    p->packed_mjp = mjp;
  }
  pb::BArgs a;
  memset(&a, 0, sizeof(a));
  a.A0 = (const bf16x8*)sv.ActF0; a.A1 = (const bf16x8*)sv.ActF1; a.cF = (const bf16x4*)sv.CsF;
  a.NU = sv.NU; a.step_recs = sv.step_recs;
  a.dS = h->ws.dS; a.Wc = h->dense + h->off_outW + (int64_t)cid * pb::H;
  a.WpB = p->WpB; a.dA = dA16; a.dAT = dAT16; a.dXe = dXe; a.gbias = h->g_dense + h->layer[0].bi;
  KPRN_REQUIRE(ldT >= (int64_t)T * Np && ldT <= (int64_t)T * Np + 512 && (ldT & 7) == 0, KPRN_E_ARG, "persistent BPTT: bad row pitch of dA^T");
  a.N = N; a.Np = Np; a.ldT = ldT; a.T = T;
  // tile height: 64 rows, one workgroup per CU (default), or 32 rows, two workgroups per CU (KPRN_BPTT_NPT=1: each weight fragment then serves one
  // path tile only -- twice the weight bytes through the L1 path -- for two independent barrier domains per CU)
  static const int npt_env = KPRN_DEV_ENV("KPRN_BPTT_NPT") ? atoi(KPRN_DEV_ENV("KPRN_BPTT_NPT")) : 2;
  const int npt = npt_env == 1 ? 1 : 2;
  a.tiles = (N + 32 * npt - 1) / (32 * npt);
  KPRN_REQUIRE(persist_bwd_offsets_ok(N, T) && Np == ((N + 7) & ~(int64_t)7), KPRN_E_ARG, "persistent BPTT: shape not covered (persist_bwd_shape_ok decides before the backward starts)");
  int grid = (int)std::min<int64_t>((int64_t)p->grid * (npt == 1 ? 2 : 1), a.tiles);
  if (const char* e = getenv("KPRN_PERSIST_BWD_GRID")) grid = (int)std::max<int64_t>(1, std::min<int64_t>(grid, atoi(e)));   // (tests: several tiles per workgroup at small N)
  const size_t lds_bytes = npt == 1 ? (size_t)pb::Geo<1>::LDS : (size_t)pb::Geo<2>::LDS;
  typedef void (*Kern)(pb::BArgs);
  Kern k = dA16 ? (Kern)pb::k_lstm16_bwd_persist<2, 0> : (Kern)pb::k_lstm16_bwd_persist<2, 2>;   // (2: no row-major copy -- dx reads the transposed image)
  if (npt == 1) k = dA16 ? (Kern)pb::k_lstm16_bwd_persist<1, 0> : (Kern)pb::k_lstm16_bwd_persist<1, 2>;
  if (dxe) { KPRN_REQUIRE(npt == 2, KPRN_E_ARG, "persistent BPTT: dx_e in the launch is built for 64-row tiles"); k = h->bf16_bptt_dxe == 16 ? (Kern)pb::k_lstm16_bwd_persist<2, 2, 16> : (Kern)pb::k_lstm16_bwd_persist<2, 2, 8>; }
#ifdef KPRN_PERSIST_VARIANTS
  // measurement builds (scripts/gpu_persist_knockouts.py bwd): KPRN_PERSIST_BWD_DBG = knock-out mask
  if (const char* e = KPRN_DEV_ENV("KPRN_PERSIST_BWD_DBG")) {
    const int dbg = atoi(e) | (dA16 ? 0 : 2);
    bool found = (dbg == 0 || dbg == 2) && npt == 2;
#define KV(D) if (dbg == D) { k = (Kern)pb::k_lstm16_bwd_persist<2, D>; found = true; }
    KV(1) KV(2) KV(3) KV(4) KV(7) KV(8) KV(15)
#undef KV
    KPRN_REQUIRE(found, KPRN_E_ARG, "this variant of the persistent BPTT kernel is not compiled in");
  }
#endif
  HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  ProfScope ps(h, "lstm_persist_bf16_bwd");
  hipLaunchKernelGGL(k, dim3(grid), dim3(64 * pb::NW), lds_bytes, strm, a);
  HIP_TRY(hipGetLastError());
}

}  // namespace bf16p
