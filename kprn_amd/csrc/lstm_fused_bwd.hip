// Fused BPTT kernels of the KPRN path scorer for gfx950 (forward: lstm_fused_fwd.hip; design: DESIGN.md).
#include <memory>

#include "lstm_fused_common.h"

namespace fused {

// ============================================================================================
// Fused backward (BPTT) -- one launch per layer, top layer first.
//
// Stands in for the backward of nn.Sequencer(nn.FastLSTM) x L + FeatureEmbedding
// (model/OneModel.lua:223-274 via MyOptimizer.lua:195 model:backward): exact BPTT over all T
// steps, weight gradients accumulated over steps, LookupTable scatter-add for layer 1.
//
// Same tile / wave ownership as the forward: 64-path tiles, wave j owns hidden units [16j,16j+16) of all four
// gates, MFMA C layout (lane (ag, arow), register r <-> row 4 ag + r, col 16 j + arow).  Per tile, t = T-1 .. 0:
//   C. per 16-row m-tile: cell backward = 7 VALU ops per element on the factors the forward saved (NPL in
//      lstm_fused_common.h) -> dA (pre-activation grads) in C layout.  Those registers ARE the A operand of
//      dW += dA^T [in | h_prev] (k-slot = ag <-> row 4 ag + r); the B operands are the saved h fragments of
//      the four waves (plane 6: already in B-operand layout, loaded 1 KiB per instruction) or, bottom layer,
//      the re-gathered x_t tile in LDS.  dW (128 registers) lives in AGPRs for the whole launch.  dA also goes
//      row-major to an LDS tile.
//   E. [dx | dh_{t-1}] = dA [W_i2g | W_o2g]: A from the LDS dA tile (ds_read_b128, next fragment always in
//      flight), B = this wave's 128-register slice of the transposed weights, AGPR-stationary.  dh_{t-1} comes
//      out in exactly the layout stage C of step t-1 wants, so it never leaves registers; dx goes to HBM in
//      fragment order for the layer below (1 KiB stores, and the layer below reloads it the same way), or --
//      bottom layer -- is scattered to the embedding gradients straight from the accumulators.
// fp32 MFMA and VALU share the SIMD's FP32 lanes on gfx950 (scripts/ubench/mfma_rate.hip: their times add, also
// across waves), so the VALU work above is the part of the step the matrix pipe cannot hide; it is kept minimal.
struct BwdArgs {
  const int32_t* idx; int64_t N; int T, F, nT;
  const float *Wt, *We, *Wr; int dt, de, dr; int Vt, Vr;
  int L, layer;
  const float* WiT;        // [64][256]  = W_i2g^T of this layer
  const float* WoT;        // [64][256]
  const float* save_frag;  // forward's fragment-order saves: [(N/16)][T][L][4 waves][NPL][64 lanes][4]
  const float* dS;         // [N] (tile slot order) top layer: d loss / d S[n][classId]; the head backward dh_T = dS[n] W_out[classId][:] is formed in-kernel
  const float* wout_row;   // W_out[classId][0..64)
  float* gWout_row; float* gbout_c;  // top layer: gradient of W_out[classId][:] and b_out[classId] (sum_n dS[n] h_T[n][:], sum_n dS[n])
  float* DX;               // [(Npad/16)][T][4 waves][64 lanes][4] fragment order: in = dx of the layer above (not top), out = dx of this layer
  float* DXe;              // bottom layer, nullable: the ENTITY slice of dx row-major [(n T + t)][de] -- what the entity-gradient gather reads
                           // (128 contiguous bytes per position instead of 4 of every 16 bytes of a fragment-order block)
  int64_t Npad;            // N rounded up to the 64-row tile
  float* gWi; float* gbi; float* gWo;   // [256][64], [256], [256][64]
  float* gWt; float* gWe; float* gWr;   // bottom layer only
  float* part;             // [grid][2*256*64 + 256] per-workgroup partial dW_i2g | dW_o2g | db
  unsigned long long* timing;  // optional [grid][8] cycle counters (KPRN_TIMING=1)
  int dbg;                 // KPRN_DBG (measurement / cross-checks only): 1 skip the embedding backward, 8 small tables through the
                           // general scatter kernel, 16 entity table through the general scatter kernel (atomics) instead of the index
  int64_t n_tiles;
  // identical-prefix plan of the batch (all nullable): a tile runs steps T-1 .. tile_k[tile]; what flows into the skipped
  // steps is summed per prefix class into PG and finished by k_prefix_bwd
  const int32_t* tile_k;
  float* PG;               // [KCAP+1][PFB] this layer: sum over rows of dA at the tile's first executed step | of dc handed below it
  HoArgs ho;               // time-split tile hand-over (lstm_fused_common.h ho_plan)
  int small_lds;           // bottom layer: the type / relation table gradients are formed inside this launch (one-hot MFMAs on dx; option "fused_small_tables")
  // layer pipeline of small batches (k_lstm_bwd_dual): both layers' workgroups are resident at once, the top layer's publish dx of (tile, step) as soon as it
  // is stored, the bottom layer's wait for it step by step -- the bottom layer runs ONE step behind the top layer instead of a whole launch behind
  unsigned* pipe_flag;     // [n_tiles][MAXT_LDS] epoch (ho.epoch) of the dx of (tile, absolute step)
  int pipe;                // 0: no pipeline; 1: producer (top layer); 2: consumer (bottom layer)
};


constexpr int PART = 2 * 256 * 64 + 256;  // floats per workgroup partial slab
constexpr int LDD = 4 * DH + 4;  // dA tile row stride

// Cycle probes are compiled in only with -DKPRN_TIMING_PROBES (KPRN_TIMING=1 then prints them): eight 64-bit counters per wave cost
// 16 scalar registers for the whole launch, and with them the bottom-layer kernel spilled.
#ifdef KPRN_TIMING_PROBES
#define KPRN_PROBES_ON 1
#else
#define KPRN_PROBES_ON 0
#endif
#define TPROBE(slot)                                                                 \
  if (KPRN_PROBES_ON && a.timing) {                                                   \
    const unsigned long long now__ = __builtin_amdgcn_s_memtime();                    \
    tacc[slot] += now__ - tlast;                                                      \
    tlast = now__;                                                                    \
  }

// what stage C needs for one 16-row m-tile, requested one m-tile ahead
struct Pre {
  f32x4 P[6];    // this wave's backward factors
  f32x4 up;      // dx of the layer above (not top)
  f32x4 bin[4];  // B operands of dW_i2g: [in][row 4 ag + r][16 nt + arow] for nt = 0..3
  f32x4 bhp[4];  // B operands of dW_o2g: h^l_{t-1}, same layout
};

// NMT: 16-row m-tiles of a tile -- 4, or 1 for small batches (fused::small_tiles; no identical-prefix plan there)
// bx / G_: this workgroup's index among, and the number of, the workgroups of THIS layer's pass (k_lstm_bwd: the launch's; k_lstm_bwd_dual: half of it)
template <bool BOTTOM, bool TOP, int NMT>
__device__ __forceinline__ void bwd_body(const BwdArgs& a, const int bx, const int G_) {
  constexpr int MTR = 16 * NMT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = (KPRN_PROBES_ON && a.timing) ? __builtin_amdgcn_s_memtime() : 0ull;
  float* dA_t = lds;                                        // [64][LDD]
  float* in_t = dA_t + MT * LDD;                            // bottom: [64][LDA] x_t
  int32_t* ids = (int32_t*)(in_t + (BOTTOM ? MT * LDA : 0));  // bottom: 2 x the id planes of a tile (x_t re-gather): this tile's, and the next one's on its way (LDS-DMA)
  constexpr int IDS_BUF = MT * MAXT_LDS * 4;                   // ints per buffer (>= 3 planes: lstm_fused_common.h ids_stage_dma)
  float* pg = (float*)(ids + (BOTTOM ? 2 * IDS_BUF : 0));      // [KCAP+1][PFB] this workgroup's share of PG

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int j = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int arow = lane & 15, ag = lane >> 4;
  const int T = a.T, L = a.L, ly = a.layer;
  // (first thing in the kernel: its one round trip is in flight under the weight loads below)
  const HoPlan ho = ho_plan(a.ho, a.tile_k, a.n_tiles, T, 1, bx, G_);

  // ---- AGPR residents: this wave's slice of [W_i2g^T | W_o2g^T] (stage E's B operand) and the dW accumulators
  //      dwi/dwo[q][nt][r] <-> dW row q*64 + 16j + 4ag + r, col 16nt + arow
  f32x4 wiT[16], woT[16];
  {
    const float* wi_row = a.WiT + (int64_t)(j * 16 + arow) * (4 * DH) + ag * 4;
    const float* wo_row = a.WoT + (int64_t)(j * 16 + arow) * (4 * DH) + ag * 4;
#pragma unroll
    for (int S = 0; S < 16; ++S) { wiT[S] = *(const f32x4*)(wi_row + S * 16); woT[S] = *(const f32x4*)(wo_row + S * 16); }
#pragma unroll
    for (int S = 0; S < 16; ++S) { asm volatile("" : "+a"(wiT[S])); asm volatile("" : "+a"(woT[S])); }
  }
  f32x4 dwi[4][4], dwo[4][4];
  float dbias[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    dbias[q] = 0.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      dwi[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; dwo[q][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("" : "+a"(dwi[q][nt])); asm volatile("" : "+a"(dwo[q][nt]));
    }
  }

  const int64_t frag_unit = (int64_t)NPL * 256;                // floats per (m-tile, t, layer, wave)
  const int64_t frag_mt_stride = (int64_t)T * L * 4 * frag_unit;  // floats per m-tile
  // this wave's 16 dx columns belong to the type (0), entity (1) or relation (2) slice.  (The type / relation gradients were once
  // formed here as a one-hot MFMA on the dx accumulators: 16 extra MFMAs + a drain per step on two of the four waves, 8
  // launch-long VGPRs and the only register spills of the kernel; they are now a passenger job of the entity-gradient launch,
  // lstm_fused_bwd.hip k_small_grad.)
  const int wcls = (j * 16 < a.dt) ? 0 : ((j * 16 < a.dt + a.de) ? 1 : 2);
  GatherSrc gsrc;
  if (BOTTOM) gsrc = gather_src(a);
  const float wout_c = TOP ? a.wout_row[j * 16 + arow] : 0.f;
  float gwo = 0.f, gbo = 0.f;  // head gradient partials of this lane: column 16j + arow over its rows / sum of dS over its rows
  for (int c = tid; c < (KCAP + 1) * PFB; c += 256) pg[c] = 0.f;  // (first read: after the barriers of a whole tile)
  // bottom layer, waves that own a type / relation slice of dx: that table's gradient, [16 rows v][this wave's 16 columns] in the MFMA C layout
  // (lane (g, col) holds rows v = 4 g + i), two accumulators so that consecutive MFMAs do not chain
  f32x4 sacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  TPROBE(0)

  // A workgroup walks ITS tiles (blockIdx.x + i gridDim.x, the forward's assignment) in the order that finds the most in the 256 MB
  // memory-side cache.  Top layer: last tile first -- the training forward wrote the saved planes in ascending order, so the most
  // recent ones are the likeliest to still be there.  Bottom layer: first tile first -- by then the top layer's launch has streamed
  // every plane of ITS layer through the cache, and what is recent is the dx the top layer wrote last, i.e. of its first tiles.
  // (0.319 -> 0.312 ms per launch.)
  const int64_t n_mine = (a.n_tiles > (int64_t)bx) ? (a.n_tiles - 1 - bx) / G_ + 1 : 0;
  // Time-split hand-over (lstm_fused_common.h ho_plan): the LIGHT workgroup of a pair runs the first ho.d steps (T-1 .. T-d) of the heavy one's
  // first tile before its own tiles and publishes (dh, dc); the HEAVY one keeps that tile for last and resumes it at step T-d-1.
  float* const ho_slot = a.ho.state + (int64_t)ho.slot * HO_STATE + threadIdx.x * 4;   // [dh | dc][NMT][256 threads][4]
  // the pair's tile is the heavy workgroup's first one (ti = 0): last in the top layer's order anyway, moved to the end of the bottom layer's
  auto own_of = [&](const int64_t it) -> int64_t {
    const int64_t ti = TOP ? n_mine - 1 - it : ((ho.role == 1) ? (it + 1 < n_mine ? it + 1 : 0) : it);
    return (int64_t)bx + ti * G_;
  };
  // A tile's start used to be a chain of round trips with the matrix pipe idle: its first step (tile_k), its ids -> LDS, its first x rows.  The first two
  // are now requested while the tile BEFORE runs (round 6): the ids by LDS-DMA into the other id buffer, tile_k into one register.
  int ipar = 1;        // id buffer of the running tile (flipped at every tile start)
  int k0_req = 0;      // tile_k of the next tile to start (per lane; made uniform in retarget)
  auto request_tile = [&](const int64_t tl, const int buf) {
    if (a.tile_k) k0_req = a.tile_k[tl];
    if (BOTTOM) ids_stage_dma<256, MTR>(a.idx, a.N, T, a.F, a.nT, tl, ids + buf * IDS_BUF);
  };
  if (n_mine > 0) request_tile(ho.role == 2 ? ho.tile : own_of(0), 0);
  for (int64_t it = 0; it < n_mine; ++it) {
    const int64_t own_tile = own_of(it);
    const bool late = ho.role == 1 && own_tile == (int64_t)bx;    // the late piece of the pair's tile: starts from the stored state
    const bool early = ho.role == 2 && it == 0;   // the early piece of the pair's tile rides IN FRONT of this workgroup's first own tile:
    // its ho.d recurrent steps run through the same recurrent loop, and a rare block between two steps stores the state and re-points everything
    // at the own tile.  (Nothing here skips a block of MFMAs: every way of leaving the tile's step 0 out -- if / else, continue, a loop with an
    // opaque trip count -- made hipcc shuffle the 128 launch-persistent dW accumulators at every tile's end or hold the step's operand registers
    // live across it; a second instantiation of the step body in front of the loop was correct but ran as cold code: +10 us per piece.)
    // Per-piece bases.  Everything below counts steps from the tile's first executed one: tt = t - k0 in [0, Te).  The step index enters the
    // addresses only through these bases, so the step bodies see the same (compile-time 0 for the last step) offsets whether or not the tile
    // sits behind a prefix.
    int64_t tile, n0;
    int k0, Te;                  // k0: steps below it belong to the prefix
    const float* frag_tile;
    float* dx_tile;              // + (mt * T + tt) * 1024
    auto retarget = [&](const int64_t tl) {   // (tl is the tile the last request_tile named)
      tile = tl;
      n0 = tl * MTR;
      k0 = a.tile_k ? __builtin_amdgcn_readfirstlane(k0_req) : 0;
      Te = T - k0;
      frag_tile = a.save_frag + tl * NMT * frag_mt_stride + (int64_t)k0 * L * 4 * frag_unit + lane * 4;
      dx_tile = a.DX + (((tl * NMT) * T + k0) * 4 + j) * 256 + lane * 4;
    };
    retarget(early ? ho.tile : own_tile);
    auto frag_ptr = [&](int mt, int t, int l, int w, int plane) -> const float* {
      return frag_tile + mt * frag_mt_stride + ((int64_t)(t * L + l) * 4 + w) * frag_unit + plane * 256;
    };
    // Stage C operands, single-buffered: each register set is re-requested for the NEXT m-tile as soon as its
    // last consumer has issued (the factors after the VALU block, each B fragment after its 32 MFMAs), so the
    // loads have ~3k cycles of MFMA issue to land in.
    f32x4 P[6], up, bin[4], bhp[4];
    auto load_P = [&](int mt, int t) {
#pragma unroll
      for (int k = 0; k < 6; ++k) P[k] = *(const f32x4*)frag_ptr(mt, t, ly, j, k);
      if constexpr (!TOP && NMT == 1) {
        if (a.pipe == 2) {   // (layer pipeline: the top layer's workgroup of this tile is a step or so ahead, in this very launch)
          const unsigned* fl = a.pipe_flag + tile * MAXT_LDS + (k0 + t);
          const long long t0 = wall_clock64();
          while (__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.ho.epoch) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > HO_TIMEOUT_TICKS) {
              if (threadIdx.x == 0) __hip_atomic_store(a.ho.fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              break;
            }
          }
          asm volatile("" ::: "memory");
          up = ho_load4(dx_tile + (mt * T + t) * 1024);
        } else up = *(const f32x4*)(dx_tile + (mt * T + t) * 1024);
      } else if (!TOP) up = *(const f32x4*)(dx_tile + (mt * T + t) * 1024);
    };
    auto load_B = [&](int nt, int mt, int t, bool has_hp) {
      if (!BOTTOM) bin[nt] = *(const f32x4*)frag_ptr(mt, t, ly - 1, nt, 6);
      if (has_hp) bhp[nt] = *(const f32x4*)frag_ptr(mt, t - 1, ly, nt, 6);
    };
    // bottom layer: the dW_i2g B operands come from the x_t tile in LDS
    auto load_bin_lds = [&](int nt, int mt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bin[nt][r] = in_t[(mt * 16 + ag * 4 + r) * LDA + nt * 16 + arow];
    };
    const int tt_hi = late ? Te - 1 - ho.d : Te - 1;   // first step this piece runs
    f32x4 dc[NMT], dh[NMT];
    auto init_state = [&]() {
#pragma unroll
      for (int m = 0; m < NMT; ++m) {
        dc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        dh[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (TOP) {
          // nn.Linear(H,46) backward restricted to the selected column (OneModel.lua:275, MyOptimizer.lua:126): the recurrent
          // dh starts at dS[n] W_out[cid][:], and gW_out[cid][:] += dS[n] h_T[n][:] straight from the saved h fragment
          const f32x4 hf = *(const f32x4*)(frag_tile + m * frag_mt_stride + ((int64_t)((Te - 1) * L + ly) * 4 + j) * frag_unit + 6 * 256);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t n = n0 + m * 16 + ag * 4 + r;
            const float d = (n < a.N) ? a.dS[n] : 0.f;
            dh[m][r] = d * wout_c;
            gwo += d * hf[r];
            gbo += d;
          }
        }
      }
    };
    if (__builtin_expect(late, 0)) {   // (uniform) the other workgroup ran the steps above tt_hi
      ho_wait(a.ho, ho.slot);
#pragma unroll
      for (int m = 0; m < NMT; ++m) { dh[m] = ho_load4(ho_slot + m * 1024); dc[m] = ho_load4(ho_slot + (NMT + m) * 1024); }
    } else {
      init_state();
    }
    // next: the tile that starts after this one (-1: none)
    auto tile_prologue = [&](const int tt_first, const int64_t next) {
      ipar ^= 1;
      if (BOTTOM) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the tile's ids have landed (requested a tile ago)
        lds_barrier();  // ... everyone's have; the previous tile's x tile is fully consumed
        f32x4 nin[MTR * 16 / 256];
        gather_load_planes<256, MTR>(a, gsrc, tile, k0 + tt_first, ids + ipar * IDS_BUF, nin);
        gather_store<256, MTR>(in_t, nin);
      }
      if (next >= 0) request_tile(next, ipar ^ 1);   // (the other id buffer: last read by the tile before, behind the barrier above)
      if (BOTTOM) lds_barrier();
    };
    tile_prologue(tt_hi, early ? own_tile : (it + 1 < n_mine ? own_of(it + 1) : (int64_t)-1));
    TPROBE(1)  // tile prologue
    load_P(0, tt_hi);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) load_B(nt, 0, tt_hi, tt_hi > 0);

    // the step body is compiled twice (REC: t > 0, there is an h_{t-1} / c_{t-1}): no MFMA sits under a run-time condition
    auto step = [&](auto rec_tag, const int t) {  // t: step counted from the tile's first executed one
      constexpr bool REC = decltype(rec_tag)::value;
      // ---- C. cell backward + dW, one m-tile at a time ------------------------------------------------
      if (BOTTOM) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) load_bin_lds(nt, 0);
      }
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        // cell backward on the saved factors (elementwise on float4: register r <-> row 4 ag + r)
        f32x4 dhv = dh[mt];
        if (!TOP) dhv += up;
        const f32x4 dC = dc[mt] + dhv * P[4];
        f32x4 dA[4];
        dA[0] = dC * P[0];
        dA[1] = dC * P[1];
        dA[2] = dC * P[2];
        dA[3] = dhv * P[3];
        dc[mt] = dC * P[5];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dA_t[(mt * 16 + ag * 4 + r) * LDD + q * DH + j * 16 + arow] = dA[q][r];
          dbias[q] += (dA[q][0] + dA[q][1]) + (dA[q][2] + dA[q][3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        TPROBE(7)  // stage C, VALU part (incl. waiting for the factors)
        // the factors are dead: request the next m-tile's (first m-tile of step t-1 after the last one)
        if (mt < NMT - 1) load_P(mt + 1, t);
        else if (REC) load_P(0, t - 1);
        // dW += dA^T [in | h_prev]: MFMA (nt, r, q) uses k-slot ag <-> row mt*16 + 4ag + r
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              KPRN_MFMA_ACC_A(dwi[q][nt], dA[q][r], bin[nt][r]);
              if (REC) KPRN_MFMA_ACC_A(dwo[q][nt], dA[q][r], bhp[nt][r]);
            }
          __builtin_amdgcn_sched_barrier(0);
          if (mt < NMT - 1) { load_B(nt, mt + 1, t, REC); if (BOTTOM) load_bin_lds(nt, mt + 1); }
          else if (REC) load_B(nt, 0, t - 1, t > 1);
        }
        TPROBE(2)  // stage C, dW MFMAs (incl. waiting for the B fragments)
      }
      lds_barrier();
      if (!REC && k0 > 0) {
        // first executed step of a tile behind a prefix: the skipped steps see the same forward values on every row, so what
        // enters them is needed only as a sum over rows (k_prefix_bwd: dh_{k0-1} = W_o2g^T sum dA_{k0}, dW_o2g += sum dA_{k0} (x)
        // h_prefix).  Column sums of the dA tile, thread = column; kept out of the registers of the hot loop on purpose.
        float cs0 = 0.f, cs1 = 0.f;
#pragma unroll 8
        for (int row = 0; row < MTR; row += 2) { cs0 += dA_t[row * LDD + tid]; cs1 += dA_t[(row + 1) * LDD + tid]; }
        pg[k0 * PFB + tid] += cs0 + cs1;  // one owner thread per entry
      }
      TPROBE(3)  // mid barrier wait
      // bottom layer: x_{t-1} is requested here (latency hides under stage E) and lands after it
      f32x4 nin[MTR * 16 / 256];
      if (BOTTOM && REC) gather_load_planes<256, MTR>(a, gsrc, tile, k0 + t - 1, ids + ipar * IDS_BUF, nin);

      // ---- E. [dx | dh_prev] = dA * [W_i2g | W_o2g]; this wave: columns 16j..16j+15 of each ----------
      f32x4 ax[4], ah[4];
      if constexpr (NMT < 4) {   // (the pins below name all four)
#pragma unroll
        for (int q = NMT; q < 4; ++q) { ax[q] = f32x4{0.f, 0.f, 0.f, 0.f}; ah[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      }
      {
        const float* abase = dA_t + arow * LDD + ag * 4;
        if (REC) {
          // groups (S, mt): one A fragment feeds 8 MFMAs (4 k-slots x {dx, dh}); the next fragment is requested mid-group
          f32x4 apre = *(const f32x4*)(abase);
#pragma unroll
          for (int S = 0; S < 16; ++S) {
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
              const f32x4 a4 = apre;
              const int gn = S * NMT + mt + 1;  // next group
              if (S == 0) { KPRN_MFMA_Z(ax[mt], a4[0], wiT[S][0]); KPRN_MFMA_Z(ah[mt], a4[0], woT[S][0]); }
              else { KPRN_MFMA(ax[mt], a4[0], wiT[S][0]); KPRN_MFMA(ah[mt], a4[0], woT[S][0]); }
              KPRN_MFMA(ax[mt], a4[1], wiT[S][1]); KPRN_MFMA(ah[mt], a4[1], woT[S][1]);
              if (gn < 16 * NMT) apre = *(const f32x4*)(abase + (gn % NMT) * 16 * LDD + (gn / NMT) * 16);
              KPRN_MFMA(ax[mt], a4[2], wiT[S][2]); KPRN_MFMA(ah[mt], a4[2], woT[S][2]);
              KPRN_MFMA(ax[mt], a4[3], wiT[S][3]); KPRN_MFMA(ah[mt], a4[3], woT[S][3]);
            }
          }
        } else if constexpr (NMT == 1) {
          // t = 0, one m-tile: the k range is split over two accumulators (even / odd k-groups) so that consecutive MFMAs never chain on
          // one accumulator; they are added behind the drain below
          f32x4 ap0 = *(const f32x4*)(abase), ap1 = *(const f32x4*)(abase + 16);
#pragma unroll
          for (int S = 0; S < 16; S += 2) {
            const f32x4 a0 = ap0, a1 = ap1;
            if (S == 0) { KPRN_MFMA_Z(ax[0], a0[0], wiT[S][0]); KPRN_MFMA_Z(ax[1], a1[0], wiT[S + 1][0]); }
            else { KPRN_MFMA(ax[0], a0[0], wiT[S][0]); KPRN_MFMA(ax[1], a1[0], wiT[S + 1][0]); }
            KPRN_MFMA(ax[0], a0[1], wiT[S][1]); KPRN_MFMA(ax[1], a1[1], wiT[S + 1][1]);
            if (S + 2 < 16) { ap0 = *(const f32x4*)(abase + (S + 2) * 16); ap1 = *(const f32x4*)(abase + (S + 3) * 16); }
            KPRN_MFMA(ax[0], a0[2], wiT[S][2]); KPRN_MFMA(ax[1], a1[2], wiT[S + 1][2]);
            KPRN_MFMA(ax[0], a0[3], wiT[S][3]); KPRN_MFMA(ax[1], a1[3], wiT[S + 1][3]);
          }
        } else {
          // t = 0: no dh; two m-tiles share a group so that consecutive MFMAs never chain on one accumulator
          f32x4 ap0 = *(const f32x4*)(abase), ap1 = *(const f32x4*)(abase + 16 * LDD);
#pragma unroll
          for (int S = 0; S < 16; ++S) {
#pragma unroll
            for (int mp = 0; mp < 2; ++mp) {
              const f32x4 a0 = ap0, a1 = ap1;
              const int gn = S * 2 + mp + 1;
              if (S == 0) { KPRN_MFMA_Z(ax[2 * mp], a0[0], wiT[S][0]); KPRN_MFMA_Z(ax[2 * mp + 1], a1[0], wiT[S][0]); }
              else { KPRN_MFMA(ax[2 * mp], a0[0], wiT[S][0]); KPRN_MFMA(ax[2 * mp + 1], a1[0], wiT[S][0]); }
              KPRN_MFMA(ax[2 * mp], a0[1], wiT[S][1]); KPRN_MFMA(ax[2 * mp + 1], a1[1], wiT[S][1]);
              if (gn < 32) {
                ap0 = *(const f32x4*)(abase + ((gn & 1) * 2) * 16 * LDD + (gn >> 1) * 16);
                ap1 = *(const f32x4*)(abase + ((gn & 1) * 2 + 1) * 16 * LDD + (gn >> 1) * 16);
              }
              KPRN_MFMA(ax[2 * mp], a0[2], wiT[S][2]); KPRN_MFMA(ax[2 * mp + 1], a1[2], wiT[S][2]);
              KPRN_MFMA(ax[2 * mp], a0[3], wiT[S][3]); KPRN_MFMA(ax[2 * mp + 1], a1[3], wiT[S][3]);
            }
          }
        }
      }
      KPRN_MFMA_DRAIN();  // ax / ah are read by VALU and stores below
      KPRN_PIN_V4(ax);
      if (REC) KPRN_PIN_V4(ah);
      if constexpr (NMT == 1) { if (!REC) ax[0] += ax[1]; }
      if (BOTTOM && REC) gather_store<256, MTR>(in_t, nin);
      const bool compact = BOTTOM && a.DXe != nullptr && wcls == 1;   // (wave-uniform)
#pragma unroll
      for (int mt = 0; mt < NMT; ++mt) {
        if (REC) dh[mt] = ah[mt];  // already in the layout stage C of step t-1 reads
        // dx in fragment order: the layer below (or, bottom layer, the small-table gradient job) reloads it the same way,
        // 1 KiB per instruction; in place -- this thread read this very slot as `up` at the start of the step.
        // Rows past N: exact zeros (dA = 0 there).
        // (with the small tables' gradients formed below, nobody reads the type / relation slices of the bottom layer's dx: not stored)
        if constexpr (TOP && !BOTTOM && NMT == 1) {
          if (a.pipe == 1) ho_store4(dx_tile + (mt * T + t) * 1024, ax[mt]);   // (read by another CU of this launch: agent scope)
          else *(f32x4*)(dx_tile + (mt * T + t) * 1024) = ax[mt];
        } else
        if (!compact && !(BOTTOM && a.small_lds && wcls != 1)) *(f32x4*)(dx_tile + (mt * T + t) * 1024) = ax[mt];
      }
      if (compact) {
        // entity columns of the bottom layer: row-major for the gather-reduce over the occurrence index (16 lanes = 64 contiguous bytes)
        float* de_row = a.DXe + ((n0 + ag * 4) * T + k0 + t) * (int64_t)a.de + (j * 16 + arow - a.dt);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) de_row[(int64_t)(mt * 16 + r) * T * a.de] = ax[mt][r];
      }
      if (BOTTOM && a.small_lds && wcls != 1) {   // (wave-uniform)
        // nn.LookupTable backward of the type / relation tables (net/FeatureEmbedding.lua:112-121, 86): tables of at most 16 rows, so the gradient is a one-hot
        // product on the dx registers this wave holds anyway (its 16 columns ARE a type / relation slice): grad[v][col] += sum_rows [id(row) == v] dx[row][col]
        // = D += A B with A = one-hot [16 v x 4 k], B = dx [4 k x 16 col]; MFMA (mt, r) contracts over the rows {16 mt + 4 ag + r}: lane (ag, arow) holds
        // exactly its B element (ax[mt][r]) and its A element (the id of its own row against v = arow), the ids come from the id planes already in LDS.
        // 16 MFMAs per step on two of the four waves.  (Round 6: as a passenger job of the entity-gradient launch this re-read both slices of dx and the
        // ids from HBM -- 25 of that launch's 60 us; as ds_add_f32 into an LDS table it cost the launch 50 us.)  Rows past N carry dx = 0.
        const int32_t* idp = ids + ipar * IDS_BUF + ((wcls == 0) ? 0 : 2) * IDS_PLANE<MTR> + (k0 + t) * MTR + ag * 4;
        // (asm MFMAs with the accumulators held in architectural VGPRs, like every other MFMA of this kernel: left to hipcc -- the builtin -- the two
        // accumulators were parked in AGPRs that belong to the launch-persistent dW tiles and shuffled around every MFMA; the small-tile instantiation came
        // out numerically wrong)
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
          const int4 id4 = *(const int4*)(idp + mt * 16);   // rows 16 mt + 4 ag + r, raw (1-based)
          const float o0 = (id4.x - 1 == arow) ? 1.f : 0.f, o1 = (id4.y - 1 == arow) ? 1.f : 0.f;
          const float o2 = (id4.z - 1 == arow) ? 1.f : 0.f, o3 = (id4.w - 1 == arow) ? 1.f : 0.f;
          KPRN_MFMA_VV(sacc[0], o0, ax[mt][0]);   // (the macro carries the two wait states between the VALU result and the MFMA that reads it)
          KPRN_MFMA_VV(sacc[1], o1, ax[mt][1]);
          KPRN_MFMA_VV(sacc[0], o2, ax[mt][2]);
          KPRN_MFMA_VV(sacc[1], o3, ax[mt][3]);
        }
        KPRN_MFMA_DRAIN();   // the accumulators cross the loop back-edge: nothing may copy them before they have landed
        asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]));
      }
      TPROBE(4)  // stage E (dX MFMAs) + outputs
      if constexpr (TOP && !BOTTOM && NMT == 1) {
        if (a.pipe == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's dx stores have been acknowledged (everything older has long landed)
      }
      lds_barrier();  // dA_t free for reuse, x_{t-1} tile visible
      if constexpr (TOP && !BOTTOM && NMT == 1) {
        if (a.pipe == 1 && tid == 0) __hip_atomic_store(a.pipe_flag + tile * MAXT_LDS + (k0 + t), a.ho.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      TPROBE(5)  // end barrier
    };
    int tt_sw = early ? Te - ho.d : -1;   // the early piece's last step
    int tt_pub = -1;                       // ... and the step behind which its state is published (below)
    for (int tt = tt_hi; tt > 0; --tt) {
      step(std::true_type{}, tt);
      if (__builtin_expect(tt == tt_pub, 0)) { ho_publish(a.ho, ho.slot); tt_pub = -1; }
      if (__builtin_expect(tt == tt_sw, 0)) {   // (uniform, once per launch at most) the pair's tile goes on in its heavy workgroup; this one turns to its own first tile
#pragma unroll
        for (int m = 0; m < NMT; ++m) { ho_store4(ho_slot + m * 1024, dh[m]); ho_store4(ho_slot + (NMT + m) * 1024, dc[m]); }
        // (the flag follows one recurrent step later: by then the stores have been acknowledged -- publishing here would drain every load in
        // flight first -- and the pair's heavy workgroup, which runs at least one other tile first, asks for them several steps from now)
        retarget(own_tile);
        init_state();
        tile_prologue(Te - 1, n_mine > 1 ? own_of(1) : (int64_t)-1);
        load_P(0, Te - 1);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) load_B(nt, 0, Te - 1, Te > 1);
        tt = Te;   // (the loop goes on with the own tile's step Te - 1)
        tt_sw = -1;
        tt_pub = Te - 1;   // (Te >= 2: the own tile has a recurrent step)
      }
    }
    step(std::false_type{}, 0);
    if (k0 > 0) {
      // ... and dc_{k0-1} = sum over rows of the dc this step hands down
      float cs = 0.f;
#pragma unroll
      for (int m = 0; m < NMT; ++m) cs += (dc[m][0] + dc[m][1]) + (dc[m][2] + dc[m][3]);
      cs += __shfl_xor(cs, 16, 64);
      cs += __shfl_xor(cs, 32, 64);
      if (ag == 0) pg[k0 * PFB + 4 * DH + j * 16 + arow] += cs;  // one owner lane per entry
    }
  }
  if (a.tile_k) {
    lds_barrier();
    for (int c = PFB + tid; c < (KCAP + 1) * PFB; c += 256) {
      const float v = pg[c];
      if (v != 0.f) unsafeAtomicAdd(a.PG + c, v);
    }
  }
  if (BOTTOM && a.small_lds && wcls != 1) {
    const f32x4 sg = sacc[0] + sacc[1];
    const int V = (wcls == 0) ? a.Vt : a.Vr, width = (wcls == 0) ? a.dt : a.dr;
    float* g = ((wcls == 0) ? a.gWt : a.gWr) + (j * 16 + arow - ((wcls == 0) ? 0 : a.dt + a.de));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = 4 * ag + i;
      if (v < V && sg[i] != 0.f) unsafeAtomicAdd(g + (int64_t)v * width, sg[i]);
    }
  }

  // ---- flush the launch-persistent accumulators: plain coalesced stores into this workgroup's slab;
  // k_reduce_partials sums the slabs (device-scope atomics from 256 workgroups onto the same 128 KB
  // were measured slower: they execute at the memory side, the per-XCD L2s are not coherent)
  KPRN_MFMA_DRAIN();
#pragma unroll
  for (int q = 0; q < 4; ++q) KPRN_PIN_A8(dwi[q], dwo[q]);
  {
    float* pw = a.part + (int64_t)bx * PART;
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
  if (TOP) {
    float v = gwo;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (ag == 0) unsafeAtomicAdd(a.gWout_row + j * 16 + arow, v);
    float bsum = gbo;
    bsum += __shfl_xor(bsum, 16, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    if (j == 0 && lane == 0) unsafeAtomicAdd(a.gbout_c, bsum);
  }
  TPROBE(6)  // flush
  if (KPRN_PROBES_ON && a.timing && tid == 0) {
    for (int k = 0; k < 8; ++k) a.timing[(int64_t)bx * 8 + k] = tacc[k];
  }
}

template <bool BOTTOM, bool TOP, int NMT>
__global__ __launch_bounds__(256, 1) void k_lstm_bwd(BwdArgs a) { bwd_body<BOTTOM, TOP, NMT>(a, (int)blockIdx.x, (int)gridDim.x); }

// Two layers, small batches: BOTH layers' BPTT in one launch, workgroups [0, g) the top layer's, [g, 2 g) the bottom layer's, all resident at once (the host
// launches this only when 2 g <= the CUs: a bottom-layer workgroup that waits must never keep a top-layer one off the chip).  The bottom layer runs one step
// behind the top layer (BwdArgs.pipe): a 16-row tile's BPTT is a latency chain per layer, and the two chains overlap instead of following each other.
template <int NMT>
__global__ __launch_bounds__(256, 1) void k_lstm_bwd_dual(BwdArgs top, BwdArgs bottom, int g) {
  if ((int)blockIdx.x < g) bwd_body<false, true, NMT>(top, (int)blockIdx.x, g);
  else bwd_body<true, false, NMT>(bottom, (int)blockIdx.x - g, g);
}

// ---------------------------------------------------------------------------------------------------------
// nn.LookupTable backward for the three tables (FeatureEmbedding.lua:29,41-49,86 + CAddTable over type slots):
// scatter-add of the bottom layer's dx, duplicates accumulating.  Input = the fused backward's fragment-order dx
// (block (m-tile, t, wave w): lane (ag, arow), register r <-> row 4 ag + r, col 16 w + arow).  One workgroup per
// 64-path tile, t by t:
//   dx tile -> LDS row-major; rows of the tile with the same entity id (every pad step hits ONE row, a pair's
//   user / item repeat across its P paths) are folded onto their leader row in LDS, then ONE L2 atomic row per
//   distinct id; the two tiny tables accumulate in LDS for the whole workgroup and are flushed once.
// HBM-bound: reads N T D floats once; algorithmic bytes per path = T (D + F) 4.
struct ScatArgs {
  const int32_t* idx; int64_t N; int T, F, nT;
  int dt, de, dr, Vt, Vr;
  const float* DX;        // [(Npad/16)][T][4][64][4]
  float *gWt, *gWe, *gWr;
  const int32_t* tile_k;    // identical-prefix plan: steps below tile_k[tile] are not executed (nullable)
  int do_small, do_entity;  // which tables this launch handles
  float* part_small;      // [grid][Vt*dt + Vr*dr] per-workgroup partial small tables; null: tables too big for LDS -> global atomics
  int64_t n_tiles;
};

__global__ __launch_bounds__(256) void k_embed_scatter_frag(ScatArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dxt = lds;                                   // [64][LDA]
  int32_t* ids = (int32_t*)(dxt + MT * LDA);          // [64][4]: first type, entity, relation (0-based), leader
  float* small_g = (float*)(ids + MT * 4);            // [Vt*dt + Vr*dr] (if it fits)
  const int tid = threadIdx.x, lane = tid & 63, w0 = tid >> 6;
  const int arow = lane & 15, ag = lane >> 4;
  const int T = a.T, D = a.dt + a.de + a.dr;
  const int n_small = a.Vt * a.dt + a.Vr * a.dr;
  const bool small_in_lds = a.part_small != nullptr && a.do_small;
  if (small_in_lds) for (int i = tid; i < n_small; i += 256) small_g[i] = 0.f;
  const int e0 = a.dt, e1 = a.dt + a.de;
  const int col = tid & 63, rg = tid >> 6;  // combine phase: this thread owns one column, rows rg, rg+4, ...
  // work items = (tile, t): independent, so the grid is sized for occupancy (latency-bound: load -> LDS -> atomics)
  const int64_t n_items = a.n_tiles * T;
  for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int64_t tile = item / T;
    const int t = (int)(item - tile * T);
    const int64_t n0 = tile * MT;
    if (a.tile_k && t < a.tile_k[tile]) continue;  // (uniform) the prefix backward owns these positions
    // (1) this step's dx tile: 16 blocks of 1 KiB, wave w0 takes blocks (mt, w) = (k, w0)
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *(const f32x4*)(a.DX + (((tile * 4 + k) * T + t) * 4 + w0) * 256 + lane * 4);
    int my_ids[4] = {0, 0, 0, -1};
    if (tid < MT) {
      int64_t n = n0 + tid;
      const bool valid = n < a.N;
      if (!valid) n = a.N - 1;
      const int32_t* f = a.idx + (n * T + t) * a.F;
      my_ids[0] = f[a.F - a.nT - 2] - 1;
      my_ids[1] = f[a.F - 2] - 1;
      my_ids[2] = f[a.F - 1] - 1;
      my_ids[3] = valid ? tid : -1;
    }
    __syncthreads();  // previous item's readers are done with dxt / ids
    if (tid < MT) {
#pragma unroll
      for (int k = 0; k < 4; ++k) ids[tid * 4 + k] = my_ids[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) dxt[(k * 16 + ag * 4 + r) * LDA + w0 * 16 + arow] = v[k][r];
    __syncthreads();
    {
      // leader = first row of the tile with the same entity id
      int ld = -1;
      if (tid < MT && ids[tid * 4 + 3] >= 0) {
        const int e = ids[tid * 4 + 1];
        ld = tid;
        for (int r2 = tid - 1; r2 >= 0; --r2) if (ids[r2 * 4 + 1] == e) ld = r2;
      }
      __syncthreads();
      if (tid < MT) ids[tid * 4 + 3] = ld;
      __syncthreads();
    }
    // (2) fold follower rows onto their leader (entity slice); types / relations into the small tables
    if (col < D) {
      for (int k = 0; k < 16; ++k) {
        const int row = rg + 4 * k;
        const int ld = ids[row * 4 + 3];
        if (ld < 0) continue;
        const float val = dxt[row * LDA + col];
        if (col < e0) {
          if (!a.do_small) continue;
          const int32_t* f = a.idx + ((n0 + row) * T + t) * a.F;
          for (int kk = 0; kk < a.nT; ++kk) {
            const int rr = (kk == 0) ? ids[row * 4 + 0] : f[a.F - a.nT - 2 + kk] - 1;
            if (small_in_lds) lds_atomic_add(&small_g[rr * a.dt + col], val);
            else unsafeAtomicAdd(a.gWt + (int64_t)rr * a.dt + col, val);
          }
        } else if (col < e1) {
          if (a.do_entity && ld != row) lds_atomic_add(&dxt[ld * LDA + col], val);
        } else {
          if (!a.do_small) continue;
          const int rr = ids[row * 4 + 2];
          if (small_in_lds) lds_atomic_add(&small_g[a.Vt * a.dt + rr * a.dr + (col - e1)], val);
          else unsafeAtomicAdd(a.gWr + (int64_t)rr * a.dr + (col - e1), val);
        }
      }
    }
    __syncthreads();
    // (3) leaders add their combined entity slice to the gradient table (fire-and-forget atomics)
    if (col >= e0 && col < e1 && a.do_entity) {
      for (int k = 0; k < 16; ++k) {
        const int row = rg + 4 * k;
        if (ids[row * 4 + 3] == row) unsafeAtomicAdd(a.gWe + (int64_t)ids[row * 4 + 1] * a.de + (col - e0), dxt[row * LDA + col]);
      }
    }
  }
  // the small tables leave as one plain slab per workgroup; k_reduce_small sums the slabs (thousands of
  // workgroups' atomics onto the same few cache lines serialise at the memory side)
  if (small_in_lds) {
    __syncthreads();
    for (int i = tid; i < n_small; i += 256) a.part_small[(int64_t)blockIdx.x * n_small + i] = small_g[i];
  }
}

// gWt | gWr += sum over the scatter workgroups' slabs: one workgroup per 16 table entries, 16 slab lanes each
__global__ __launch_bounds__(256) void k_reduce_small(const float* __restrict__ part, int nslab, int n_small, int nt_small, float* __restrict__ gWt, float* __restrict__ gWr) {
  __shared__ float red[16][17];
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + e;
  float acc = 0.f;
  if (i < n_small) for (int s = sl; s < nslab; s += 16) acc += part[(int64_t)s * n_small + i];
  red[sl][e] = acc;
  __syncthreads();
  if (sl == 0 && i < n_small) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][e];
    if (i < nt_small) unsafeAtomicAdd(gWt + i, v); else unsafeAtomicAdd(gWr + (i - nt_small), v);
  }
}

// gW_i2g / gW_o2g / gb += sum over workgroup slabs, every layer in one launch (blockIdx.z = layer).  Normally this job rides in
// the entity-gradient launch (bidx::entity_grad, SlabReduce); this kernel is the stand-alone form.
__global__ void k_reduce_partials(bidx::SlabReduce a) { bidx::slab_reduce_block(a, blockIdx.x, blockIdx.y, blockIdx.z); }

// WT[n][k] = W[k][n] for the [256][64] weights of every layer in one launch (blockIdx.y = matrix)
struct TransArgs { const float* W[4]; float* WT[4]; };
__global__ void k_transpose_256x64(TransArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over 64*256 outputs
  if (i >= 64 * 256) return;
  const int n = i >> 8, k = i & 255;
  a.WT[blockIdx.y][i] = a.W[blockIdx.y][k * 64 + n];
}

// ---- host side ----
// the W^T copies the backward kernels read are stale: hand out the job (and consider it done -- the caller launches it)
bool transpose_job(kprn_handle* h, kk::TransposeJob* tj) {
  State* s = st(h);
  if (!s->wt_dirty) return false;
  const int L = h->cfg.L;
  if (!s->WT) HIP_TRY(kprn_dev_malloc((void**)&s->WT, (size_t)2 * 2 * 64 * 256 * sizeof(float)));
  tj->n = 2 * L;
  for (int m = 0; m < 4; ++m) {
    const int l = (m >> 1) < L ? (m >> 1) : 0;
    tj->W[m] = h->dense + ((m & 1) ? h->layer[l].Wo : h->layer[l].Wi);
    tj->WT[m] = s->WT + (size_t)(l * 2 + (m & 1)) * 64 * 256;
  }
  s->wt_dirty = false;
  return true;
}

bool bwd_supported(const kprn_handle* h, int T) { return fwd_supported(h, T); }

template <bool BOTTOM, bool TOP, int NMT = 4>
static void launch_bwd(kprn_handle* h, const BwdArgs& a, int grid) {
  size_t lds_bytes = (size_t)MT * LDD * sizeof(float);
  if (BOTTOM) lds_bytes += (size_t)MT * LDA * sizeof(float) + 2 * (MT * MAXT_LDS * 4) * sizeof(int32_t);
  lds_bytes += (size_t)(KCAP + 1) * PFB * sizeof(float);
  static PerDeviceOnce attr_done;  // one per template instantiation (the call is host time in front of every launch otherwise)
  if (attr_done.need()) HIP_TRY(hipFuncSetAttribute((const void*)k_lstm_bwd<BOTTOM, TOP, NMT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipLaunchKernelGGL((k_lstm_bwd<BOTTOM, TOP, NMT>), dim3(grid), dim3(256), lds_bytes, h->stream, a);
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
    if (s->DX) hipFree(s->DX);
    const int64_t cn = std::max<int64_t>(N, s->cap_Nb);
    const int ct = std::max(T, s->cap_Tb);
    HIP_TRY(kprn_dev_malloc((void**)&s->DX, (size_t)ct * (cn + 2 * MT) * DH * sizeof(float)));
    if (s->DXe) hipFree(s->DXe);
    HIP_TRY(kprn_dev_malloc((void**)&s->DXe, (size_t)(ct * (cn + 2 * MT) + KCAP) * c.de * sizeof(float)));
    s->cap_Nb = cn; s->cap_Tb = ct;
  }
  if (!s->part) HIP_TRY(kprn_dev_malloc((void**)&s->part, (size_t)2 * s->num_cu * PART * sizeof(float)));  // one slab set per layer
  static const bool want_timing = KPRN_DEV_ENV("KPRN_TIMING") != nullptr;
  if (want_timing && !s->timing) HIP_TRY(kprn_dev_malloc((void**)&s->timing, (size_t)s->num_cu * 12 * sizeof(unsigned long long)));
  if (s->wt_dirty) {  // (normally done already: the transposes ride in the loss-stage launch, transpose_job())
    ProfScope ps(h, "weight_transpose");
    kk::TransposeJob tj;
    transpose_job(h, &tj);
    TransArgs ta;
    for (int m = 0; m < 4; ++m) { ta.W[m] = tj.W[m < tj.n ? m : 0]; ta.WT[m] = tj.WT[m < tj.n ? m : 0]; }
    hipLaunchKernelGGL(k_transpose_256x64, dim3(64, tj.n), dim3(256), 0, strm, ta);
    HIP_TRY(hipGetLastError());
  }
  float* gd = h->g_dense;
  // small batches ran the forward on tiles of one 16-row m-tile (fused::small_tiles): the same here -- the saves and dx are laid out per
  // 16-row block either way
  const bool small = small_tiles(h, N, b->tile_k != nullptr);
  const int64_t n_tiles64 = (N + MT - 1) / MT;
  const int64_t n_tiles = small ? (N + 15) / 16 : n_tiles64;
  const int nmt = small ? 1 : 4;
  const int grid = (int)std::min<int64_t>(n_tiles, (int64_t)s->num_cu);
  bool have_r1 = false, reduced = false;
  std::unique_ptr<ProfScope> bwd_scope;
  bidx::SlabReduce ra;
  BwdArgs pipe_top;
  for (int l = L - 1; l >= 0; --l) {
    BwdArgs a;
    a.pipe = 0; a.pipe_flag = nullptr;
    a.idx = b->idx_s ? b->idx_s : b->idx; a.N = N; a.T = T; a.F = b->F; a.nT = c.num_types;
    a.tile_k = b->tile_k; a.PG = s->PG + (size_t)l * (KCAP + 1) * PFB;
    a.Wt = h->dense + h->off_Wt; a.We = h->We; a.Wr = h->dense + h->off_Wr;
    a.dt = c.dt; a.de = c.de; a.dr = c.dr; a.Vt = c.Vt; a.Vr = c.Vr;
    a.L = L; a.layer = l;
    a.WiT = s->WT + (size_t)(l * 2 + 0) * 64 * 256; a.WoT = s->WT + (size_t)(l * 2 + 1) * 64 * 256;
    a.save_frag = s->save_frag; a.dS = h->ws.dS; a.wout_row = h->dense + h->off_outW + (int64_t)cid * DH; a.DX = s->DX;
    a.gWout_row = gd + h->off_outW + (int64_t)cid * DH; a.gbout_c = gd + h->off_outb + cid; a.Npad = n_tiles64 * MT;
    a.gWi = gd + h->layer[l].Wi; a.gbi = gd + h->layer[l].bi; a.gWo = gd + h->layer[l].Wo;
    a.gWt = gd + h->off_Wt; a.gWe = h->g_We; a.gWr = gd + h->off_Wr;
    a.n_tiles = n_tiles;
    a.part = s->part + (size_t)l * s->num_cu * PART; a.timing = s->timing;
    a.dbg = kprn_dbg_mask();
    a.ho = handover_args(h, grid);
    const bool bottom = (l == 0), top = (l == L - 1);
    // type / relation gradients: a passenger job of the entity-gradient launch when the shapes allow (one type slot, slices in
    // 16-column blocks, tables of at most 16 rows), else the general scatter kernel
    const bool small_job = c.num_types == 1 && (c.dt % 16) == 0 && (c.de % 16) == 0 && (c.dr % 16) == 0 && c.Vt <= 16 && c.Vr <= 16 && !(a.dbg & 8);
    const bool have_index = b->key_sorted != nullptr && !(a.dbg & 16);
    // with the index, the entity columns of the bottom layer's dx leave the kernel row-major for the gather-reduce
    s->DXe_on = have_index && !(a.dbg & 1);
    a.DXe = (bottom && s->DXe_on) ? s->DXe : nullptr;
    // ... or, round 6, formed inside the bottom layer's launch (one-hot MFMAs on the dx registers; option "fused_small_tables")
    // (64-path tiles only: at small batches the passenger job hides in a launch that is latency-bound anyway, and the 16 extra MFMAs per step do not)
    const bool small_lds = bottom && small_job && have_index && !(a.dbg & 1) && h->fused_small_tables && !small;
    a.small_lds = small_lds ? 1 : 0;
    {
      // one event pair around the L back-to-back launches of the family (an event pair costs ~4 us of stream time)
      if (top) { bwd_scope.reset(new ProfScope(h, "lstm_fused_bwd")); bwd_scope->launches = L; }
      // small batches, two layers, both layers' workgroups fit the chip at once: ONE launch, the bottom layer a step behind the top layer (k_lstm_bwd_dual)
      const bool pipe = small && L == 2 && 2 * grid <= s->num_cu && h->bwd_pipe && !s->timing && a.ho.fault != nullptr;
      if (pipe && top) {
        if (!s->pipe_flag) {
          HIP_TRY(kprn_dev_malloc((void**)&s->pipe_flag, (size_t)s->num_cu * MAXT_LDS * sizeof(unsigned)));
          HIP_TRY(hipMemsetAsync(s->pipe_flag, 0, (size_t)s->num_cu * MAXT_LDS * sizeof(unsigned), strm));
        }
        a.pipe = 1; a.pipe_flag = s->pipe_flag;
        pipe_top = a;   // (launched with the bottom layer's, below)
        bwd_scope->launches = 1;
      } else if (pipe && bottom) {
        a.pipe = 2; a.pipe_flag = s->pipe_flag; a.ho = pipe_top.ho;   // (one epoch for the pair)
        size_t lds_bytes = (size_t)MT * LDD * sizeof(float) + (size_t)MT * LDA * sizeof(float) + 2 * (MT * MAXT_LDS * 4) * sizeof(int32_t) + (size_t)(KCAP + 1) * PFB * sizeof(float);
        static PerDeviceOnce once;
        if (once.need()) HIP_TRY(hipFuncSetAttribute((const void*)k_lstm_bwd_dual<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL((k_lstm_bwd_dual<1>), dim3(2 * grid), dim3(256), lds_bytes, h->stream, pipe_top, a, grid);
        HIP_TRY(hipGetLastError());
      } else
      if (small) { if (bottom) launch_bwd<true, false, 1>(h, a, grid); else launch_bwd<false, true, 1>(h, a, grid); }   // (L == 2)
      else if (bottom && top) launch_bwd<true, true>(h, a, grid);
      else if (bottom) launch_bwd<true, false>(h, a, grid);
      else if (top) launch_bwd<false, true>(h, a, grid);
      else launch_bwd<false, false>(h, a, grid);
      if (bottom || s->timing) bwd_scope.reset();
      if (bottom && h->after_bptt_hook) h->after_bptt_hook(h);   // (a deferred part of the scoring pass: beside the tail below, kprn_internal.h score_rest_in_backward)
    }
    if (bottom) have_r1 = prefix_backward(h, b, n_tiles64);  // the skipped steps of every layer; leaves their dx sums in DX's virtual tile
    if (bottom) {
      for (int q = 0; q < 2; ++q) {
        const int ll = q < L ? q : 0;
        ra.part[q] = s->part + (size_t)ll * s->num_cu * PART;
        ra.gWi[q] = gd + h->layer[ll].Wi; ra.gWo[q] = gd + h->layer[ll].Wo; ra.gbi[q] = gd + h->layer[ll].bi;
      }
      ra.nslab = grid; ra.n_elem = PART; ra.L = L; ra.ny = 16;
      ra.r1 = s->r1; ra.kmax = have_r1 ? b->h_kmax : 0; ra.kcap = KCAP; ra.r1_stride = R1; ra.G = 4; ra.H = DH;
    }
    if (bottom && have_index && !(a.dbg & 1)) {
      // the weight-gradient slab reduce and the small-table gradients ride along: independent, latency-bound jobs in one launch
      ProfScope ps(h, "entity_grad+dw_reduce");
      bidx::SmallGrad sg;
      sg.DX = s->DX; sg.idx = b->idx_s ? b->idx_s : b->idx; sg.tile_k = b->tile_k; sg.N = N; sg.n_mtiles = n_tiles * nmt; sg.T = T; sg.F = b->F;
      sg.nT = c.num_types; sg.dt = c.dt; sg.de = c.de; sg.dr = c.dr; sg.Vt = c.Vt; sg.Vr = c.Vr; sg.gWt = a.gWt; sg.gWr = a.gWr; sg.nblocks = 4 * s->num_cu;
      { static const int sgb = KPRN_DEV_ENV("KPRN_SG_BLOCKS") ? atoi(KPRN_DEV_ENV("KPRN_SG_BLOCKS")) : 0; if (sgb > 0) sg.nblocks = sgb; }   // (measurement)
      bidx::entity_grad(strm, s->DXe, /*compact entity slice=*/2, b->key_sorted, b->pos_sorted, b->n_index, N, T, DH, c.dt, c.de, c.Ve, a.gWe, &ra,
                        (small_job && !a.small_lds) ? &sg : nullptr);
      reduced = true;
    }
    const bool small_in_kernel = small_job && have_index;   // (handled above)
    if (bottom && !(a.dbg & 1) && (!small_in_kernel || !have_index)) {
      ProfScope ps(h, "embed_scatter");
      ScatArgs sa;
      sa.idx = b->idx_s ? b->idx_s : b->idx; sa.tile_k = b->tile_k; sa.N = N; sa.T = T; sa.F = b->F; sa.nT = c.num_types;
      sa.dt = c.dt; sa.de = c.de; sa.dr = c.dr; sa.Vt = c.Vt; sa.Vr = c.Vr;
      sa.DX = s->DX; sa.gWt = a.gWt; sa.gWe = a.gWe; sa.gWr = a.gWr; sa.n_tiles = n_tiles64;
      sa.do_small = small_in_kernel ? 0 : 1; sa.do_entity = have_index ? 0 : 1;
      const int n_small = c.Vt * c.dt + c.Vr * c.dr;
      const bool small_fits = n_small <= 4096;
      const size_t lds_b = (size_t)MT * LDA * sizeof(float) + MT * 4 * sizeof(int32_t) + (size_t)(small_fits ? n_small : 0) * sizeof(float);
      const int sgrid = (int)std::min<int64_t>(n_tiles64 * T, (int64_t)s->num_cu * 8);
      const bool slabs = small_fits && sa.do_small;
      if (slabs && (!s->part_small || s->part_small_n < n_small)) {
        HIP_TRY(hipStreamSynchronize(strm));
        if (s->part_small) hipFree(s->part_small);
        HIP_TRY(kprn_dev_malloc((void**)&s->part_small, (size_t)s->num_cu * 8 * n_small * sizeof(float)));
        s->part_small_n = n_small;
      }
      sa.part_small = small_fits ? s->part_small : nullptr;
      hipLaunchKernelGGL(k_embed_scatter_frag, dim3(sgrid), dim3(256), lds_b, strm, sa);
      if (slabs) hipLaunchKernelGGL(k_reduce_small, dim3((n_small + 15) / 16), dim3(256), 0, strm, s->part_small, sgrid, n_small, c.Vt * c.dt, sa.gWt, sa.gWr);
      HIP_TRY(hipGetLastError());
    }
    if (s->timing) {
      HIP_TRY(hipStreamSynchronize(strm));
      std::vector<unsigned long long> tb((size_t)grid * 8);
      HIP_TRY(hipMemcpy(tb.data(), s->timing, tb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double sum[8] = {0};
      for (int g = 0; g < grid; ++g) for (int k = 0; k < 8; ++k) sum[k] += (double)tb[(size_t)g * 8 + k];
      {   // (per-workgroup totals: the longest, and the mean of the first 21 / of the others -- the heavy and the light workgroups of the bench's tile mix)
        double tmax = 0, th = 0, tl = 0; int nh = 0, nl = 0, imax = 0;
        for (int g = 0; g < grid; ++g) {
          double tot = 0; for (int k = 0; k < 8; ++k) tot += (double)tb[(size_t)g * 8 + k];
          if (tot > tmax) { tmax = tot; imax = g; }
          if (g < 21) { th += tot; ++nh; } else { tl += tot; ++nl; }
        }
        if (getenv("KPRN_TIMING_DUMP")) {   // (measurement build only) every workgroup's total, in workgroup order
          fprintf(stderr, "[kprn timing dump] bwd layer %d:", l);
          for (int g = 0; g < grid; ++g) { double tot = 0; for (int k = 0; k < 8; ++k) tot += (double)tb[(size_t)g * 8 + k]; fprintf(stderr, " %.0f", tot / 1000.0); }
          fprintf(stderr, "\n");
        }
        fprintf(stderr, "[kprn timing] bwd layer %d per-WG total: max %.0f (WG %d) mean of WG 0..20 %.0f, of the others %.0f; tile-prologue of WG 0: %.0f, WG 100: %.0f\n", l, tmax, imax,
                th / (nh ? nh : 1), tl / (nl ? nl : 1), (double)tb[1], grid > 100 ? (double)tb[100 * 8 + 1] : 0.0);
      }
      fprintf(stderr, "[kprn timing] bwd layer %d N=%lld grid=%d avg cycles/WG: prologue %.0f tile-prologue %.0f stageC-valu %.0f stageC-mfma %.0f midbar %.0f stageE %.0f endbar %.0f flush %.0f\n",
              l, (long long)N, grid, sum[0] / grid, sum[1] / grid, sum[7] / grid, sum[2] / grid, sum[3] / grid, sum[4] / grid, sum[5] / grid, sum[6] / grid);
    }
  }
  if (!reduced) {
    ProfScope ps(h, "dw_reduce");
    hipLaunchKernelGGL(k_reduce_partials, dim3((PART + 255) / 256, ra.ny, L), dim3(256), 0, strm, ra);
    HIP_TRY(hipGetLastError());
  }
}

void params_changed(kprn_handle* h) {
  if (!h->fused_state) return;
  State* s = (State*)h->fused_state;
  s->wt_dirty = true;
  s->pf_batch = -1;  // the prefix table is a function of the parameters
  s->mc_dirty = true;
}

void release(kprn_handle* h) {
  State* s = (State*)h->fused_state;
  if (!s) return;
  for (float* p : {s->save_frag, s->WT, s->DX, s->DXe, s->part, s->part_small, s->pfb, s->pfs, s->pfx, s->PG, s->r1, s->mc_bias, s->mc_hseq[0], s->mc_hseq[1],
                   s->ho_state[0], s->ho_state[1]}) if (p) hipFree(p);
  for (unsigned* p : {s->ho_flag[0], s->ho_flag[1], s->pipe_flag}) if (p) hipFree(p);
  if (s->mc_wsp) hipFree(s->mc_wsp);
  if (s->timing) hipFree(s->timing);
  delete s;
  h->fused_state = nullptr;
}

}  // namespace fused
