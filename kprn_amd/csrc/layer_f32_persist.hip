// One recurrent layer of the wide fp32 shapes as ONE persistent launch: "d = 64" reading B (D = H = 192, L = 2: one launch per layer) and
// run_scripts/config.sh as shipped (rnn, D = 200, H = 250).  gfx950 only.
//
// Stands for   nn.Sequencer(nn.FastLSTM(D, H))                                release/songPathRnn/model/OneModel.lua:236,268-274
//        and   nn.Sequencer(nn.Recurrence(nn.MaskZero(act(i2h x + h2h h))))   release/songPathRnn/model/OneModel.lua:240-266,268-273
// for all T steps of a 64-path tile, replacing the T per-step launches of gemm_tiled.hip (lstm_step / rnn_step: 0.61 / 0.47 of the fp32-MFMA peak).
// What those launches spend beside their MFMAs is operand staging -- both operands global -> registers -> masks -> ds_write -> ds_read every 32 k,
// on VALU slots that fp32 MFMA cannot overlap -- and state through HBM between steps.  Here:
//   * A workgroup (4 waves, one per SIMD) owns a 64-path tile for t = 0 .. T-1.  [x_t | h_{t-1}] sits in LDS (row pitch KX + KH + 4 floats); h_t is
//     written there by the cell, c_t stays in registers (FastLSTM), x_{t+1} is fetched into registers under the step's MFMAs.
//   * The product is taken TRANSPOSED on v_mfma_f32_16x16x4_f32 (weights = first operand): a lane then holds four consecutive hidden units of one
//     path, all four gates of a unit in the same lane (wave w owns units 64 c + 16 w .. + 15 of chunk c, n-tile q = gate q), so the cell is
//     lane-local.  rnn: the wave's four n-tiles are four groups of 16 units (64 w + 16 q ..), one chunk covers 256 units.
//   * Weights never touch the VALU: each wave streams ITS rows L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: a lane's 16 bytes -- four consecutive
//     k of its weight row -- land in the lane's own slot of a 4 KB group buffer) 2-3 groups of 16 k ahead, and reads them back as B fragments with
//     one ds_read_b128.  Per group of 16 k a wave issues 4 DMA, 8 ds_read_b128 and 64 MFMAs (2 048 matrix cycles); the fragments of group g + 1 are
//     read under the MFMAs of group g.  K is padded to whole pairs of groups (pad columns of the LDS tile are zero; a weight row read past its end
//     delivers the next row's finite values, times zero).
//   * Saves (training) in the generic backward's own layouts -- gates [T][N][4H], c, h [T][N][H] (rnn: pre-activations [T][N][H]) -- so the
//     backward of kprn_api.hip runs unchanged.
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "kprn_internal.h"

namespace lp32 {

// compile-time loop: the body sees its index as a constant
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f32x4 f32x4u __attribute__((aligned(4)));

constexpr int ROWS = 64, NTHR = 256;
constexpr int XPMAX = 16;   // 16-byte pieces of a tile row one thread prefetches (Din <= 256)
constexpr int GB = 4096;    // bytes of one wave's weight group in the DMA ring: 4 n-tiles x 64 lanes x 16 B
constexpr int BS_FLOATS = 4 * 256;   // the forward's bias planes in LDS

struct LArgs {
  const float* in; int64_t N; int T; int Din; int H;
  const float* Wi; const float* Wo; const float* bi; const float* bo;
  float* hs; float* cs; float* act; const float* mask;
  const float* Wc; const float* Uc; const float* bc;   // gru: the candidate's maps (c_i2h weight / bias, c_h2h)
  int relu, write_all_h;
  int64_t tiles;
  int KX, KH, PITCH, R;
};

__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }
// One LDS-DMA instruction: lane l's 16 bytes land at lds_dst + 16 l (lds_dst wave-uniform).  Inline asm: hipcc does not count it (its own counted
// waits only get stricter); the waits for the ring are written by hand below.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// gate functions at fp32 accuracy (gemm_tiled.hip has the error analysis: absolute error of sigma / tanh <= 1.5e-7)
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
__device__ __forceinline__ void store4(float* __restrict__ p, const f32x4 v, int nv) {
  if (nv >= 4) *(f32x4u*)p = v;
  else {
#pragma unroll
    for (int r = 0; r < 3; ++r) if (r < nv) p[r] = v[r];
  }
}

// CELL 0: FastLSTM (gate rows i, g, f, o of W: row q H + u), NCH chunks of 64 hidden units.  CELL 1: rnn, one chunk of 256 units (NCH = 1).
// CELL 2: nn.GRU (OneModel.lua:237-238), NCH chunks of 64 units in the FastLSTM lane layout.  A step has two DEPENDENT products -- [r; z] = sigmoid(i2g x + o2g h'),
// then n = tanh(c_i2h x + c_h2h (r * h')) -- so it runs as (NCH + 1) / 2 passes over [x_t | h_{t-1}] whose four n-tiles are the r and z rows of TWO chunks,
// the cell's first half (r, z, r * h' -- h' is still in this lane's registers from the step before), a barrier pair around r * h' taking h_{t-1}'s place in
// the LDS tile, and ONE pass whose four n-tiles are the candidate rows of the four chunks over [x_t | r * h']: every MFMA of the step does wanted work at H > 192,
// the weight stream and its ring are the other cells'.  Saves in the per-step pipeline's record: act [T][N][4H] = [r | z | n | r * h'], h [T][N][H].
template <int CELL, int NCH, bool SAVE>
__global__ __launch_bounds__(NTHR, 1) void k_layer(LArgs a) {
  constexpr int NPR = (NCH + 1) / 2;                        // gru: passes over the [r; z] rows
  constexpr int NPASS = (CELL == 2) ? NPR + 1 : NCH;        // products (accumulator fills) per step
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* const At = (float*)smem;                                  // [64][PITCH]: x_t | h_{t-1} (pad columns zero)
  const int tid = threadIdx.x, lane = tid & 63, arow = lane & 15, ag = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KX = a.KX, KH = a.KH, PITCH = a.PITCH, R = a.R, H = a.H, Din = a.Din, T = a.T;
  const int ngx = KX >> 4, ngh = KH >> 4;
  char* const ring = smem + (size_t)ROWS * PITCH * 4 + (size_t)w * R * GB;   // this wave's R group buffers
  const unsigned ring_lds = lds_off(ring);
  // (tile range of this workgroup: wave-uniform 32-bit scalars -- the host keeps tiles < 2^31)
  const int t_beg = __builtin_amdgcn_readfirstlane((int)(a.tiles * (int64_t)blockIdx.x / (int64_t)gridDim.x));
  const int t_end = __builtin_amdgcn_readfirstlane((int)(a.tiles * ((int64_t)blockIdx.x + 1) / (int64_t)gridDim.x));
  if (t_beg >= t_end) return;
  for (int i = tid; i < ROWS * PITCH; i += NTHR) At[i] = 0.f;
  // The cell's bias quads, staged once per launch: Bs[plane][256] (FastLSTM: gates i, g, f, o; rnn: b_i2h + b_h2h; gru: r, z, candidate), zero past H.  (Read from
  // global memory inside the cell, every chunk's cell began with an exposed round trip: hipcc's vmcnt(0) in front of the first use.)
  float* const Bs = (float*)(smem + (size_t)ROWS * PITCH * 4 + (size_t)4 * R * GB);
  for (int i = tid; i < BS_FLOATS; i += NTHR) {
    const int q = i >> 8, u = i & 255;
    float v = 0.f;
    if (u < H) {
      if (CELL == 0) v = a.bi[q * H + u];
      else if (CELL == 1) v = (q == 0) ? a.bi[u] + a.bo[u] : 0.f;
      else v = (q < 2) ? a.bi[q * H + u] : (q == 2 ? a.bc[u] : 0.f);
    }
    Bs[i] = v;
  }
  bar();

  // ---- weight rows of this lane's four n-tiles
  auto wrow = [&](int q, int c) -> int {
    int u;
    if (CELL == 0) u = 64 * c + 16 * w + arow;
    else if (CELL == 1) u = 256 * c + 64 * w + 16 * q + arow;
    else if (c < NPR) u = 64 * std::min(2 * c + (q >> 1), NCH - 1) + 16 * w + arow;   // gru: n-tiles r, z of chunk 2 c, r, z of chunk 2 c + 1
    else u = 64 * q + 16 * w + arow;                                                   // gru: candidate rows of chunk q
    if (u >= H) u = H - 1;   // (units past H: any valid row, the result is dropped)
    if (CELL == 0) return q * H + u;
    if (CELL == 2 && c < NPR) return (q & 1) * H + u;
    return u;
  };
  // ---- the DMA stream: groups in the order they are consumed.  Cursor of the NEXT group to request: (tile, step, chunk, segment, group).
  int l_tile = t_beg, l_t = 0, l_c = 0, l_seg = 0, l_g = 0;
  int l_n = 0, l_slot = 0;    // groups requested so far; ring slot of the next request
  const float* wp[4];         // this lane's source of the next group, per n-tile
  auto l_setup = [&]() {
    const float* base = l_seg ? a.Wo : a.Wi;
    if (CELL == 2 && l_c == NPR) base = l_seg ? a.Uc : a.Wc;
    const int ld = l_seg ? H : Din;
#pragma unroll
    for (int q = 0; q < 4; ++q) wp[q] = base + (int64_t)wrow(q, l_c) * ld + 4 * ag;
  };
  l_setup();
  auto l_issue = [&]() {   // request one group (if any is left) into slot l_n % R
    if (l_tile >= t_end) return;
    const unsigned dst = ring_lds + (unsigned)l_slot * GB;
    l_slot = (l_slot + 1 == R) ? 0 : l_slot + 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      dma16(wp[q], (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + q * 1024)));
      wp[q] += 16;
    }
    ++l_n;
    if (++l_g == (l_seg ? ngh : ngx)) {
      l_g = 0;
      if (l_seg == 0 && l_t > 0) l_seg = 1;
      else {
        l_seg = 0;
        if (++l_c == NPASS) { l_c = 0; if (++l_t == T) { l_t = 0; ++l_tile; } }
      }
      l_setup();
    }
  };
  for (int i = 0; i < R; ++i) l_issue();
  int c_n = 0, c_slot = 0;   // groups consumed so far; ring slot of group c_n

  // ---- fragment sets (double buffered: group n + 1 is read under the MFMAs of group n)
  f32x4 fa[2][4], fb[2][4];
  unsigned a_lane[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_lane[i] = (unsigned)(((16 * i + arow) * PITCH + 4 * ag) * 4);
  const unsigned b_lane = (unsigned)lane * 16u;
  auto read_a = [&](int set, int kbase) {   // (PITCH is a multiple of 4 floats: every piece is 16-byte aligned -- one ds_read_b128 each)
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[set][i] = *(const f32x4*)__builtin_assume_aligned(smem + a_lane[i] + (unsigned)kbase * 4u, 16);
  };
  auto slot_of = [&](int ahead) -> int { int sl = c_slot + ahead; return sl >= R ? sl - R : sl; };   // ring slot of group c_n + ahead (ahead <= 2 <= R)
  auto read_b = [&](int set, int slot) {
    const char* src = ring + (size_t)slot * GB + b_lane;
#pragma unroll
    for (int q = 0; q < 4; ++q) fb[set][q] = *(const f32x4*)__builtin_assume_aligned(src + q * 1024, 16);
  };
  // The weights of group n have landed: loads retire in issue order, so it is enough that no more loads are outstanding than were ISSUED behind it --
  // the younger groups (4 DMA each).  (The step's x rows and MaskZero flags, requested behind the groups in flight at the head of a step, make the first
  // waits of a step stricter than they need be; allowing for them -- a run-time count, i.e. a 32-way switch around s_waitcnt's immediate in front of
  // every wait -- measured 8 % SLOWER on both shapes (profiles/r05/bench_i_*): the fixed three-way form stays.)
  auto wait_w = [&](int n) {
    const int younger = l_n - 1 - n;
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  f32x4 acc[4][4];   // [i: 16-row m-tile][q: n-tile]
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // a quarter of a group's 64 MFMAs (k-slot jj of every fragment): 512 matrix cycles, the shadow one piece of housekeeping issues in.
  // sched_barrier behind it: the MFMA builtins are plain register operations to hipcc -- left alone it bunches them away from the loads they are meant to cover.
  auto mfma16 = [&](int set, int jj) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[set][q][jj], fa[set][i][jj], acc[i][q], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- x rows: thread (row = tid >> 2, lane4 = tid & 3) owns the 16-byte pieces (lane4 + 4 j) of its row
  const int xr = tid >> 2, xl = tid & 3;
  const int npx = (Din + 15) >> 4;   // pieces per thread (the last one may lie past Din)
  f32x4 xp[XPMAX];
  auto x_request = [&](int64_t row0, int t) {
    int64_t r = row0 + xr;
    if (r >= a.N) r = a.N - 1;
    const float* src = a.in + ((int64_t)t * a.N + r) * Din + 4 * xl;
#pragma unroll
    for (int j = 0; j < XPMAX; ++j)
      if (j < npx) xp[j] = *(const f32x4u*)(src + 16 * ((4 * xl + 16 * j + 3 < Din) ? j : 0));   // (a piece past Din re-reads piece 0; it is not stored)
  };
  auto x_store = [&]() {
#pragma unroll
    for (int j = 0; j < XPMAX; ++j)
      if (j < npx && 4 * xl + 16 * j + 3 < Din) *(f32x4*)(At + xr * PITCH + 4 * xl + 16 * j) = xp[j];
  };

  for (int tile = t_beg; tile < t_end; ++tile) {
    const int64_t row0 = (int64_t)tile * ROWS;
    // (wave-uniform) every row of the tile exists: with a whole 16-unit block below H as well, a quad's stores need no lane predicate and no partial form -- the
    // exec-mask bookkeeping of `if (row < N && nv > 0) store4(.., nv)` around every store was a tenth of the cell's instructions
    const bool tfull = row0 + ROWS <= a.N;
    f32x4 cst[CELL == 0 ? NCH : 1][4];   // FastLSTM: c_t of this lane's quads
    f32x4 hn[CELL == 1 ? 4 : NCH][4];    // h_t of the step (written to LDS behind the step's last MFMA)
    f32x4 zst[CELL == 2 ? NCH : 1][4], rh[CELL == 2 ? NCH : 1][4];   // gru: z_t and r_t * h_{t-1} between the two halves of the cell
    if constexpr (CELL == 2) {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) hn[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < (CELL == 0 ? NCH : 1); ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) cst[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    x_request(row0, 0);
    x_store();
    bar();
    for (int t = 0; t < T; ++t) {
      if (CELL != 2 && t + 1 < T) x_request(row0, t + 1);   // (gru: behind the cell's first half, when r * h' has left its registers)
      float mk[4];
      if (CELL == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t r = row0 + 16 * i + arow;
          mk[i] = a.mask[(int64_t)t * a.N + (r < a.N ? r : a.N - 1)];
        }
      }
      const int nseg = t > 0 ? 2 : 1;
      // prologue of the step: the first group's fragments
      wait_w(c_n);
      read_a(0, 0);
      read_b(0, c_slot);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NPASS; ++c) {
        if constexpr (CELL == 2) {
          if (c == NPR) {
            if (t > 0) {
              bar();   // every wave has read h_{t-1} for the last time: r * h' takes its place (the candidate's recurrent operand)
#pragma unroll
              for (int ch = 0; ch < NCH; ++ch) {
                const int u0 = 64 * ch + 16 * w + 4 * ag;
                if (u0 < KH) {
#pragma unroll
                  for (int i = 0; i < 4; ++i) *(f32x4*)(At + (16 * i + arow) * PITCH + KX + u0) = rh[ch][i];
                }
              }
              bar();
            }
            if (t + 1 < T) x_request(row0, t + 1);
          }
        }
        zero_acc();
        for (int seg = 0; seg < nseg; ++seg) {
          const int ng = seg ? ngh : ngx, k0 = seg ? KX : 0;
          for (int g = 0; g < ng; g += 2) {
            // group n = c_n (set 0), then n + 1 (set 1); the group behind the pair: the next pair of this segment, the other segment, the next chunk --
            // or nothing (the step's last pair: the LDS tile is about to be rewritten)
            const bool last_pair = (g + 2 == ng) && (seg + 1 == nseg);
            const bool step_end = last_pair && (c + 1 == NPASS);
            // Each half: the 64 MFMAs of one group in four quarters; behind the first the DMA request of the group R ahead (into the slot this
            // group's fragments were read from: those reads fed the MFMAs just issued), behind the second the wait for the next group's weights
            // and its B fragments, behind the third its A fragments.
            __builtin_amdgcn_sched_barrier(0);
            mfma16(0, 0);
            l_issue();                                   // group n + R
            __builtin_amdgcn_sched_barrier(0);
            mfma16(0, 1);
            wait_w(c_n + 1);
            read_b(1, slot_of(1));
            __builtin_amdgcn_sched_barrier(0);
            mfma16(0, 2);
            read_a(1, k0 + 16 * (g + 1));
            __builtin_amdgcn_sched_barrier(0);
            mfma16(0, 3);
            mfma16(1, 0);
            l_issue();                                   // group n + 1 + R
            __builtin_amdgcn_sched_barrier(0);
            mfma16(1, 1);
            if (!step_end) {
              wait_w(c_n + 2);
              read_b(0, slot_of(2));
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma16(1, 2);
            if (!step_end) {
              int nk;                                    // k of the group behind this pair inside the LDS tile
              if (g + 2 < ng) nk = k0 + 16 * (g + 2);
              else if (seg + 1 < nseg) nk = KX;
              else nk = 0;
              read_a(0, nk);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma16(1, 3);
            c_n += 2;
            c_slot = slot_of(2);
          }
        }
        // ---- the cell of chunk c on the accumulators (lane-local)
        if constexpr (CELL == 0) {
          const int u0 = 64 * c + 16 * w + 4 * ag, nv = H - u0;
          f32x4 bq[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) bq[q] = *(const f32x4*)(Bs + q * 256 + u0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + 16 * i + arow;
            f32x4 ig, gg, fg, og, cc, hh;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              ig[r] = sigm(acc[i][0][r] + bq[0][r]);
              gg[r] = tanh_fast(acc[i][1][r] + bq[1][r]);
              fg[r] = sigm(acc[i][2][r] + bq[2][r]);
              og[r] = sigm(acc[i][3][r] + bq[3][r]);
              cc[r] = fg[r] * cst[c][i][r] + ig[r] * gg[r];
              hh[r] = og[r] * tanh_fast(cc[r]);
              if (r >= nv) { cc[r] = 0.f; hh[r] = 0.f; }   // units past H: zero columns of the next step's operand
            }
            cst[c][i] = cc;
            hn[c][i] = hh;
            auto put = [&](auto nvc) __attribute__((always_inline)) {
              const int64_t o = ((int64_t)t * a.N + row) * H + u0;
              if (SAVE) {
                store4(a.cs + o, cc, nvc);
                float* gdst = a.act + ((int64_t)t * a.N + row) * (4 * (int64_t)H) + u0;
                store4(gdst, ig, nvc); store4(gdst + H, gg, nvc); store4(gdst + 2 * H, fg, nvc); store4(gdst + 3 * H, og, nvc);
              }
              if (a.write_all_h || t == T - 1) store4(a.hs + o, hh, nvc);
            };
            if (tfull && 64 * c + 16 * w + 16 <= H) put(std::integral_constant<int, 4>{});
            else if (row < a.N && nv > 0) put(nv);
          }
        } else if constexpr (CELL == 2) {
          if (c < NPR) {   // first half: r, z of chunks 2 c, 2 c + 1;  r * h' (h' = this lane's h_{t-1}: zero at t = 0)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
              const int ch = 2 * c + cc;
              if (ch < NCH) {
                const int u0 = 64 * ch + 16 * w + 4 * ag, nv = H - u0;
                const f32x4 br = *(const f32x4*)(Bs + u0), bz = *(const f32x4*)(Bs + 256 + u0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                  const int64_t row = row0 + 16 * i + arow;
                  f32x4 rg, zg, rhv;
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    rg[r] = sigm(acc[i][2 * cc][r] + br[r]);
                    zg[r] = sigm(acc[i][2 * cc + 1][r] + bz[r]);
                    rhv[r] = (r < nv) ? rg[r] * hn[ch][i][r] : 0.f;   // units past H: zero columns of the candidate's operand
                  }
                  zst[ch][i] = zg;
                  rh[ch][i] = rhv;
                  if (SAVE) {
                    auto put = [&](auto nvc) __attribute__((always_inline)) {
                      float* gdst = a.act + ((int64_t)t * a.N + row) * (4 * (int64_t)H) + u0;
                      store4(gdst, rg, nvc); store4(gdst + H, zg, nvc); store4(gdst + 3 * H, rhv, nvc);
                    };
                    if (tfull && 64 * ch + 16 * w + 16 <= H) put(std::integral_constant<int, 4>{});
                    else if (row < a.N && nv > 0) put(nv);
                  }
                }
              }
            }
          } else {   // second half: n = tanh(.), h = (1 - z) n + z h'
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
              const int u0 = 64 * ch + 16 * w + 4 * ag, nv = H - u0;
              const f32x4 bn = *(const f32x4*)(Bs + 512 + u0);
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int64_t row = row0 + 16 * i + arow;
                f32x4 ng, hh;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  ng[r] = tanh_fast(acc[i][ch][r] + bn[r]);
                  const float z = zst[ch][i][r];
                  hh[r] = (r < nv) ? (1.f - z) * ng[r] + z * hn[ch][i][r] : 0.f;
                }
                hn[ch][i] = hh;
                auto put = [&](auto nvc) __attribute__((always_inline)) {
                  const int64_t o = ((int64_t)t * a.N + row) * H + u0;
                  if (SAVE) store4(a.act + ((int64_t)t * a.N + row) * (4 * (int64_t)H) + 2 * H + u0, ng, nvc);
                  if (a.write_all_h || t == T - 1) store4(a.hs + o, hh, nvc);
                };
                if (tfull && 64 * ch + 16 * w + 16 <= H) put(std::integral_constant<int, 4>{});
                else if (row < a.N && nv > 0) put(nv);
              }
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int u0 = 256 * c + 64 * w + 16 * q + 4 * ag, nv = H - u0;
            const f32x4 bv = *(const f32x4*)(Bs + u0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int64_t row = row0 + 16 * i + arow;
              const f32x4 pre = acc[i][q] + bv;
              const bool live = mk[i] != 0.f;
              f32x4 v;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float x = a.relu ? fmaxf(pre[r], 0.f) : tanh_fast(pre[r]);
                v[r] = (live && r < nv) ? x : 0.f;
              }
              hn[q][i] = v;
              auto put = [&](auto nvc) __attribute__((always_inline)) {
                const int64_t o = ((int64_t)t * a.N + row) * H + u0;
                if (SAVE) store4(a.act + o, pre, nvc);
                if (a.write_all_h || t == T - 1) store4(a.hs + o, v, nvc);
              };
              if (tfull && 256 * c + 64 * w + 16 * q + 16 <= H) put(std::integral_constant<int, 4>{});
              else if (row < a.N && nv > 0) put(nv);
            }
          }
        }
      }
      // ---- every wave has read [x_t | h_{t-1}] for the last time: h_t and x_{t+1} take their place
      bar();
      if (t + 1 < T) {
        x_store();
        if constexpr (CELL != 1) {
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const int u0 = 64 * c + 16 * w + 4 * ag;
            if (u0 < KH) {
#pragma unroll
              for (int i = 0; i < 4; ++i) *(f32x4*)(At + (16 * i + arow) * PITCH + KX + u0) = hn[c][i];
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int u0 = 64 * w + 16 * q + 4 * ag;
            if (u0 < KH) {
#pragma unroll
              for (int i = 0; i < 4; ++i) *(f32x4*)(At + (16 * i + arow) * PITCH + KX + u0) = hn[q][i];
            }
          }
        }
      }
      bar();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// =====================================================================================================================================================
// BPTT through one layer as ONE persistent launch: the cell backward of all T steps + the recurrent gradient dh_{t-1} = dA_t W_o2g, replacing per step one
// element-wise launch (kk::lstm_gates_bwd / rnn_cell_bwd: saves in, dH / dC in and out, dA out) and one dh product launch (kprn_api.hip backward_generic).
// dW and dx stay products over the dA this launch writes ([T][N][GH], the generic layout).
//   * A workgroup owns a 64-path tile for t = T-1 .. 0; dc_t and dh live in registers, in the SAME lane layout as the forward's cell (lane (arow, ag) of wave w:
//     path 16 i + arow, hidden units 64 c + 16 w + 4 ag .. + 3 -- rnn: 64 w + 16 q + 4 ag), because the product is taken transposed with the weight
//     fragment (rows of W_o2g^T) as the first MFMA operand and n-tile c of wave w = output units 64 c + 16 w ..: dh_{t-1} lands where the cell backward reads it.
//   * K (the gate columns) is walked in chunks of 256 = 64 hidden units x 4 gates (rnn: 256 units): the cell backward of a chunk writes its dA quads to
//     global memory and into an LDS tile [64][256 + 4] (k = gate * 64 + unit of the chunk), barrier, 16 groups of 16 k of product on it, barrier.
//   * W_o2g^T fragments stream L2 -> LDS by DMA three groups ahead (as in the forward; `global_load_lds_dwordx4 voffset, sbase`: the per-group address
//     arithmetic is scalar); the saves of the NEXT chunk are requested right behind a chunk's cell backward, so they land under its product.  Those
//     requests sit between DMA groups in the in-order return stream: the hand-written waits for the first three groups of a product allow for them
//     (vmcnt(groups + NPF)) instead of draining them.
// measurement builds (scripts/build_variants.py, KPRN_VARIANT_DEFS=-DKPRN_BPTT_DBG=<mask>): 1 no dA stores to global memory, 2 no save requests, 4 no products, (rnn only) 8 no LDS tile writes, 16 no quad shift / masks.  Results are
// wrong by construction; the shipped library has the mask at 0.
#ifndef KPRN_BPTT_DBG
#define KPRN_BPTT_DBG 0
#endif
__device__ __forceinline__ void store4d(float* __restrict__ p, const f32x4 v, int nv) {
  if (KPRN_BPTT_DBG & 1) { asm volatile("" :: "v"(v), "v"(p)); return; }
  store4(p, v, nv);
}
struct BPArgs {
  const float* act;      // FastLSTM: gate values [T][N][4H];  rnn: unused
  const float* cs;       // FastLSTM: c [T][N][H]
  const float* hs;       // rnn: h [T][N][H] (the activation's derivative is formed from it)
  const float* mask;     // rnn: [T][N]
  const float* dHup;     // UP: [T][N][H] gradient from the layer above;  else [N][H]: the head's gradient, applied at t = T-1
  const float* WoT;      // [Hp rows][WG] W_o2g^T (row = output unit; zero rows / slack behind H);  gru: [c_h2h^T | o2g^T] (WG = 3H: k = candidate unit, r row, z row)
  float* dA;             // [T][N][GH]
  float* dbias; float* dbias2;   // FastLSTM (up to 2 chunks) / rnn: the bias gradient (column sums of dA) is formed HERE, += per workgroup (nullptr: not wanted); rnn: both biases
  int64_t N; int T, H, GH, WG, relu;
  int64_t tiles;
};

constexpr int BP_LD = 260;      // floats per row of the dA chunk tile
constexpr int BP_R = 3;         // DMA ring depth (groups)

template <int N_> __device__ __forceinline__ void vmwait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000); }
// DMA with a scalar base: address = sbase + voff (32-bit, unsigned)
__device__ __forceinline__ void dma16s(const float* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// CELL 0: FastLSTM, NCH chunks (of 64 units) = n-tiles per wave;  CELL 1: rnn, one chunk of 256 units, 4 n-tiles per wave.  UP: a layer above exists.
// CELL 2: nn.GRU (OneModel.lua:237-238) in the rnn's lane layout (n-tile q of wave w = units 64 w + 16 q ..).  A step's backward has two DEPENDENT products:
//   d pre_n = dh (1 - z)(1 - n^2) -> d(r h') = d pre_n c_h2h -> d pre_r = d(r h') h' r (1 - r) -> dh' = dh z + d(r h') r + [d pre_r | d pre_z] o2g,
// walked as THREE K chunks over one [c_h2h^T | o2g^T] stream: chunk 0 (k = candidate units) forms d(r h') in the accumulators, the cell's second part
// turns it into d pre_r and restarts them, chunks 1 (r rows) and 2 (z rows) accumulate the rest of dh'.  The saves a part needs are requested one product
// ahead of it where the registers allow (r under chunk 0; the gradient from above of step t - 1 under chunk 1, added to dh behind it); z, n, h' are fetched at
// the head of the step.
template <int CELL, int NCH, bool UP>
__global__ __launch_bounds__(NTHR, 1) void k_bptt(BPArgs a) {
  constexpr int NT = (CELL == 0) ? NCH : 4;          // n-tiles (groups of 16 output units) per wave
  constexpr int KCH = (CELL == 0) ? NCH : (CELL == 2 ? 3 : 1);   // K chunks per step
  constexpr int NPF = (CELL == 0) ? 28 : 32;         // 16-byte save requests per lane and chunk (gru: 16 / 16 or none / none behind the cell's three parts)
  constexpr int GBB = NT * 1024;                     // bytes of one wave's weight group
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* const Dt = (float*)smem;                    // dA chunk tile [64][BP_LD]
  const int tid = threadIdx.x, lane = tid & 63, arow = lane & 15, ag = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, GH = a.GH, T = a.T;
  const int WG = a.WG;                                            // floats per row of the transposed weights
  const int NG = (CELL == 2) ? (((H + 31) >> 5) << 1) : 16;       // groups of 16 k per chunk (gru: the chunk is H wide, in whole pairs of groups)
  char* const ring = smem + (size_t)ROWS * BP_LD * 4 + (size_t)w * BP_R * GBB;
  const unsigned ring_lds = lds_off(ring);
  const int t_beg = __builtin_amdgcn_readfirstlane((int)(a.tiles * (int64_t)blockIdx.x / (int64_t)gridDim.x));
  const int t_end = __builtin_amdgcn_readfirstlane((int)(a.tiles * ((int64_t)blockIdx.x + 1) / (int64_t)gridDim.x));
  if (t_beg >= t_end) return;

  // ---- weight stream.  Lane's row of n-tile cc: output unit uo(cc) = 64 cc + 16 w + arow (rnn: 64 w + 16 cc + arow); its byte offset inside W_o2g^T + 16 ag
  unsigned voff[NT];
#pragma unroll
  for (int cc = 0; cc < NT; ++cc) {
    int uo = (CELL == 0) ? 64 * cc + 16 * w + arow : 64 * w + 16 * cc + arow;
    if (uo >= H) uo = H - 1;
    voff[cc] = (unsigned)(((int64_t)uo * WG + 4 * ag) * 4);
  }
  // cursor of the next group to request: (tile, step, chunk, group); steps T-1 .. 1 have a product, step 0 has none
  int l_tile = (T > 1) ? t_beg : t_end, l_t = T - 1, l_c = 0, l_g = 0, l_n = 0, l_slot = 0;
  auto l_issue = [&]() {
    if (l_tile >= t_end) return;
    // k of the group inside the row: FastLSTM gate (g >> 2), units 64 c + 16 (g & 3) ..;  rnn: units 16 g ..
    const int koff = (CELL == 0) ? (l_g >> 2) * H + 64 * l_c + 16 * (l_g & 3) : (CELL == 2 ? __builtin_amdgcn_readfirstlane(l_c * H + 16 * l_g) : 16 * l_g);
    const float* sb = a.WoT + koff;
    const unsigned dst = ring_lds + (unsigned)l_slot * GBB;
    l_slot = (l_slot + 1 == BP_R) ? 0 : l_slot + 1;
#pragma unroll
    for (int cc = 0; cc < NT; ++cc) dma16s(sb, voff[cc], (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + cc * 1024)));
    ++l_n;
    if (++l_g == NG) {
      l_g = 0;
      if (++l_c == KCH) { l_c = 0; if (--l_t == 0) { l_t = T - 1; ++l_tile; } }
    }
  };
  for (int i = 0; i < BP_R; ++i) l_issue();
  int c_n = 0, c_slot = 0;
  auto slot_of = [&](int ahead) -> int { int sl = c_slot + ahead; return sl >= BP_R ? sl - BP_R : sl; };
  // group n has landed; pf: the NPF save requests of this chunk were issued behind it (it is one of the first BP_R groups of the chunk's product)
  auto wait_w = [&](int n, int pf /* save requests issued behind the group: 0, NPF (gru: 0 or 16) */) {
    const int younger = l_n - 1 - n;
    if (CELL == 2 && pf == 16) {
      if (younger >= 2) vmwait<2 * NT + 16>();
      else if (younger == 1) vmwait<NT + 16>();
      else vmwait<16>();
    } else if (pf) {
      if (younger >= 2) vmwait<2 * NT + NPF>();
      else if (younger == 1) vmwait<NT + NPF>();
      else vmwait<NPF>();
    } else {
      if (younger >= 2) vmwait<2 * NT>();
      else if (younger == 1) vmwait<NT>();
      else vmwait<0>();
    }
  };

  f32x4 fa[2][4], fb[2][NT];
  unsigned a_lane[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_lane[i] = (unsigned)(((16 * i + arow) * BP_LD + 4 * ag) * 4);
  const unsigned b_lane = (unsigned)lane * 16u;
  auto read_a = [&](int set, int g) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[set][i] = *(const f32x4*)__builtin_assume_aligned(smem + a_lane[i] + (unsigned)g * 64u, 16);
  };
  auto read_b = [&](int set, int slot) {
    const char* src = ring + (size_t)slot * GBB + b_lane;
#pragma unroll
    for (int cc = 0; cc < NT; ++cc) fb[set][cc] = *(const f32x4*)__builtin_assume_aligned(src + cc * 1024, 16);
  };
  f32x4 acc[4][NT];   // dh_{t-1} forming: [i][n-tile]
  auto mfma_q = [&](int set, int jj) {
#pragma unroll
    for (int cc = 0; cc < NT; ++cc)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[set][cc][jj], fa[set][i][jj], acc[i][cc], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- saves of one chunk's cell backward (requested a chunk ahead)
  struct Sv { f32x4 g[CELL == 0 ? 4 : 1][4]; f32x4 c[4], cp[4]; f32x4 up[CELL == 0 ? 1 : 4][4]; f32x4 hq[CELL == 0 ? 1 : 4][4]; };
  Sv sv;
  // gru: the record's r, z, n planes (act [T][N][4H] = [r | z | n | r h']), h' = h_{t-1} (sv.hq), the gradient from above (sv.up); which: 1 r, 2 z + n, 4 up, 8 h'
  f32x4 gr[CELL == 2 ? 4 : 1][4], gz[CELL == 2 ? 4 : 1][4], gn[CELL == 2 ? 4 : 1][4];
  // Buffer loads: descriptor + plane offset in scalar registers, ONE 32-bit lane offset per (n-tile, m-tile).  (With ordinary pointers the 64 requests in front of a
  // tile's first step hold 64 address pairs beside their 64 quads: 1.0-1.3 KB of scratch per lane.)
  unsigned rl4[4], u4[4];   // this lane's rows of the tile (clamped to the batch) x 4 bytes; first unit of its quads (a quad straddling H re-reads the last full one) x 4 bytes
  auto tile_lanes = [&](int64_t row0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t rl = 16 * i + arow;
      if (row0 + rl >= a.N) rl = a.N - 1 - row0;
      rl4[i] = (unsigned)rl * 4u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int u = 64 * w + 16 * q + 4 * ag;
      if (u >= H) u = 0;
      u4[q] = (unsigned)u * 4u;
    }
  };
  auto ldq = [&](rsrc_t r, unsigned voff, unsigned soff) -> f32x4 { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0)); };
  auto request_gru = [&](int64_t row0, int t, int which) {
    if (KPRN_BPTT_DBG & 2) return;
    asm volatile("" ::: "memory");
    const rsrc_t ra = make_rsrc(a.act + ((int64_t)t * a.N + row0) * GH);
    const rsrc_t rh = make_rsrc(a.hs + ((int64_t)(t > 0 ? t - 1 : 0) * a.N + row0) * H);   // (t = 0: any valid rows, the cell takes h' = 0)
    const rsrc_t ru = make_rsrc(a.dHup + (UP ? ((int64_t)t * a.N + row0) * H : row0 * H));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned va = rl4[i] * (unsigned)GH + u4[q], vh = rl4[i] * (unsigned)H + u4[q];
        if (which & 1) gr[q][i] = ldq(ra, va, 0u);
        if (which & 2) { gz[q][i] = ldq(ra, va, (unsigned)H * 4u); gn[q][i] = ldq(ra, va, (unsigned)H * 8u); }
        if (which & 8) sv.hq[q][i] = ldq(rh, vh, 0u);
        if (which & 4) sv.up[q][i] = ldq(ru, vh, 0u);
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  f32x4 dzk[CELL == 2 ? 4 : 1][4];   // gru: d pre_z between the cell's first and third part
  auto request = [&](int64_t row0, int t, int c) {   // unconditional loads from clamped addresses: exactly NPF requests per lane
    if (KPRN_BPTT_DBG & 2) return;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t row = row0 + 16 * i + arow;
      if (row >= a.N) row = a.N - 1;
      if constexpr (CELL == 0) {
        int u = 64 * c + 16 * w + 4 * ag;
        if (u >= H) u = 0;   // (units past H: any valid quad, masked in the cell.  A quad STRADDLING H is read where it lies -- up to 12 bytes past its row, i.e. the next row's first values or, behind a plane's last row, the 64 bytes of slack every device allocation carries (kprn_api.hip dalloc) -- and its lanes past H are masked by SELECTS: shifting it back inside the row cost every element of every plane a runtime component select, 14 % of the shipped rnn's launch: profiles/r06/probe_bptt_knockouts.txt)
        const float* ar = a.act + ((int64_t)t * a.N + row) * GH + u;
#pragma unroll
        for (int q = 0; q < 4; ++q) sv.g[q][i] = *(const f32x4u*)(ar + q * H);
        sv.c[i] = *(const f32x4u*)(a.cs + ((int64_t)t * a.N + row) * H + u);
        sv.cp[i] = *(const f32x4u*)(a.cs + ((int64_t)(t > 0 ? t - 1 : 0) * a.N + row) * H + u);
        sv.up[0][i] = *(const f32x4u*)(a.dHup + (UP ? ((int64_t)t * a.N + row) * H : row * H) + u);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          int u = 64 * w + 16 * q + 4 * ag;
          if (u >= H) u = 0;
          sv.hq[q][i] = *(const f32x4u*)(a.hs + ((int64_t)t * a.N + row) * H + u);
          sv.up[q][i] = *(const f32x4u*)(a.dHup + (UP ? ((int64_t)t * a.N + row) * H : row * H) + u);
        }
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // column sums of this workgroup's dA quads (the bias gradient): per lane, reduced over the 16 lanes of a unit quad and added to the gradient once, at the end --
  // the separate sweep over dA (kernels_basic.hip k_colsum) re-read 1.2 GB per layer at dims B
  constexpr bool BSUM = (CELL == 1) || (CELL == 0 && NCH <= 2);   // (three chunks: 48 more registers spilled 500 bytes per lane)
  // FastLSTM from three chunks up: the same sums, kept in LDS.  A part's quads are summed over its four m-tiles in registers and over the 16 lanes of a unit quad by
  // shuffles; lane arow = 0 then adds them to ITS slots of Bs[gate][256] -- every slot has one owner for the whole launch: no atomics, no barrier -- and flushes them at the end.
  constexpr bool BSUM_LDS = (CELL == 0 && NCH >= 3);
  constexpr bool BSUM_GRU = (CELL == 2);   // gru: slots Bs[0 r | 1 z | 2 n][256] (d pre_r, d pre_z -> i2g.bias; d pre_n -> c_i2h.bias = dbias2)
  float* const Bs = (float*)(smem + (size_t)ROWS * BP_LD * 4 + (size_t)4 * BP_R * GBB);
  if (BSUM_LDS && a.dbias && arow == 0) {
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int g = 0; g < 4; ++g) *(f32x4*)(Bs + g * 256 + 64 * c + 16 * w + 4 * ag) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (BSUM_GRU && a.dbias && arow == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int g = 0; g < 3; ++g) *(f32x4*)(Bs + g * 256 + 64 * w + 16 * q + 4 * ag) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 bsum[BSUM ? (CELL == 0 ? 4 * NCH : 4) : 1];
#pragma unroll
  for (int j = 0; j < (BSUM ? (CELL == 0 ? 4 * NCH : 4) : 1); ++j) bsum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int tile = t_beg; tile < t_end; ++tile) {
    const int64_t row0 = (int64_t)tile * ROWS;
    const bool tfull = row0 + ROWS <= a.N;   // (wave-uniform) every row of the tile exists: see the forward
    f32x4 dh[4][NT], dc[CELL == 0 ? NCH : 1][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int cc = 0; cc < NT; ++cc) { dh[i][cc] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[i][cc] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int c = 0; c < (CELL == 0 ? NCH : 1); ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) dc[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (CELL == 2) {
      // the gradient that enters at the last step becomes dh right away (the cell's first part never looks at sv.up: one plane less alive across the products)
      tile_lanes(row0);
      request_gru(row0, T - 1, 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int u0 = 64 * w + 16 * q + 4 * ag, nv = H - u0;
        constexpr int sh = 0;   // (quads are read where they lie: see request)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int rs = (r + sh < 4) ? r + sh : 3;
            dh[i][q][r] = (r < nv && row0 + 16 * i + arow < a.N) ? sv.up[q][i][rs] : 0.f;
          }
      }
    } else request(row0, T - 1, 0);
    for (int t = T - 1; t >= 0; --t) {
      const float upw = (UP || t == T - 1) ? 1.f : 0.f;   // the head's gradient enters at the last step only
      float mk[4];
      if (CELL == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t r = row0 + 16 * i + arow;
          mk[i] = (r < a.N) ? a.mask[(int64_t)t * a.N + r] : 0.f;
        }
      }
      // ---- gru: one part of the cell backward (c = 0, 1, 2: see the kernel's header), dA quads -> global + LDS tile
      auto gru_part = [&](auto c_) __attribute__((always_inline)) {
        constexpr int c = decltype(c_)::value;
        // (the other cells' take-over of a part's planes in front of its first dA store measured 5 % SLOWER here -- 1.656 -> 1.736 ms, one call, alternating: not applied)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int u0 = 64 * w + 16 * q + 4 * ag, nv = H - u0;
              constexpr int sh = 0;   // (quads are read where they lie: see request)
              f32x4 ps = f32x4{0.f, 0.f, 0.f, 0.f}, psz = f32x4{0.f, 0.f, 0.f, 0.f};   // the quad's column sums over its m-tiles (step 0: d pre_z leaves with the first part)
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int64_t row = row0 + 16 * i + arow;
                f32x4 d;
                int col;   // column block of the record this part's quad belongs to
                if (c == 0) {          // d pre_n -> tile; d pre_z kept; dh <- the direct path dh z
                  col = 2 * H;
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const int rs = (r + sh < 4) ? r + sh : 3;
                    const float z = gz[q][i][rs], nn = gn[q][i][rs];
                    const float hp = t > 0 ? sv.hq[q][i][rs] : 0.f;
                    const float dhv = dh[i][q][r];   // (the gradient from above is in it: added at the tile's start / behind chunk 1's product of the step before)
                    const bool ok = r < nv && row < a.N;
                    d[r] = ok ? dhv * (1.f - z) * (1.f - nn * nn) : 0.f;
                    dzk[q][i][r] = ok ? dhv * (hp - nn) * z * (1.f - z) : 0.f;
                    dh[i][q][r] = ok ? dhv * z : 0.f;
                  }
                } else if (c == 1) {   // the accumulators hold d(r h'): d pre_r -> tile; dh += d(r h') r; the accumulators start again
                  col = 0;
#pragma unroll
                  for (int r = 0; r < 4; ++r) {
                    const int rs = (r + sh < 4) ? r + sh : 3;
                    const float rr = gr[q][i][rs], hp = sv.hq[q][i][rs], drh = acc[i][q][r];
                    const bool ok = r < nv && row < a.N;
                    d[r] = ok ? drh * hp * rr * (1.f - rr) : 0.f;
                    dh[i][q][r] += ok ? drh * rr : 0.f;
                  }
                  acc[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
                } else {               // d pre_z -> tile; the gradient from above of step t - 1 (requested under chunk 1's product) joins dh
                  col = H;
                  d = dzk[q][i];
                  if (UP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                      const int rs = (r + sh < 4) ? r + sh : 3;
                      dh[i][q][r] += (r < nv && row < a.N) ? sv.up[q][i][rs] : 0.f;
                    }
                  }
                }
                ps += d;
                if (c == 0 && t == 0) psz += dzk[q][i];
                *(f32x4*)(Dt + (16 * i + arow) * BP_LD + 64 * w + 16 * q + 4 * ag) = d;
                if (row < a.N && nv > 0) {   // (the uniform whole-block form of the other cells measured 1 % slower here: profiles/r06/bench_y_*)
                  float* gdst = a.dA + ((int64_t)t * a.N + row) * GH + u0;
                  store4d(gdst + col, d, nv);
                  if (t == 0) { store4d(gdst + H, dzk[q][i], nv); store4d(gdst, f32x4{0.f, 0.f, 0.f, 0.f}, nv); }
                }
              }
              if (a.dbias) {   // (uniform) the bias gradient's share of this quad: over its 16 lanes, into the owner lane's LDS slots (see BSUM_LDS)
#pragma unroll
                for (int m = 1; m < 16; m <<= 1)
#pragma unroll
                  for (int r = 0; r < 4; ++r) { ps[r] += __shfl_xor(ps[r], m, 64); if (c == 0 && t == 0) psz[r] += __shfl_xor(psz[r], m, 64); }
                if (arow == 0) {
                  *(f32x4*)(Bs + (c == 0 ? 2 : (c == 1 ? 0 : 1)) * 256 + u0) += ps;
                  if (c == 0 && t == 0) *(f32x4*)(Bs + 256 + u0) += psz;
                }
              }
            }
      };
      // ---- gru: the product of one chunk on the tile the part just wrote (npf: save requests issued behind the groups in flight)
      auto gru_product = [&](int npf) __attribute__((always_inline)) {
        if (KPRN_BPTT_DBG & 4) return;
        bar();
        wait_w(c_n, npf);
        read_a(0, 0);
        read_b(0, c_slot);
        __builtin_amdgcn_sched_barrier(0);
        for (int g = 0; g < NG; g += 2) {
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(0, 0);
          l_issue();
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(0, 1);
          wait_w(c_n + 1, g + 1 < BP_R ? npf : 0);
          read_b(1, slot_of(1));
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(0, 2);
          read_a(1, g + 1);
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(0, 3);
          mfma_q(1, 0);
          l_issue();
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(1, 1);
          if (g + 2 < NG) {
            wait_w(c_n + 2, g + 2 < BP_R ? npf : 0);
            read_b(0, slot_of(2));
          }
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(1, 2);
          if (g + 2 < NG) read_a(0, g + 2);
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(1, 3);
          c_n += 2;
          c_slot = slot_of(2);
        }
        bar();   // every wave is done with the tile: the next part may overwrite it
      };
      if constexpr (CELL == 2) {
        // step 0: h' = 0 -- no d(r h'), no d pre_r; the first part writes the step's whole dA
        // Which plane is requested where is a register budget (512 per lane: dh, the accumulators, the fragment sets and d pre_z take 256) and an allocator matter: planes
        // requested under chunk 2's product are alive across the step loop's back edge, and every such schedule came out with ~1 KB of scratch per lane, reloaded inside
        // the MFMA loops.  Measured at bench.py's --dims gru batch (65 536 paths, H 250, L 1): every part fetching its own planes (three exposed round trips a step)
        // 3.32 ms; r and the gradient from above under chunks 0 / 1, z, n, h' exposed at the head of the step (this) 1.74 ms = 0.45 of the fp32 peak on the three
        // products; anything under chunk 2 (z, n, h' or z, n) 3.25 ms.  The per-step launches this replaces: 2.65 ms.
        request_gru(row0, t, 10);                          // z, n, h'
        gru_part(std::integral_constant<int, 0>{});
        if (t > 0) {
          request_gru(row0, t, 1);                         // r, under chunk 0's product
          gru_product((KPRN_BPTT_DBG & 2) ? 0 : 16);
          gru_part(std::integral_constant<int, 1>{});
          if (UP) request_gru(row0, t - 1, 4);             // the gradient from above of step t - 1, under chunk 1's product
          gru_product(((KPRN_BPTT_DBG & 2) || !UP) ? 0 : 16);
          gru_part(std::integral_constant<int, 2>{});
          gru_product(0);
        }
      }
#pragma unroll
      for (int c = 0; c < (CELL == 2 ? 0 : KCH); ++c) {
        // ---- cell backward of chunk c: dA quads -> global + LDS tile
        // Every requested plane is taken over HERE, in front of the first dA store (KPRN_BPTT_TAKEOVER: cell mask, 0 = off for A/B builds): hipcc waits for a loaded register at
        // its first use with a COUNT, and with the part's own stores in the same in-order counter the later quads' waits turned into waits for the stores' acknowledgements.
        // One call, alternating (profiles/r06/bench_x_*): shipped rnn 0.628 -> 0.618 ms, dims B 1.351 -> 1.330.
#ifndef KPRN_BPTT_TAKEOVER
#define KPRN_BPTT_TAKEOVER 3
#endif
        if constexpr (CELL != 2) {
          if ((KPRN_BPTT_TAKEOVER >> CELL) & 1) {
            if constexpr (CELL == 0) {
              asm volatile("" :: "v"(sv.g[0][0]), "v"(sv.g[0][1]), "v"(sv.g[0][2]), "v"(sv.g[0][3]), "v"(sv.g[1][0]), "v"(sv.g[1][1]), "v"(sv.g[1][2]), "v"(sv.g[1][3]));
              asm volatile("" :: "v"(sv.g[2][0]), "v"(sv.g[2][1]), "v"(sv.g[2][2]), "v"(sv.g[2][3]), "v"(sv.g[3][0]), "v"(sv.g[3][1]), "v"(sv.g[3][2]), "v"(sv.g[3][3]));
              asm volatile("" :: "v"(sv.c[0]), "v"(sv.c[1]), "v"(sv.c[2]), "v"(sv.c[3]), "v"(sv.cp[0]), "v"(sv.cp[1]), "v"(sv.cp[2]), "v"(sv.cp[3]));
              asm volatile("" :: "v"(sv.up[0][0]), "v"(sv.up[0][1]), "v"(sv.up[0][2]), "v"(sv.up[0][3]));
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q)
                asm volatile("" :: "v"(sv.hq[q][0]), "v"(sv.hq[q][1]), "v"(sv.hq[q][2]), "v"(sv.hq[q][3]), "v"(sv.up[q][0]), "v"(sv.up[q][1]), "v"(sv.up[q][2]), "v"(sv.up[q][3]));
            }
          }
        }
        if constexpr (CELL == 2) {
          // (gru: its own step body above)
        } else if constexpr (CELL == 0) {
          const int u0 = 64 * c + 16 * w + 4 * ag, nv = H - u0;
          constexpr int sh = 0;   // (quads are read where they lie: see request)
          f32x4 ps[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};   // (BSUM_LDS: this part's sums over its m-tiles)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int64_t row = row0 + 16 * i + arow;
            f32x4 di, dg, df, dO;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int rs = (r + sh < 4) ? r + sh : 3;
              const float ig = sv.g[0][i][rs], gg = sv.g[1][i][rs], fg = sv.g[2][i][rs], og = sv.g[3][i][rs];
              const float tc = tanh_fast(sv.c[i][rs]);
              const float cp = t > 0 ? sv.cp[i][rs] : 0.f;
              const float dhv = dh[i][c][r] + upw * sv.up[0][i][rs];
              const float dov = dhv * tc;
              const float dcv = dc[c][i][r] + dhv * og * (1.f - tc * tc);
              const bool ok = r < nv && row < a.N;
              di[r] = ok ? dcv * gg * ig * (1.f - ig) : 0.f;
              dg[r] = ok ? dcv * ig * (1.f - gg * gg) : 0.f;
              df[r] = ok ? dcv * cp * fg * (1.f - fg) : 0.f;
              dO[r] = ok ? dov * og * (1.f - og) : 0.f;
              dc[c][i][r] = ok ? dcv * fg : 0.f;
            }
            if constexpr (BSUM) { bsum[4 * c] += di; bsum[4 * c + 1] += dg; bsum[4 * c + 2] += df; bsum[4 * c + 3] += dO; }
            if constexpr (BSUM_LDS) { ps[0] += di; ps[1] += dg; ps[2] += df; ps[3] += dO; }
            float* lrow = Dt + (16 * i + arow) * BP_LD + 16 * w + 4 * ag;
            *(f32x4*)(lrow) = di; *(f32x4*)(lrow + 64) = dg; *(f32x4*)(lrow + 128) = df; *(f32x4*)(lrow + 192) = dO;
            auto put = [&](auto nvc) __attribute__((always_inline)) {
              float* gdst = a.dA + ((int64_t)t * a.N + row) * GH + u0;
              store4d(gdst, di, nvc); store4d(gdst + H, dg, nvc); store4d(gdst + 2 * H, df, nvc); store4d(gdst + 3 * H, dO, nvc);
            };
            if (tfull && 64 * c + 16 * w + 16 <= H) put(std::integral_constant<int, 4>{});
            else if (row < a.N && nv > 0) put(nv);
          }
          if constexpr (BSUM_LDS) {
            if (a.dbias) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int m = 1; m < 16; m <<= 1)
#pragma unroll
                  for (int r = 0; r < 4; ++r) ps[g][r] += __shfl_xor(ps[g][r], m, 64);
                if (arow == 0) *(f32x4*)(Bs + g * 256 + u0) += ps[g];
              }
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int u0 = 64 * w + 16 * q + 4 * ag, nv = H - u0;
            constexpr int sh = 0;   // (quads are read where they lie: see request)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int64_t row = row0 + 16 * i + arow;
              f32x4 d;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int rs = (KPRN_BPTT_DBG & 16) ? r : ((r + sh < 4) ? r + sh : 3);   // (16: measurement -- no quad shift, no masks)
                const float hv = sv.hq[q][i][rs];
                const float der = a.relu ? (hv > 0.f ? 1.f : 0.f) : (1.f - hv * hv);
                const float dhv = dh[i][q][r] + upw * sv.up[q][i][rs];
                d[r] = (KPRN_BPTT_DBG & 16) ? dhv * der : ((r < nv && mk[i] != 0.f) ? dhv * der : 0.f);
              }
              if constexpr (BSUM) bsum[q] += d;
              if (!(KPRN_BPTT_DBG & 8)) *(f32x4*)(Dt + (16 * i + arow) * BP_LD + 64 * w + 16 * q + 4 * ag) = d;   // (8: measurement -- no LDS tile writes)
              if (tfull && 64 * w + 16 * q + 16 <= H) store4d(a.dA + ((int64_t)t * a.N + row) * GH + u0, d, 4);
              else if (row < a.N && nv > 0) store4d(a.dA + ((int64_t)t * a.N + row) * GH + u0, d, nv);
            }
          }
        }
        // ---- the saves of the chunk behind this one (this tile's next chunk / step); the next tile requests its own first chunk
        const bool more = (CELL == 2) ? true : !(c + 1 == KCH && t == 0);
        if constexpr (CELL != 2) {
          if (more) {
            if (c + 1 < KCH) request(row0, t, c + 1); else request(row0, t - 1, 0);
          }
        }
        const int npf = (KPRN_BPTT_DBG & 2) ? 0 : ((CELL == 2) ? 0 : (more ? NPF : 0));   // requests issued behind the product's first groups
        if (t == 0) continue;   // (uniform) step 0: no dh_{-1} to form
        if (KPRN_BPTT_DBG & 4) continue;
        bar();
        // ---- product of the chunk: 16 groups in 8 pairs; fragments of group n + 1 are read under the MFMAs of group n
        wait_w(c_n, npf);
        read_a(0, 0);
        read_b(0, c_slot);
        __builtin_amdgcn_sched_barrier(0);
        for (int g = 0; g < NG; g += 2) {
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(0, 0);
          l_issue();
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(0, 1);
          wait_w(c_n + 1, g + 1 < BP_R ? npf : 0);
          read_b(1, slot_of(1));
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(0, 2);
          read_a(1, g + 1);
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(0, 3);
          mfma_q(1, 0);
          l_issue();
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(1, 1);
          if (g + 2 < NG) {
            wait_w(c_n + 2, g + 2 < BP_R ? npf : 0);
            read_b(0, slot_of(2));
          }
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(1, 2);
          if (g + 2 < NG) read_a(0, g + 2);
          __builtin_amdgcn_sched_barrier(0);
          mfma_q(1, 3);
          c_n += 2;
          c_slot = slot_of(2);
        }
        bar();   // every wave is done with the tile: the next chunk's cell backward may overwrite it
      }
      // dh_{t-1} is complete: it becomes the step's dh, the accumulators start again from zero
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int cc = 0; cc < NT; ++cc) {
          if (CELL == 2) dh[i][cc] += acc[i][cc]; else dh[i][cc] = acc[i][cc];   // (gru: on top of dh z + d(r h') r)
          acc[i][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
  }
  if constexpr (BSUM_LDS) {
    if (a.dbias && arow == 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int u0 = 64 * c + 16 * w + 4 * ag;
          const f32x4 v = *(const f32x4*)(Bs + g * 256 + u0);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (u0 + r < H) atomicAdd(a.dbias + g * H + u0 + r, v[r]);
        }
    }
  }
  if constexpr (BSUM_GRU) {
    if (a.dbias && arow == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          const int u0 = 64 * w + 16 * q + 4 * ag;
          const f32x4 v = *(const f32x4*)(Bs + g * 256 + u0);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (u0 + r < H) atomicAdd((g < 2 ? a.dbias + g * H : a.dbias2) + u0 + r, v[r]);
        }
    }
  }
  if constexpr (BSUM) {
    if (a.dbias) {
#pragma unroll
      for (int j = 0; j < (CELL == 0 ? 4 * NCH : 4); ++j) {
        f32x4 v = bsum[j];
#pragma unroll
        for (int m = 1; m < 16; m <<= 1)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], m, 64);   // over the 16 paths (arow) of the quad's lanes
        const int u0 = (CELL == 0) ? 64 * (j >> 2) + 16 * w + 4 * ag : 64 * w + 16 * j + 4 * ag;
        const int col = (CELL == 0) ? (j & 3) * H + u0 : u0;
        if (arow == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (u0 + r < H) {
              atomicAdd(a.dbias + col + r, v[r]);
              if (CELL == 1 && a.dbias2) atomicAdd(a.dbias2 + col + r, v[r]);
            }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// W [R][C] -> WT [C][ldo >= R] (64 x 64 tiles through LDS)
__global__ __launch_bounds__(256) void k_transpose_f32(const float* __restrict__ W, float* __restrict__ WT, int R, int C, int ldo) {
  __shared__ float t[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    t[r][c] = (r0 + r < R && c0 + c < C) ? W[(int64_t)(r0 + r) * C + c0 + c] : 0.f;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int c = e >> 6, r = e & 63;
    if (c0 + c < C && r0 + r < R) WT[(int64_t)(c0 + c) * ldo + r0 + r] = t[r][c];
  }
}

// ---- host side --------------------------------------------------------------------------------------------------------------------------------
static int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}
static size_t lds_need(int Din, int H, int R) {
  const int KX = (Din + 31) & ~31, KH = (H + 31) & ~31;
  return (size_t)ROWS * (KX + KH + 4) * 4 + (size_t)4 * R * GB + (size_t)BS_FLOATS * 4;
}
// shapes the launch takes: fp32, Din a multiple of 4 up to 256, H up to 256 (FastLSTM: up to 4 chunks of 64 units; rnn: one chunk of 256), enough
// tiles to give every CU one
// (a tile per CU at least, unless forced: below ~16 k paths the per-step launches put more workgroups on the chip than N / 64 persistent ones)
bool supported(int cell, int64_t N, int Din, int H, bool force) {
  if (cell < 0 || cell > 2) return false;
  if ((Din & 3) || Din < 16 || Din > 16 * XPMAX || H < 16 || H > 256) return false;
  if (N < (force ? (int64_t)1 : (int64_t)ROWS * num_cus()) || (N + ROWS - 1) / ROWS >= ((int64_t)1 << 31)) return false;
  return lds_need(Din, H, 2) <= (size_t)160 * 1024;
}

void forward_layer(hipStream_t s, int cell, const float* in, int64_t N, int T, int Din, int H, const float* Wi, const float* Wo, const float* bi, const float* bo,
                   float* hs, float* cs, float* act, const float* mask, int relu, bool save, bool write_all_h, const float* Wc, const float* Uc, const float* bc) {
  LArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in; a.N = N; a.T = T; a.Din = Din; a.H = H; a.Wi = Wi; a.Wo = Wo; a.bi = bi; a.bo = bo; a.hs = hs; a.cs = cs; a.act = act; a.mask = mask;
  a.Wc = Wc; a.Uc = Uc; a.bc = bc;
  a.relu = relu; a.write_all_h = (write_all_h || save) ? 1 : 0;
  a.tiles = (N + ROWS - 1) / ROWS;
  a.KX = (Din + 31) & ~31; a.KH = (H + 31) & ~31; a.PITCH = a.KX + a.KH + 4;
  a.R = lds_need(Din, H, 3) <= (size_t)160 * 1024 ? 3 : 2;
  const size_t lds = lds_need(Din, H, a.R);
  const int grid = (int)std::min<int64_t>(a.tiles, num_cus());
  typedef void (*Kern)(LArgs);
  Kern k = nullptr;
  if (cell == 1) k = save ? (Kern)k_layer<1, 1, true> : (Kern)k_layer<1, 1, false>;
  else if (cell == 2) {
    const int nch = (H + 63) / 64;
    if (nch == 1) k = save ? (Kern)k_layer<2, 1, true> : (Kern)k_layer<2, 1, false>;
    else if (nch == 2) k = save ? (Kern)k_layer<2, 2, true> : (Kern)k_layer<2, 2, false>;
    else if (nch == 3) k = save ? (Kern)k_layer<2, 3, true> : (Kern)k_layer<2, 3, false>;
    else k = save ? (Kern)k_layer<2, 4, true> : (Kern)k_layer<2, 4, false>;
  } else {
    const int nch = (H + 63) / 64;
    if (nch == 1) k = save ? (Kern)k_layer<0, 1, true> : (Kern)k_layer<0, 1, false>;
    else if (nch == 2) k = save ? (Kern)k_layer<0, 2, true> : (Kern)k_layer<0, 2, false>;
    else if (nch == 3) k = save ? (Kern)k_layer<0, 3, true> : (Kern)k_layer<0, 3, false>;
    else k = save ? (Kern)k_layer<0, 4, true> : (Kern)k_layer<0, 4, false>;
  }
  HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR), lds, s, a);
  HIP_TRY(hipGetLastError());
}

// ---- BPTT host side
bool bptt_supported(int cell, int64_t N, int H, bool force) {
  if (cell < 0 || cell > 2) return false;
  if (H < 16 || H > 256) return false;
  if (N < (force ? (int64_t)1 : (int64_t)ROWS * num_cus()) || (N + ROWS - 1) / ROWS >= ((int64_t)1 << 31)) return false;
  return true;
}
bool bptt_sums_bias(int cell, int H) { return cell >= 0 && cell <= 2; }   // (in registers; FastLSTM from three chunks up and the GRU launch: in LDS slots)
size_t bptt_scratch_floats(int H, int GH) { return (size_t)(H + 8) * GH + 1024; }   // W_o2g^T + zero slack behind its last row

// act / cs / hs / mask: the forward's saves (generic layouts); dHup: the gradient from above ([T][N][H] when up, else the head's [N][H], applied at
// t = T-1); Wo [GH][H]; wot: scratch of bptt_scratch_floats(); dA out [T][N][GH]
void bptt_layer(hipStream_t s, int cell, const float* act, const float* cs, const float* hs, const float* mask, const float* dHup, bool up, const float* Wo,
                float* wot, float* dA, int64_t N, int T, int H, int relu, const float* Uc, float* dbias, float* dbias2) {
  const int GH = cell == 1 ? H : 4 * H;                        // floats per row of dA (gru: the record's four blocks, the last one unused)
  const int WG = cell == 2 ? 3 * H : GH;                       // floats per row of the transposed weights
  HIP_TRY(hipMemsetAsync(wot, 0, bptt_scratch_floats(H, WG) * sizeof(float), s));
  if (cell == 2) {   // [c_h2h^T | o2g^T]: row = output unit, k = candidate unit | r row | z row
    hipLaunchKernelGGL(k_transpose_f32, dim3((unsigned)((H + 63) / 64), (unsigned)((H + 63) / 64)), dim3(256), 0, s, Uc, wot, H, H, WG);
    hipLaunchKernelGGL(k_transpose_f32, dim3((unsigned)((H + 63) / 64), (unsigned)((2 * H + 63) / 64)), dim3(256), 0, s, Wo, wot + H, 2 * H, H, WG);
  } else {
    hipLaunchKernelGGL(k_transpose_f32, dim3((unsigned)((H + 63) / 64), (unsigned)((GH + 63) / 64)), dim3(256), 0, s, Wo, wot, GH, H, GH);
  }
  BPArgs a;
  memset(&a, 0, sizeof(a));
  a.act = act; a.cs = cs; a.hs = hs; a.mask = mask; a.dHup = dHup; a.WoT = wot; a.dA = dA; a.N = N; a.T = T; a.H = H; a.GH = GH; a.WG = WG; a.relu = relu;
  a.dbias = bptt_sums_bias(cell, H) ? dbias : nullptr; a.dbias2 = dbias2;
  a.tiles = (N + ROWS - 1) / ROWS;
  const int nch = cell == 0 ? (H + 63) / 64 : 1;
  const int nt = cell == 0 ? nch : 4;
  const size_t lds = (size_t)ROWS * BP_LD * 4 + (size_t)4 * BP_R * nt * 1024 + 4096;   // dA chunk tile, the four waves' weight rings, the bias sums' slots
  const int grid = (int)std::min<int64_t>(a.tiles, num_cus());
  typedef void (*Kern)(BPArgs);
  Kern k = nullptr;
#define KPRN_BK(C_, N_) (up ? (Kern)k_bptt<C_, N_, true> : (Kern)k_bptt<C_, N_, false>)
  if (cell == 1) k = KPRN_BK(1, 1);
  else if (cell == 2) k = KPRN_BK(2, 1);
  else if (nch == 1) k = KPRN_BK(0, 1);
  else if (nch == 2) k = KPRN_BK(0, 2);
  else if (nch == 3) k = KPRN_BK(0, 3);
  else k = KPRN_BK(0, 4);
#undef KPRN_BK
  HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, dim3(grid), dim3(NTHR), lds, s, a);
  HIP_TRY(hipGetLastError());
}

}  // namespace lp32
