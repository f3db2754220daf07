// Identical-prefix steps of the fused LSTM path (gfx950; forward: lstm_fused_fwd.hip, backward: lstm_fused_bwd.hip).
//
// A left-padded path set (movie_data_format.py:250-254) feeds every padded path the same id tuple for its first k steps.
// nn.Sequencer(nn.FastLSTM) (model/OneModel.lua:268-274) starts every path from h = c = 0, so after k such steps all
// those paths are in ONE state per layer.  The fused kernels therefore start a tile of such paths at step k
// (batch_index.hip prefix_plan orders the paths and gives every 64-path tile its k), and the shared steps are run here,
// once per batch, on a single row:
//   k_prefix_fwd   h, c after 1..kmax copies of the reference step -> per class k the recurrent half of the first executed
//                  step (W_o2g h_{k-1}: one vector, added to every row by one extra MFMA k-slot) and c_{k-1}; plus the
//                  backward factors of the prefix steps (same definition as the training saves, NPL).
//   k_prefix_bwd   BPTT through the prefix steps on the SUM over paths of what flows into them.  The backward of a
//                  step is linear in (dh, dc) with coefficients that depend on the forward values only -- identical
//                  rows, so the sum over rows can be taken first:  the fused backward hands over, per class k,
//                  sum_rows dA_k and sum_rows dc_{k-1}; this kernel adds the missing rank-1 term of step k
//                  (dW_o2g += sum dA_k (x) h_{k-1}), walks t = k-1 .. 0, and leaves the summed dx of every prefix step
//                  where the embedding backward expects it (type / relation rows directly, the entity slice in the
//                  virtual tile of DX that the batch index points at).
// Exact up to the order of fp32 additions.  One workgroup each; plain VALU dot products (a few microseconds, 1 row).
#include "lstm_fused_common.h"
#include "adam_rows_dev.h"

namespace fused {

struct PrefFwdArgs {
  int kmax;        // longest prefix of the batch (host copy of the plan header)
  int32_t ref[16]; // the reference step's ids (1-based)
  int F, nT;
  const float *Wt, *We, *Wr;
  int dt, de, dr;
  const float* Wi[2];
  const float* bi[2];
  const float* Wo[2];
  float* pfb;  // [KCAP+1][L][PFB]
  float* pfs;  // [KCAP][L][NPL][64]
  float* pfx;  // [64]
};

// Register-resident weights: thread r keeps row r of W_i2g and W_o2g of every layer (2 x 64 floats per layer).  The rows
// come in through LDS (coalesced 16-byte global loads, all four matrices requested before anything else; a thread-per-row
// read from global would touch 64 cache lines per wave instruction), so the whole kernel pays one memory round trip and
// the steps themselves are LDS + VALU only.
constexpr int LDW = DH + 4;
__device__ __forceinline__ float row_dot(const f32x4 (&w)[16], const float* v) {
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0 += w[i][0] * v[4 * i] + w[i][2] * v[4 * i + 2];
    acc1 += w[i][1] * v[4 * i + 1] + w[i][3] * v[4 * i + 3];
  }
  return acc0 + acc1;
}
__device__ __forceinline__ void rows_request(const float* __restrict__ W, f32x4 (&st)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) st[i] = *(const f32x4*)(W + (int64_t)(i * 256 + threadIdx.x) * 4);
}
// staged [256][64] matrix -> LDS, 128 rows at a time -> this thread's row
__device__ __forceinline__ void rows_land(const f32x4 (&st)[16], float* tile, f32x4 (&row)[16]) {
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();  // previous use of the tile is over
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = (i * 256 + threadIdx.x) * 4;  // element inside this half: row e / 64 (0..127)
      *(f32x4*)(tile + (e >> 6) * LDW + (e & 63)) = st[half * 8 + i];
    }
    __syncthreads();
    if ((int)(threadIdx.x >> 7) == half) {
#pragma unroll
      for (int i = 0; i < 16; ++i) row[i] = *(const f32x4*)(tile + (threadIdx.x & 127) * LDW + 4 * i);
    }
  }
}

// Layer by layer (all prefix steps of layer 0, then all of layer 1, the lower layer's h_t kept in LDS): one layer's two weight rows per thread are resident
// at a time, 128 registers instead of 256 -- the block also rides as a passenger of the row catch-up's launch (k_catchup_prefix), whose other workgroups
// should keep two to a CU.  Every dot product sees the operands it saw in the step-major order: bit-identical results.
template <int L>
__device__ __forceinline__ void prefix_fwd_block(const PrefFwdArgs& a) {
  __shared__ float x[DH], hcur[DH], cst[DH], gates[4 * DH], hhist[KCAP][DH];
  __shared__ __attribute__((aligned(16))) float wtile[128 * LDW];
  const int r = threadIdx.x;
  const int kmax = a.kmax;
  if (r < DH) {
    const int32_t* ids = a.ref;
    float v = 0.f;
    if (r < a.dt) {
      for (int q = 0; q < a.nT; ++q) v += a.Wt[(int64_t)(ids[a.F - a.nT - 2 + q] - 1) * a.dt + r];  // CAddTable over the type slots
    } else if (r < a.dt + a.de) {
      v = a.We[(int64_t)(ids[a.F - 2] - 1) * a.de + (r - a.dt)];
    } else {
      v = a.Wr[(int64_t)(ids[a.F - 1] - 1) * a.dr + (r - a.dt - a.de)];
    }
    x[r] = v;
    a.pfx[r] = v;
  }
#pragma unroll
  for (int l = 0; l < L; ++l) {
    f32x4 wi[16], wo[16];
    const float bias = a.bi[l][r];
    {   // (requested layer by layer: the staging registers of a second layer in flight would be the 128 this order saves)
      f32x4 sa[16], sb[16];
      rows_request(a.Wi[l], sa);
      rows_request(a.Wo[l], sb);
      rows_land(sa, wtile, wi);
      rows_land(sb, wtile, wo);
    }
    if (r < DH) { hcur[r] = 0.f; cst[r] = 0.f; }
    __syncthreads();
    float rec = 0.f;  // (W_o2g h_{t-1})[r]
    for (int t = 0; t < kmax; ++t) {
      const float* in = (l == 0) ? x : hhist[t];   // (layer 1 reads layer 0's h_t, written a whole layer ago)
      gates[r] = (bias + rec) + row_dot(wi, in);
      __syncthreads();
      if (r < DH) {
        // gate order of the packed pre-activations: i, g, f, o (FastLSTM; lstm_fused_fwd.hip cell_step)
        const float gi = fast_sigmoid(gates[r]), gg = fast_tanh(gates[DH + r]), gf = fast_sigmoid(gates[2 * DH + r]), go = fast_sigmoid(gates[3 * DH + r]);
        const float cp = cst[r];
        const float cn = gf * cp + gi * gg;
        const float tc = fast_tanh(cn);
        const float hh = go * tc;
        float* sv = a.pfs + (int64_t)(t * L + l) * NPL * DH + r;
        sv[0 * DH] = gi * gg * (1.0f - gi);
        sv[1 * DH] = gi * (1.0f - gg * gg);
        sv[2 * DH] = cp * gf * (1.0f - gf);
        sv[3 * DH] = hh * (1.0f - go);
        sv[4 * DH] = go * (1.0f - tc * tc);
        sv[5 * DH] = gf;
        sv[6 * DH] = hh;
        cst[r] = cn;
        hcur[r] = hh;
      }
      __syncthreads();
      // class t+1 starts behind this step: the recurrent half of its first executed step
      const float rr = row_dot(wo, hcur);
      rec = rr;
      a.pfb[((t + 1) * L + l) * PFB + r] = rr;
      if (r < DH) { a.pfb[((t + 1) * L + l) * PFB + 4 * DH + r] = cst[r]; if (l + 1 < L) hhist[t][r] = hcur[r]; }
      __syncthreads();   // (hcur is rewritten by the next step's cell)
    }
  }
}

template <int L>
__global__ __launch_bounds__(256) void k_prefix_fwd(PrefFwdArgs a) { prefix_fwd_block<L>(a); }

// The catch-up of a batch's entity rows (lazy-exact Adam, kernels_basic.hip) and this batch's prefix table in ONE launch: both sit in the serial stretch
// between the optimiser step and the next forward, the table is one workgroup's latency chain (12-16 us) and the catch-up is a chain of dependent loads whose
// time does not depend on its occupancy (measured with 72 KB of LDS per workgroup: 13.0 against 12.8 us).  Workgroup 0 = the table, the others = the rows.
// Host side: only when the reference step's entity IS the pad row -- the one row both jobs touch, zero before and after (zeroPadTokens) whichever comes first.
template <int G, int L>
__global__ __launch_bounds__(256, 2) void k_catchup_prefix(kk_dev::AdamRowsArgs ra, PrefFwdArgs pa) {
  if (blockIdx.x == 0) { prefix_fwd_block<L>(pa); return; }
  kk_dev::adam_rows_lane_block<G>(ra, (int64_t)blockIdx.x - 1);
}

struct PrefBwdArgs {
  int kmax;
  int32_t ref[16];
  int F, nT, L, T;
  int dt, de, dr;
  const float* Wi[2];
  const float* Wo[2];
  const float* pfs;
  const float* pfx;
  float* PG;  // [L][KCAP+1][PFB], consumed (left zero)
  float* r1;  // [L][KCAP][R1]: per prefix step the vectors of its rank-1 weight-gradient terms (added by k_reduce_partials)
  float *gWt, *gWe, *gWr;
  float* DXv;         // row 0 of the virtual tile of DX (fragment order): [t][4 waves][256]
  float* DXe_v;       // nullable: rows of the compact entity dx at the virtual positions (Npad T + t): [t][de]
  int entity_direct;  // no batch index in use: add the entity slice to gWe here
};

// Thread (c = tid & 63, part = tid >> 6) owns rows part*64 .. +63, column c of the layer's [256][64] weights (for the
// transposed products).  The weight-gradient terms of the prefix steps are rank-1: their vectors go to `r1`, and the slab
// reduce that follows adds sum_t u_t (x) v_t while it touches every element of the gradient anyway.
__device__ __forceinline__ float col_dot(const float (&w)[64], const float* v256, float (*red)[DH]) {
  const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
  for (int i = 0; i < 64; i += 2) { acc0 += w[i] * v256[part * 64 + i]; acc1 += w[i + 1] * v256[part * 64 + i + 1]; }
  red[part][c] = acc0 + acc1;
  __syncthreads();
  return (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);  // (valid in every thread; used by threads < 64)
}

__global__ __launch_bounds__(256) void k_prefix_bwd(PrefBwdArgs a) {
  __shared__ float dAn[4 * DH], dAg[4 * DH], up[KCAP][DH], red[4][DH], red2[4][DH];
  const int r = threadIdx.x;
  const int c = r & 63, part = r >> 6;
  const int L = a.L;
  const int kmax = a.kmax;
  const int32_t* ids = a.ref;
  for (int l = L - 1; l >= 0; --l) {
    float wi[64], wo[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      wi[i] = a.Wi[l][(int64_t)(part * 64 + i) * DH + c];
      wo[i] = a.Wo[l][(int64_t)(part * 64 + i) * DH + c];
    }
    float dA_next = 0.f;  // (summed) dA of prefix step t+1, element r
    float dc_next = 0.f;  // (summed) dc handed from prefix step t+1 to t, element r < 64
    for (int t = kmax - 1; t >= 0; --t) {
      float* pg = a.PG + ((int64_t)l * (KCAP + 1) + (t + 1)) * PFB;
      float* r1 = a.r1 + ((int64_t)l * KCAP + t) * R1;
      // everything that enters step t from step t+1: the prefix's own dA_{t+1} and the paths whose first executed step is t+1
      const float v = dA_next + pg[r];
      pg[r] = 0.f;
      dAn[r] = v;
      r1[r] = v;                                                                      // dW_o2g += dA_{t+1} (x) h_t
      float pgc = 0.f;
      if (r < DH) {
        r1[2 * 4 * DH + r] = a.pfs[((int64_t)(t * L + l) * NPL + 6) * DH + r];        // h_t
        r1[2 * 4 * DH + DH + r] = (l == 0) ? a.pfx[r] : a.pfs[((int64_t)(t * L + l - 1) * NPL + 6) * DH + r];  // in_t
        pgc = pg[4 * DH + r];
        pg[4 * DH + r] = 0.f;
      }
      __syncthreads();
      const float dh = col_dot(wo, dAn, red);  // dh_t = W_o2g^T dA_{t+1}
      if (r < DH) {
        const float dc = dc_next + pgc;
        const float dhv = dh + ((l < L - 1) ? up[t][r] : 0.f);
        const float* P = a.pfs + (int64_t)(t * L + l) * NPL * DH + r;
        const float dC = dc + dhv * P[4 * DH];
        dAg[r] = dC * P[0 * DH];
        dAg[DH + r] = dC * P[1 * DH];
        dAg[2 * DH + r] = dC * P[2 * DH];
        dAg[3 * DH + r] = dhv * P[3 * DH];
        dc_next = dC * P[5 * DH];
      }
      __syncthreads();
      dA_next = dAg[r];
      r1[4 * DH + r] = dA_next;                                                       // dW_i2g += dA_t (x) in_t, db += dA_t
      const float dx = col_dot(wi, dAg, red2);  // dx_t = W_i2g^T dA_t
      if (r < DH) {
        if (l > 0) {
          up[t][r] = dx;  // (read above by this same thread)
        } else {
          // the reference step's rows of the three tables: every skipped occurrence of prefix step t, summed
          a.DXv[((int64_t)t * 4 + (r >> 4)) * 256 + (r & 15) * 4] = dx;
          if (a.DXe_v && r >= a.dt && r < a.dt + a.de) a.DXe_v[(int64_t)t * a.de + (r - a.dt)] = dx;
          if (r < a.dt) {
            for (int q = 0; q < a.nT; ++q) unsafeAtomicAdd(a.gWt + (int64_t)(ids[a.F - a.nT - 2 + q] - 1) * a.dt + r, dx);
          } else if (r < a.dt + a.de) {
            if (a.entity_direct) unsafeAtomicAdd(a.gWe + (int64_t)(ids[a.F - 2] - 1) * a.de + (r - a.dt), dx);
          } else {
            unsafeAtomicAdd(a.gWr + (int64_t)(ids[a.F - 1] - 1) * a.dr + (r - a.dt - a.de), dx);
          }
        }
      }
      __syncthreads();  // dAn / dAg / red are rewritten by the next step
    }
    // classes above kmax never occur; what is left of PG for this layer is zero already
  }
}

// ---- host side ----
static void ensure_prefix_buffers(State* s, hipStream_t stream) {
  if (s->pfb) return;
  HIP_TRY(kprn_dev_malloc((void**)&s->pfb, (size_t)(KCAP + 1) * 2 * PFB * sizeof(float)));
  HIP_TRY(kprn_dev_malloc((void**)&s->pfs, (size_t)KCAP * 2 * NPL * DH * sizeof(float)));
  HIP_TRY(kprn_dev_malloc((void**)&s->pfx, (size_t)DH * sizeof(float)));
  HIP_TRY(kprn_dev_malloc((void**)&s->PG, (size_t)2 * (KCAP + 1) * PFB * sizeof(float)));
  HIP_TRY(kprn_dev_malloc((void**)&s->r1, (size_t)2 * KCAP * R1 * sizeof(float)));
  // ON THE ENGINE'S STREAM: hipMemset on the null stream may run asynchronously to the host, and the engine's stream is
  // non-blocking -- the zeroing could land AFTER k_prefix_fwd had filled the table (seen as a correct first pass over a new
  // engine's first batch and a second pass with every padded path 2 % off: tests/test_gpu_fullsize.py scores twice)
  HIP_TRY(hipMemsetAsync(s->PG, 0, (size_t)2 * (KCAP + 1) * PFB * sizeof(float), stream));
  HIP_TRY(hipMemsetAsync(s->pfb, 0, (size_t)(KCAP + 1) * 2 * PFB * sizeof(float), stream));
}

static void prefix_fwd_args(kprn_handle* h, const kprn_batch* b, State* s, PrefFwdArgs& a) {
  const kprn_config& c = h->cfg;
  a.kmax = b->h_kmax;
  for (int q = 0; q < 16; ++q) a.ref[q] = b->h_ref[q];
  a.F = b->F; a.nT = c.num_types;
  a.Wt = h->dense + h->off_Wt; a.We = h->We; a.Wr = h->dense + h->off_Wr;
  a.dt = c.dt; a.de = c.de; a.dr = c.dr;
  for (int l = 0; l < 2; ++l) {
    const int ll = l < c.L ? l : 0;
    a.Wi[l] = h->dense + h->layer[ll].Wi; a.bi[l] = h->dense + h->layer[ll].bi; a.Wo[l] = h->dense + h->layer[ll].Wo;
  }
  a.pfb = s->pfb; a.pfs = s->pfs; a.pfx = s->pfx;
}

// the catch-up of b's rows AND b's prefix table as one launch (k_catchup_prefix); false: not applicable, the caller launches the catch-up by itself and
// prefix_forward follows where it always did.  The caller has joined the scoring stream.
bool catch_up_with_prefix(kprn_handle* h, const kprn_batch* b, float* W, float* g, float* m, float* v, int32_t* last, int32_t t_now, const float* step_tab,
                          float b1, float b2, float eps) {
  State* s = st(h);
  const kprn_config& c = h->cfg;
  if (!h->catchup_prefix || !b->tile_k || b->h_kmax == 0 || b->n_uniq <= 0) return false;
  if (!(c.de == 32 || c.de == 64 || c.de == 128) || (((uintptr_t)W | (uintptr_t)m | (uintptr_t)v) & 15)) return false;
  if (b->h_ref[b->F - 2] != c.Ve) return false;   // (1-based id of the pad row: the last one)
  ensure_prefix_buffers(s, h->stream);
  if (s->pf_batch == b->serial) return false;
  PrefFwdArgs pa;
  prefix_fwd_args(h, b, s, pa);
  kk_dev::AdamRowsArgs ra{W, g, m, v, last, b->uniq, b->uniq + b->uniq_cap, t_now, 0, step_tab, b1, b2, eps, (int64_t)c.Ve - 1, -1.f};
  const int G = c.de / 4;
  const unsigned nb = (unsigned)(((int64_t)b->n_uniq * G + 255) / 256) + 1;
#define KPRN_CP(G_) \
  { if (c.L == 1) hipLaunchKernelGGL((k_catchup_prefix<G_, 1>), dim3(nb), dim3(256), 0, h->stream, ra, pa); \
    else hipLaunchKernelGGL((k_catchup_prefix<G_, 2>), dim3(nb), dim3(256), 0, h->stream, ra, pa); }
  if (G == 8) KPRN_CP(8) else if (G == 16) KPRN_CP(16) else KPRN_CP(32)
#undef KPRN_CP
  HIP_TRY(hipGetLastError());
  s->pf_batch = b->serial;
  return true;
}

// prefix table for (current parameters, this batch's reference step); cached until either changes
void prefix_forward(kprn_handle* h, const kprn_batch* b) {
  State* s = st(h);
  ensure_prefix_buffers(s, h->stream);
  if (s->pf_batch == b->serial) return;
  const kprn_config& c = h->cfg;
  PrefFwdArgs a;
  if (!b->tile_k || b->h_kmax == 0) { s->pf_batch = b->serial; return; }  // class 0 only: zeros since allocation, never rewritten
  join_score(h);  // a scoring pass on the side stream may still be reading the table
  prefix_fwd_args(h, b, s, a);
  ProfScope ps(h, "prefix_fwd");
  if (c.L == 1) hipLaunchKernelGGL(k_prefix_fwd<1>, dim3(1), dim3(256), 0, h->stream, a);
  else hipLaunchKernelGGL(k_prefix_fwd<2>, dim3(1), dim3(256), 0, h->stream, a);
  HIP_TRY(hipGetLastError());
  s->pf_batch = b->serial;
}

// after the fused backward of every layer: BPTT through the prefix steps on the class sums the layers left in PG.
// Returns false when the batch skips nothing (no launch).
bool prefix_backward(kprn_handle* h, const kprn_batch* b, int64_t n_tiles) {
  State* s = st(h);
  const kprn_config& c = h->cfg;
  if (!b->tile_k || b->h_kmax == 0) return false;
  PrefBwdArgs a;
  a.kmax = b->h_kmax;
  for (int q = 0; q < 16; ++q) a.ref[q] = b->h_ref[q];
  a.F = b->F; a.nT = c.num_types; a.L = c.L; a.T = b->T;
  a.dt = c.dt; a.de = c.de; a.dr = c.dr;
  float* gd = h->g_dense;
  for (int l = 0; l < 2; ++l) {
    const int ll = l < c.L ? l : 0;
    a.Wi[l] = h->dense + h->layer[ll].Wi; a.Wo[l] = h->dense + h->layer[ll].Wo;
  }
  a.pfs = s->pfs; a.pfx = s->pfx; a.PG = s->PG; a.r1 = s->r1;
  a.gWt = gd + h->off_Wt; a.gWe = h->g_We; a.gWr = gd + h->off_Wr;
  a.DXv = s->DX + (size_t)n_tiles * 4 * b->T * 4 * 256;
  a.DXe_v = s->DXe_on ? s->DXe + (size_t)n_tiles * MT * b->T * c.de : nullptr;
  a.entity_direct = (kprn_dbg_mask() & 16) ? 1 : 0;
  ProfScope ps(h, "prefix_bwd");
  hipLaunchKernelGGL(k_prefix_bwd, dim3(1), dim3(256), 0, h->stream, a);
  HIP_TRY(hipGetLastError());
  return true;
}

}  // namespace fused
