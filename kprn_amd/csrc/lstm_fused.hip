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
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }

// gather this thread's share of one step's x rows for `tile` into registers
template <int NTHREADS>
__device__ __forceinline__ void gather_load(const FwdArgs& a, int64_t tile, int t, f32x4 (&v)[1024 / NTHREADS]) {
  constexpr int PER = 1024 / NTHREADS;  // 64 rows x 16 float4 chunks
  const int c_t = a.dt >> 2, c_e = (a.dt + a.de) >> 2;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int c = threadIdx.x + k * NTHREADS;
    const int row = c >> 4, ch = c & 15;
    int64_t n = tile * MT + row;
    if (n >= a.N) n = a.N - 1;
    const int32_t* f = a.idx + (n * a.T + t) * a.F;
    f32x4 out;
    if (ch < c_t) {
      out = *(const f32x4*)(a.Wt + (int64_t)(f[a.F - a.nT - 2] - 1) * a.dt + ch * 4);
      for (int q = 1; q < a.nT; ++q) out += *(const f32x4*)(a.Wt + (int64_t)(f[a.F - a.nT - 2 + q] - 1) * a.dt + ch * 4);
    } else if (ch < c_e) {
      out = *(const f32x4*)(a.We + (int64_t)(f[a.F - 2] - 1) * a.de + (ch - c_t) * 4);
    } else {
      out = *(const f32x4*)(a.Wr + (int64_t)(f[a.F - 1] - 1) * a.dr + (ch - c_e) * 4);
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

// One LSTM step of one layer for the 64-row tile, executed by the 4 waves of a group.
//  in_buf / hp_buf / out_buf: LDS tiles [64][LDA]; hp_buf == nullptr at t == 0 (h0 = 0).
//  wi/wo: this wave's register-stationary B fragments [gate][S]; bias[gate]; c[mt][r] state.
template <bool SAVE>
__device__ __forceinline__ void lstm_step(const float* in_buf, const float* hp_buf, float* out_buf, const f32x4 (&wi)[4][4],
                                          const f32x4 (&wo)[4][4], const float (&bias)[4], float (&c)[4][4], bool first, int j, int lane,
                                          float* save_frag_t /* base for (gmt = tile*4, t, ly, j) or null */, int64_t frag_mt_stride,
                                          float* save_h_t /* &save_h[t][ly][tile*64][0] or null */, int64_t rows_valid) {
  const int arow = lane & 15, ag = lane >> 4;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    // one 16-row m-tile at a time: 4 independent accumulators (one per gate) cover the 40-cycle
    // dependent-MFMA latency; keeping m-tiles apart keeps the live register set small.
    f32x4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{bias[q], bias[q], bias[q], bias[q]};
#pragma unroll
    for (int S = 0; S < 4; ++S) {
      const f32x4 a4 = *(const f32x4*)(in_buf + (mt * 16 + arow) * LDA + S * 16 + ag * 4);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], wi[q][S][jj], acc[q], 0, 0, 0);
    }
    if (!first) {
#pragma unroll
      for (int S = 0; S < 4; ++S) {
        const f32x4 a4 = *(const f32x4*)(hp_buf + (mt * 16 + arow) * LDA + S * 16 + ag * 4);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], wo[q][S][jj], acc[q], 0, 0, 0);
      }
    }
    // ---- cell math, lane-local: acc[q][r] <-> row mt*16 + 4*ag + r, hidden col 16j + arow
    f32x4 vi, vg, vf, vo, vc, vh;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float ig = fast_sigmoid(acc[0][r]);
      const float gg = fast_tanh(acc[1][r]);
      const float fg = fast_sigmoid(acc[2][r]);
      const float og = fast_sigmoid(acc[3][r]);
      const float cp = first ? 0.f : c[mt][r];
      const float cc = fg * cp + ig * gg;
      const float hh = og * fast_tanh(cc);
      c[mt][r] = cc;
      vi[r] = ig; vg[r] = gg; vf[r] = fg; vo[r] = og; vc[r] = cc; vh[r] = hh;
      out_buf[(mt * 16 + ag * 4 + r) * LDA + j * 16 + arow] = hh;
    }
    if (SAVE) {
      float* fb = save_frag_t + (int64_t)mt * frag_mt_stride + lane * 4;
      *(f32x4*)(fb + 0 * 256) = vi;
      *(f32x4*)(fb + 1 * 256) = vg;
      *(f32x4*)(fb + 2 * 256) = vf;
      *(f32x4*)(fb + 3 * 256) = vo;
      *(f32x4*)(fb + 4 * 256) = vc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = mt * 16 + ag * 4 + r;
        if (row < rows_valid) save_h_t[(int64_t)row * DH + j * 16 + arow] = vh[r];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
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
  gather_load<NT>(a, blockIdx.x, 0, gv);
  gather_store<NT>(xbuf(0), gv);
  __syncthreads();

  int64_t tile = blockIdx.x;
  int t = 0;
  for (int64_t s = 0; s < total_slots; ++s) {
    const int par = (int)(s & 1);
    // (1) issue the gather for the NEXT slot (latency hidden under this slot's MFMAs)
    int tn = t + 1;
    int64_t tile_n = tile;
    if (tn == T) { tn = 0; tile_n += gridDim.x; }
    const bool have_next = (s + 1) < total_slots;
    if (have_next) gather_load<NT>(a, tile_n, tn, gv);

    // (2) the layers of this step, bottom-up; h tiles hand over through LDS
    const int64_t rows_valid = a.N - tile * MT;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const float* in_buf = (l == 0) ? xbuf(par) : hbuf(l - 1, par);
      const float* hp_buf = hbuf(l, par ^ 1);
      float* out_buf = hbuf(l, par);
      float* sf = nullptr; float* sh = nullptr; int64_t stride_mt = 0;
      if (SAVE) {
        stride_mt = (int64_t)T * L * 4 * 5 * 256;  // floats per global m-tile
        sf = a.save_frag + (((tile * 4) * T + t) * L + l) * (4 * 5 * 256) + (int64_t)j * (5 * 256);
        sh = a.save_h + (((int64_t)t * L + l) * a.N + tile * MT) * DH;
      }
      lstm_step<SAVE>(in_buf, hp_buf, out_buf, wi[l], wo[l], bias[l], c[l], t == 0, j, lane, sf, stride_mt, sh, rows_valid);
      if (l + 1 < L) __syncthreads();  // h_l tile complete before layer l+1 reads it
    }
    // (3) land the gathered rows of the next slot (xbuf[par^1] was last read one slot ago)
    if (have_next) gather_store<NT>(xbuf(par ^ 1), gv);
    __syncthreads();
    // (4) nn.Linear head on the finished tile: reads hbuf(L-1, par); the next slot writes par^1
    if (t == T - 1) head_tile(a, hbuf(L - 1, par), tile, j, lane);
    t = tn; tile = tile_n;
  }
}

// ---- host side ----------------------------------------------------------------------------
struct State {
  float* save_frag = nullptr;
  float* save_h = nullptr;
  int64_t cap_N = 0;
  int cap_T = 0;
  int num_cu = 0;
  bool attr_set = false;
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
  return (h->D == DH && c.H == DH && c.L >= 1 && c.L <= 2 && (c.dt % 4) == 0 && (c.de % 4) == 0 && (c.dr % 4) == 0 && T >= 1);
}

bool bwd_supported(const kprn_handle*, int) { return false; }

template <int L, bool SAVE>
static void launch_fwd(kprn_handle* h, const FwdArgs& a, int grid) {
  const size_t lds_bytes = (size_t)(2 + 2 * L) * MT * LDA * sizeof(float);
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
      HIP_TRY(hipMalloc((void**)&s->save_frag, (size_t)mts * ct * c.L * 4 * 5 * 256 * sizeof(float)));
      HIP_TRY(hipMalloc((void**)&s->save_h, (size_t)ct * c.L * (cn + 64) * DH * sizeof(float)));
      s->cap_N = cn; s->cap_T = ct;
    }
    a.save_frag = s->save_frag; a.save_h = s->save_h;
  }
  const int per_cu = 1;
  const int grid = (int)std::min<int64_t>(a.n_tiles, (int64_t)s->num_cu * per_cu);
  ProfScope ps(h, "lstm_fused_fwd");
  if (c.L == 1) { if (save) launch_fwd<1, true>(h, a, grid); else launch_fwd<1, false>(h, a, grid); }
  else { if (save) launch_fwd<2, true>(h, a, grid); else launch_fwd<2, false>(h, a, grid); }
}

void backward(kprn_handle*, const kprn_batch*, int) { throw KprnError{KPRN_E_UNSUPPORTED, "fused backward not built"}; }
void params_changed(kprn_handle*) {}
void release(kprn_handle* h) {
  State* s = (State*)h->fused_state;
  if (!s) return;
  if (s->save_frag) hipFree(s->save_frag);
  if (s->save_h) hipFree(s->save_h);
  delete s;
  h->fused_state = nullptr;
}

}  // namespace fused
