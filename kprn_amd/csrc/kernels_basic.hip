// HBM-bound and element-wise kernels of the path-scoring engine (gfx950).
// Each kernel cites the reference module it stands in for (paths under
// /root/reference/release/songPathRnn/).
#include <string.h>
#include "kprn_internal.h"
#include "adam_rows_dev.h"

namespace {

constexpr int TPB = 256;
inline unsigned nblocks(int64_t n, int per = TPB) { return (unsigned)((n + per - 1) / per); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ void lds_atomic_add(float* p, float v) {  // ds_add_f32 (not the flat aperture path)
  typedef __attribute__((address_space(3))) float lds_float;
  __hip_atomic_fetch_add((lds_float*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------
// index validation: ids must be in 1..V (int2torch.lua:60-63 makes them 1-based)
__global__ void k_validate(const int32_t* __restrict__ idx, int64_t nsteps, int F, int nT, int Vt, int Ve, int Vr, int32_t* flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsteps) return;
  const int32_t* f = idx + i * F;
  bool bad = false;
  for (int k = 0; k < nT; ++k) { int32_t v = f[F - nT - 2 + k]; bad |= (v < 1 || v > Vt); }
  { int32_t v = f[F - 2]; bad |= (v < 1 || v > Ve); }
  { int32_t v = f[F - 1]; bad |= (v < 1 || v > Vr); }
  if (bad) atomicOr(flag, 1);
}

// ---------------------------------------------------------------------------------------
// FeatureEmbedding:getEmbeddingNetworkBothEntitiesAndTypes (net/FeatureEmbedding.lua:112-121):
// x[n,t,:] = [ sum_k Wt[type_k] | We[ent] | Wr[rel] ]; one float4 (or scalar) per thread.
template <int VEC>
__global__ void k_embed(const int32_t* __restrict__ idx, int64_t N, int T, int F, int nT, const float* __restrict__ Wt,
                        const float* __restrict__ We, const float* __restrict__ Wr, int dt, int de, int dr, float* __restrict__ X,
                        int time_major) {
  const int D = dt + de + dr;
  const int DV = D / VEC;
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = N * T * DV;
  if (gid >= total) return;
  int j = (int)(gid % DV) * VEC;
  int64_t nt = gid / DV;  // = n*T + t  (input order)
  int t = (int)(nt % T);
  int64_t n = nt / T;
  const int32_t* f = idx + nt * F;
  float out[VEC];
  if (j < dt) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) out[q] = 0.f;
    for (int k = 0; k < nT; ++k) {
      const float* row = Wt + (int64_t)(f[F - nT - 2 + k] - 1) * dt + j;
#pragma unroll
      for (int q = 0; q < VEC; ++q) out[q] = (k == 0) ? row[q] : out[q] + row[q];
    }
  } else if (j < dt + de) {
    const float* row = We + (int64_t)(f[F - 2] - 1) * de + (j - dt);
#pragma unroll
    for (int q = 0; q < VEC; ++q) out[q] = row[q];
  } else {
    const float* row = Wr + (int64_t)(f[F - 1] - 1) * dr + (j - dt - de);
#pragma unroll
    for (int q = 0; q < VEC; ++q) out[q] = row[q];
  }
  float* dst = time_major ? X + ((int64_t)t * N + n) * D + j : X + nt * D + j;
#pragma unroll
  for (int q = 0; q < VEC; ++q) dst[q] = out[q];
}

// The same, one wave per (n, t) row: no 64-bit divisions per element (they were most of k_embed's time at D = 200), VEC-wide accesses
// for any dims divisible by VEC (the shipped config's 50 / 100 / 50 take VEC = 2), and -- mask != nullptr -- MaskZero's row mask
// (the row is not all zeros, k_row_nonzero) written by the same pass instead of a second sweep over X.
template <int VEC>
__global__ void k_embed_rows(const int32_t* __restrict__ idx, int64_t NT, int64_t N, int T, int F, int nT, const float* __restrict__ Wt,
                             const float* __restrict__ We, const float* __restrict__ Wr, int dt, int de, int dr, float* __restrict__ X,
                             int time_major, float* __restrict__ mask) {
  typedef float vf __attribute__((ext_vector_type(VEC)));
  const int lane = threadIdx.x & 63;
  const int64_t nt = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;   // = n*T + t (input order)
  if (nt >= NT) return;
  const int64_t n = nt / T;
  const int t = (int)(nt - n * T);
  const int32_t* f = idx + nt * F;
  const int64_t orow = time_major ? (int64_t)t * N + n : nt;
  float* dst = X + orow * (dt + de + dr);
  const float* we = We + (int64_t)(f[F - 2] - 1) * de;
  const float* wr = Wr + (int64_t)(f[F - 1] - 1) * dr;
  const float* wt0 = Wt + (int64_t)(f[F - nT - 2] - 1) * dt;
  const int DV = (dt + de + dr) / VEC;
  int nz = 0;
  for (int jv = lane; jv < DV; jv += 64) {
    const int j = jv * VEC;
    vf v;
    if (j < dt) {
      v = *(const vf*)(wt0 + j);
      for (int k = 1; k < nT; ++k) v = v + *(const vf*)(Wt + (int64_t)(f[F - nT - 2 + k] - 1) * dt + j);
    } else if (j < dt + de) {
      v = *(const vf*)(we + (j - dt));
    } else {
      v = *(const vf*)(wr + (j - dt - de));
    }
    *(vf*)(dst + j) = v;
#pragma unroll
    for (int q = 0; q < VEC; ++q) nz |= (v[q] != 0.f) ? 1 : 0;
  }
  if (mask) {
    nz = __any(nz);
    if (lane == 0) mask[orow] = nz ? 1.f : 0.f;
  }
}

// ---------------------------------------------------------------------------------------
// nn.FastLSTM gate math (Element-Research rnn): chunks [i, g, f, o]
__global__ void k_gates_fwd(float* __restrict__ act, const float* __restrict__ c_prev, float* __restrict__ c, float* __restrict__ h,
                            int64_t N, int H) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= N * H) return;
  int j = (int)(gid % H);
  int64_t n = gid / H;
  float* a = act + n * 4 * H;
  float ig = sigmoidf_(a[j]);
  float gg = tanhf(a[H + j]);
  float fg = sigmoidf_(a[2 * H + j]);
  float og = sigmoidf_(a[3 * H + j]);
  float cp = c_prev ? c_prev[gid] : 0.f;
  float cc = fg * cp + ig * gg;
  float hh = og * tanhf(cc);
  a[j] = ig; a[H + j] = gg; a[2 * H + j] = fg; a[3 * H + j] = og;
  c[gid] = cc;
  h[gid] = hh;
}

__global__ void k_gates_bwd(const float* __restrict__ act, const float* __restrict__ c, const float* __restrict__ c_prev,
                            const float* __restrict__ dH_up, float* __restrict__ dH, float* __restrict__ dC, float* __restrict__ dA,
                            int64_t N, int H) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= N * H) return;
  int j = (int)(gid % H);
  int64_t n = gid / H;
  const float* a = act + n * 4 * H;
  float ig = a[j], gg = a[H + j], fg = a[2 * H + j], og = a[3 * H + j];
  float tc = tanhf(c[gid]);
  float dh = dH[gid] + (dH_up ? dH_up[gid] : 0.f);
  float dO = dh * tc;
  float dc = dC[gid] + dh * og * (1.f - tc * tc);
  float cp = c_prev ? c_prev[gid] : 0.f;
  float* d = dA + n * 4 * H;
  d[j] = dc * gg * ig * (1.f - ig);
  d[H + j] = dc * ig * (1.f - gg * gg);
  d[2 * H + j] = dc * cp * fg * (1.f - fg);
  d[3 * H + j] = dO * og * (1.f - og);
  dC[gid] = dc * fg;
  dH[gid] = 0.f;  // refilled with the recurrent gradient dA * Wo by the caller (t > 0)
}

__global__ void k_add_bias(float* __restrict__ Y, const float* __restrict__ b, int64_t total, int cols) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  Y[gid] += b[gid % cols];
}

// out[c] += sum_r A[r][c]; block = 256 threads handles 256 rows x 64-col strip with LDS-free partials
__global__ void k_colsum(const float* __restrict__ A, int64_t rows, int cols, int64_t ld, float* __restrict__ out, int rows_per_block, float* __restrict__ out2) {
  int c = blockIdx.y * 64 + (threadIdx.x & 63);
  int sub = threadIdx.x >> 6;  // 4 row-subgroups
  int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float acc = 0.f;
  if (c < cols) {
    // eight rows in flight per lane (one 4-byte load each: the dependent add chain of a rolled loop left the kernel at 1.3 TB/s)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    int64_t r = r0 + sub;
    const float* p = A + r * ld + c;
    const int64_t st = 4 * ld;
    for (; r + 28 < r1; r += 32, p += 8 * st) {
      a0 += p[0]; a1 += p[st]; a2 += p[2 * st]; a3 += p[3 * st]; a4 += p[4 * st]; a5 += p[5 * st]; a6 += p[6 * st]; a7 += p[7 * st];
    }
    for (; r < r1; r += 4, p += st) a0 += *p;
    acc = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  }
  __shared__ float red[4][64];
  red[sub][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sub == 0 && c < cols) {
    float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    unsafeAtomicAdd(out + c, v);
    if (out2) unsafeAtomicAdd(out2 + c, v);   // (a second vector that sees the same gradient: the rnn cell's h2h.bias next to i2h.bias)
  }
}

// ---------------------------------------------------------------------------------------
// rnnType "rnn" (OneModel.lua:240-266): nn.Recurrence(nn.MaskZero(act(i2h x + h2h h'), 1)), one step of one layer.
// mask[n] = the step input row n is not all zeros (MaskZero zeroes the output rows -- and their gradients -- of
// all-zero input rows: pad steps have zero embeddings after zeroPadTokens).  One wave per row.
__global__ void k_row_nonzero(const float* __restrict__ in, int64_t N, int D, float* __restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (n >= N) return;
  int nz = 0;
  for (int k = lane; k < D; k += 64) nz |= (in[n * D + k] != 0.f) ? 1 : 0;
  nz = __any(nz);
  if (lane == 0) mask[n] = nz ? 1.f : 0.f;
}

// pre[n][j] (in: i2h x + i2h.b (+ h2h h')) += h2h.b[j]; h = mask * act(pre)
__global__ void k_rnn_cell_fwd(float* __restrict__ pre, const float* __restrict__ bh, const float* __restrict__ mask, float* __restrict__ h,
                               int64_t N, int H, int relu) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int j = (int)(i - n * H);
  const float a = pre[i] + bh[j];
  pre[i] = a;
  const float v = relu ? fmaxf(a, 0.f) : tanhf(a);
  h[i] = (mask[n] != 0.f) ? v : 0.f;
}

// dA = mask * (dH + dH_up) * act'(pre)
__global__ void k_rnn_cell_bwd(const float* __restrict__ pre, const float* __restrict__ hcur, const float* __restrict__ mask,
                               const float* __restrict__ dH_up, const float* __restrict__ dH, float* __restrict__ dA, int64_t N, int H, int relu) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  float d = dH[i];
  if (dH_up) d += dH_up[i];
  const float der = relu ? (pre[i] > 0.f ? 1.f : 0.f) : (1.f - hcur[i] * hcur[i]);
  dA[i] = (mask[n] != 0.f) ? d * der : 0.f;
}

// ---------------------------------------------------------------------------------------
// rnnType "gru" (OneModel.lua:237-238, nn.GRU): one step of one layer on the step record a[n][4H] = [r | z | n | r*h'].
//   [r; z] = sigmoid(i2g x + o2g h');  n = tanh(c_i2h x + c_h2h (r * h'));  h = (1 - z) n + z h'
__global__ void k_gru_gates_fwd(float* __restrict__ a, const float* __restrict__ hp, int64_t N, int H) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int j = (int)(i - n * H);
  float* row = a + n * 4 * H;
  const float r = sigmoidf_(row[j]);
  const float z = sigmoidf_(row[H + j]);
  row[j] = r;
  row[H + j] = z;
  row[3 * H + j] = hp ? r * hp[i] : 0.f;
}

__global__ void k_gru_out_fwd(float* __restrict__ a, const float* __restrict__ hp, float* __restrict__ h, int64_t N, int H) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int j = (int)(i - n * H);
  float* row = a + n * 4 * H;
  const float nn = tanhf(row[2 * H + j]);
  row[2 * H + j] = nn;
  const float z = row[H + j];
  h[i] = (1.f - z) * nn + z * (hp ? hp[i] : 0.f);
}

// dh = dH (+ dH_up): d pre_n -> dA[.,2H..3H), d pre_z -> dA[.,H..2H), direct path dh z -> dHdir
__global__ void k_gru_bwd1(const float* __restrict__ a, const float* __restrict__ hp, const float* __restrict__ dH, const float* __restrict__ dH_up,
                           float* __restrict__ dA, float* __restrict__ dHdir, int64_t N, int H) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int j = (int)(i - n * H);
  const float* row = a + n * 4 * H;
  float* drow = dA + n * 4 * H;
  float dh = dH[i];
  if (dH_up) dh += dH_up[i];
  const float z = row[H + j], nn = row[2 * H + j];
  const float hpv = hp ? hp[i] : 0.f;
  drow[2 * H + j] = dh * (1.f - z) * (1.f - nn * nn);
  drow[H + j] = dh * (hpv - nn) * z * (1.f - z);
  drow[3 * H + j] = 0.f;
  dHdir[i] = dh * z;
}

// d(r*h') (dA[.,3H..4H), from the candidate's recurrent GEMM) -> d pre_r -> dA[.,0..H); dH = direct + d(r*h') r
__global__ void k_gru_bwd2(const float* __restrict__ a, const float* __restrict__ hp, float* __restrict__ dA, const float* __restrict__ dHdir,
                           float* __restrict__ dH, int64_t N, int H) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * H) return;
  const int64_t n = i / H;
  const int j = (int)(i - n * H);
  const float* row = a + n * 4 * H;
  float* drow = dA + n * 4 * H;
  const float r = row[j];
  const float drh = hp ? drow[3 * H + j] : 0.f;
  drow[j] = hp ? drh * hp[i] * r * (1.f - r) : 0.f;
  dH[i] = dHdir[i] + drh * r;
}

// ---------------------------------------------------------------------------------------
// reducer over the P paths of a pair + nn.Sigmoid (OneModel.lua:284-294):
//   2: module/LogSumExp.lua:13-27   0: nn.Max(2)   1: module/TopK.lua:17-24 + nn.Mean(2)
__device__ float reduce_col(const float* s, int P, int C, int reducer, int K) {
  if (reducer == 2) {
    float m = s[0];
    for (int p = 1; p < P; ++p) m = fmaxf(m, s[(int64_t)p * C]);
    float sum = 0.f;
    for (int p = 0; p < P; ++p) sum += expf(s[(int64_t)p * C] - m);
    return logf(sum) + m;
  } else if (reducer == 0) {
    float m = s[0];
    for (int p = 1; p < P; ++p) m = fmaxf(m, s[(int64_t)p * C]);
    return m;
  } else {
    int kk = K < P ? K : P;
    // k largest by repeated selection with an exclusion bound (value, index) -- P is small (<= 28)
    float acc = 0.f;
    float last_v = INFINITY; int last_i = -1;
    for (int q = 0; q < kk; ++q) {
      float best = -INFINITY; int bi = -1;
      for (int p = 0; p < P; ++p) {
        float v = s[(int64_t)p * C];
        bool after = (v < last_v) || (v == last_v && p > last_i);
        if (after && (bi < 0 || v > best)) { best = v; bi = p; }
      }
      acc += best; last_v = best; last_i = bi;
    }
    return acc / (float)kk;
  }
}

// + nn.Select(2, classId) (MyOptimizer.lua:126 / test_from_checkpoint.lua:82): sel[b] = probs[b][cid]
__global__ void k_pool(const float* __restrict__ S, int B, int P, int C, int reducer, int K, float* __restrict__ pooled, float* __restrict__ probs,
                       int cid, float* __restrict__ sel, float* __restrict__ sel_host) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * C) return;
  int c = (int)(gid % C);
  int64_t b = gid / C;
  float y = reduce_col(S + b * P * C + c, P, C, reducer, K);
  float pr = sigmoidf_(y);
  pooled[gid] = y;
  probs[gid] = pr;
  if (sel && c == cid) sel[b] = pr;
  if (sel_host && c == cid) sel_host[b] = pr;   // page-locked mirror (kprn_forward_batch hands the probabilities out without a copy operation)
}

// nn.Select(2, classId) first: only the selected class is reduced (what model:forward hands the caller, test_from_checkpoint.lua:82)
__global__ void k_pool_sel(const float* __restrict__ S, int B, int P, int C, int reducer, int K, int cid, float* __restrict__ sel, float* __restrict__ sel_host) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float pr = sigmoidf_(reduce_col(S + (int64_t)b * P * C + cid, P, C, reducer, K));
  sel[b] = pr;
  if (sel_host) sel_host[b] = pr;
}

// ---------------------------------------------------------------------------------------
// The whole loss stage of a training step in one launch:
//   A  reducer over the P paths + nn.Sigmoid + nn.Select for every class          (OneModel.lua:284-294, MyOptimizer.lua:126)
//   B  nn.BCECriterion forward / backward on column classId, back through sigmoid and reducer -> dS[n]   (MyOptimizer.lua:193-195)
//   C  nn.Linear(H,46) backward restricted to that column: gW[cid][:] += sum_n dS[n] hT[n][:], gb[cid] += sum_n dS[n]
//   D  loss = fixed-order sum of the per-pair terms: per-workgroup partials, summed in index order by the last workgroup
// (was five launches: pool, select, bce, sum, head_bwd).  One workgroup = LOSS_PPW pairs.  hT may be null (generic pipeline: its own head backward).
// d[q] = d loss / d s[q]; slot (nullable): the pair's q-th gradient goes to dS[slot[q]] instead (fused path: tile slot of the path)
__device__ float bce_pair(const float* __restrict__ s, int P, int C, int reducer, int K, int literal, float invB, float t, float* __restrict__ dpair,
                          float* __restrict__ dS, const int32_t* __restrict__ slot, float* lossterm) {
  auto d = [&](int q) -> float& { return slot ? dS[slot[q]] : dpair[q]; };
  const float eps = 1e-12f;
  const float y = reduce_col(s, P, C, reducer, K);
  const float p = sigmoidf_(y);
  *lossterm = -(t * logf(p + eps) + (1.f - t) * logf(1.f - p + eps)) * invB;
  float dy;
  if (literal) {
    float dp = -(t - p) / ((1.f - p + eps) * (p + eps)) * invB;
    dy = dp * p * (1.f - p);
  } else {
    dy = (p - t) * invB;
  }
  if (reducer == 2) {
    float m = s[0];
    for (int q = 1; q < P; ++q) m = fmaxf(m, s[(int64_t)q * C]);
    float sum = 0.f;
    for (int q = 0; q < P; ++q) sum += expf(s[(int64_t)q * C] - m);
    for (int q = 0; q < P; ++q) d(q) = expf(s[(int64_t)q * C] - m) / sum * dy;
  } else if (reducer == 0) {
    int arg = 0;
    for (int q = 1; q < P; ++q) if (s[(int64_t)q * C] > s[(int64_t)arg * C]) arg = q;
    for (int q = 0; q < P; ++q) d(q) = (q == arg) ? dy : 0.f;
  } else {
    int kk = K < P ? K : P;
    for (int q = 0; q < P; ++q) d(q) = 0.f;
    float last_v = INFINITY; int last_i = -1;
    for (int r = 0; r < kk; ++r) {
      float best = -INFINITY; int bi = -1;
      for (int q = 0; q < P; ++q) {
        float v = s[(int64_t)q * C];
        bool after = (v < last_v) || (v == last_v && q > last_i);
        if (after && (bi < 0 || v > best)) { best = v; bi = q; }
      }
      d(bi) = dy / (float)kk; last_v = best; last_i = bi;
    }
  }
  return p;
}

constexpr int LOSS_PPW = 16;  // pairs per workgroup of the loss stage: small on purpose (latency-bound: many workgroups in flight)
__global__ __launch_bounds__(256) void k_loss_stage(const float* __restrict__ S, const float* __restrict__ labels, const float* __restrict__ hT,
                                                    int B, int P, int C, int H, int cid, int reducer, int K, int literal, float invB,
                                                    float* __restrict__ pooled, float* __restrict__ probs, float* __restrict__ sel,
                                                    float* __restrict__ dS, const int32_t* __restrict__ slot_of, float* __restrict__ gW_row,
                                                    float* __restrict__ gb_c, float* __restrict__ partial, int n_loss_blocks, kk::TransposeJob tj, float* __restrict__ partial_host,
                                                    kk::PoolJob pj) {
  if ((int)blockIdx.x >= n_loss_blocks + 64 * tj.n) {  // (workgroup-uniform) passenger: the pooling stage of a scoring pass (k_pool_sel's arithmetic)
    const int b = ((int)blockIdx.x - n_loss_blocks - 64 * tj.n) * 256 + (int)threadIdx.x;
    if (b >= pj.B) return;
    const float pr = sigmoidf_(reduce_col(pj.S + (int64_t)b * pj.P * C + pj.cid, pj.P, C, reducer, K));
    pj.sel[b] = pr;
    if (pj.sel_host) pj.sel_host[b] = pr;
    return;
  }
  if ((int)blockIdx.x >= n_loss_blocks) {  // (workgroup-uniform) the passenger job: 256x64 weight transposes
    const int rb = blockIdx.x - n_loss_blocks;
    const int m = rb >> 6, i = (rb & 63) * 256 + threadIdx.x;  // over the 64*256 outputs of matrix m
    tj.WT[m][i] = tj.W[m][(i & 255) * 64 + (i >> 8)];
    return;
  }
  __shared__ float lossw[LOSS_PPW];
  __shared__ float red[4][65];
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * LOSS_PPW;
  const int nb = (B - b0 < LOSS_PPW) ? (B - b0) : LOSS_PPW;
  // A (only when the caller wants every class: training needs column classId alone, which B computes)
  if (pooled) for (int i = tid; i < nb * C; i += 256) {
    const int b = b0 + i / C, c = i % C;
    const float y = reduce_col(S + (int64_t)b * P * C + c, P, C, reducer, K);
    const float pr = sigmoidf_(y);
    pooled[(int64_t)b * C + c] = y;
    probs[(int64_t)b * C + c] = pr;
    if (c == cid) sel[b] = pr;
  }
  // B
  if (tid < LOSS_PPW) {
    float lt = 0.f;
    if (tid < nb) {
      const int b = b0 + tid;
      const float pr = bce_pair(S + (int64_t)b * P * C + cid, P, C, reducer, K, literal, invB, labels[b], dS + (int64_t)b * P, dS,
                                slot_of ? slot_of + (int64_t)b * P : nullptr, &lt);
      if (!pooled) sel[b] = pr;
    }
    lossw[tid] = lt;
  }
  __syncthreads();  // this workgroup's dS rows are visible to it
  // C
  if (hT) {
    const int64_t n0 = (int64_t)b0 * P, n1 = n0 + (int64_t)nb * P;
    const int sub = tid >> 6, col = tid & 63;
    for (int c0 = 0; c0 < H; c0 += 64) {
      const int cc = c0 + col;
      float acc = 0.f, sd = 0.f;
      for (int64_t n = n0 + sub; n < n1; n += 4) {
        const float d = dS[n];
        sd += d;
        if (cc < H) acc += d * hT[n * H + cc];
      }
      red[sub][col] = acc;
      if (col == 0) red[sub][64] = sd;
      __syncthreads();
      if (sub == 0) {
        if (cc < H) unsafeAtomicAdd(gW_row + cc, (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]));
        if (col == 0 && c0 == 0) unsafeAtomicAdd(gb_c, (red[0][64] + red[1][64]) + (red[2][64] + red[3][64]));
      }
      __syncthreads();
    }
  }
  // D: per-workgroup partial, summed in index order by k_sum_partials when the loss is asked for (a last-workgroup
  //    reduction here needs a device-scope release per workgroup = an L2 write-back each: measured 30 us)
  if (tid == 0) {
    float s = 0.f;
    for (int i = 0; i < LOSS_PPW; ++i) s += lossw[i];
    partial[blockIdx.x] = s;
    if (partial_host) partial_host[blockIdx.x] = s;   // page-locked mirror: the host adds the partials itself (kprn_train_step's early loss)
  }
}

// loss = fixed-order sum of the per-workgroup partials (reproducible)
__global__ void k_sum_partials(const float* __restrict__ partial, int n, float* __restrict__ out, int accumulate) {
  __shared__ float pl[256];
  float s = 0.f;
  for (int base = 0; base < n; base += 256) {
    const int i = base + (int)threadIdx.x;
    pl[threadIdx.x] = (i < n) ? partial[i] : 0.f;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int m = (n - base < 256) ? (n - base) : 256;
      for (int k = 0; k < m; ++k) s += pl[k];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = s;
    if (accumulate) { out[1] += s; out[2] += 1.0f; }   // running sum / count of the losses since the last kprn_read_loss_sum
  }
}

// nn.Linear(H,46) backward restricted to the selected column (OneModel.lua:275; Select at MyOptimizer.lua:126)
__global__ void k_head_bwd(const float* __restrict__ dS, const float* __restrict__ hT, const float* __restrict__ Wout, int64_t N, int H,
                           int cid, float* __restrict__ dH, float* __restrict__ gWout, float* __restrict__ gbout, int rows_per_block) {
  int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    float w = Wout[(int64_t)cid * H + j];
    float acc = 0.f;
    for (int64_t n = r0; n < r1; ++n) {
      float d = dS[n];
      acc += d * hT[n * H + j];
      if (dH) dH[n * H + j] = d * w;
    }
    unsafeAtomicAdd(gWout + (int64_t)cid * H + j, acc);
  }
  if (threadIdx.x == 0) {
    float acc = 0.f;
    for (int64_t n = r0; n < r1; ++n) acc += dS[n];
    unsafeAtomicAdd(gbout + cid, acc);
  }
}

// The same for H <= 512 in one pass at memory speed: a workgroup takes rows_per_block rows, its four waves every fourth row of them, a lane the
// columns lane + 64 g; the waves' sums meet in LDS and leave as H atomics per workgroup.  (k_head_bwd above: 64 rows per workgroup = 1 024 atomics
// on each of the H addresses of the selected row at 65 536 paths -- same-address atomics serialise in L2 -- and one load in flight per thread:
// 0.08-0.09 ms for 100 MB on configs[3] / the shipped shape.)  Loads are unconditional (idle lanes re-read column H - 1 and drop the sum).
template <int NG, bool HAS_DH>
__global__ __launch_bounds__(256) void k_head_bwd_w(const float* __restrict__ dS, const float* __restrict__ hT, const float* __restrict__ Wout, int64_t N, int H,
                                                    int cid, float* __restrict__ dH, float* __restrict__ gWout, float* __restrict__ gbout, int rows_per_block) {
  __shared__ float red[4][NG * 64 + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
  float acc[NG], w[NG];
  int jc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int j = g * 64 + lane;
    jc[g] = j < H ? j : H - 1;
    acc[g] = 0.f;
    w[g] = Wout[(int64_t)cid * H + jc[g]];
  }
  float bsum = 0.f;
#pragma unroll 4
  for (int64_t n = r0 + wave; n < r1; n += 4) {
    const float d = dS[n];
    bsum += d;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      acc[g] += d * hT[n * H + jc[g]];
      if (HAS_DH && g * 64 + lane < H) dH[n * H + g * 64 + lane] = d * w[g];
    }
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) red[wave][g * 64 + lane] = acc[g];
  if (lane == 0) red[wave][NG * 64] = bsum;
  __syncthreads();
  for (int j = threadIdx.x; j < H; j += 256) unsafeAtomicAdd(gWout + (int64_t)cid * H + j, red[0][j] + red[1][j] + red[2][j] + red[3][j]);
  if (threadIdx.x == 0) unsafeAtomicAdd(gbout + cid, red[0][NG * 64] + red[1][NG * 64] + red[2][NG * 64] + red[3][NG * 64]);
}

// ---------------------------------------------------------------------------------------
// nn.LookupTable backward = scatter-add (duplicates accumulate).  The two tiny tables (types,
// relations) are reduced in LDS per block first; entity rows go straight to L2 atomics.
__global__ void k_embed_scatter(const int32_t* __restrict__ idx, int64_t N, int T, int F, int nT, const float* __restrict__ dX, int dt, int de,
                                int dr, int Vt, int Vr, float* __restrict__ gWt, float* __restrict__ gWe, float* __restrict__ gWr,
                                int use_lds, int steps_per_block, int skip_entity, int skip_type, int skip_rel) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = dt + de + dr;
  const int nt_small = Vt * dt, nr_small = Vr * dr;
  if (use_lds) {
    for (int i = threadIdx.x; i < nt_small + nr_small; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
  }
  int64_t s0 = (int64_t)blockIdx.x * steps_per_block;
  int64_t total = N * T;
  int64_t s1 = s0 + steps_per_block < total ? s0 + steps_per_block : total;
  int64_t nelem = (s1 - s0) * D;
  for (int64_t e = threadIdx.x; e < nelem; e += blockDim.x) {
    int64_t step = s0 + e / D;  // time-major linear index = t*N + n
    int j = (int)(e % D);
    int t = (int)(step / N);
    int64_t n = step % N;
    const int32_t* f = idx + (n * T + t) * F;
    if ((j < dt) ? skip_type : ((j < dt + de) ? skip_entity : skip_rel)) continue;   // (that table is handled elsewhere)
    float v = dX[step * D + j];
    if (j < dt) {
      for (int k = 0; k < nT; ++k) {
        int row = f[F - nT - 2 + k] - 1;
        if (use_lds) lds_atomic_add(&lds[row * dt + j], v);
        else unsafeAtomicAdd(gWt + (int64_t)row * dt + j, v);
      }
    } else if (j < dt + de) {
      if (!skip_entity) unsafeAtomicAdd(gWe + (int64_t)(f[F - 2] - 1) * de + (j - dt), v);  // else: bidx::entity_grad (no atomics)
    } else {
      int row = f[F - 1] - 1;
      if (use_lds) lds_atomic_add(&lds[nt_small + row * dr + (j - dt - de)], v);
      else unsafeAtomicAdd(gWr + (int64_t)row * dr + (j - dt - de), v);
    }
  }
  if (use_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < nt_small; i += blockDim.x) { float v = lds[i]; if (v != 0.f) unsafeAtomicAdd(gWt + i, v); }
    for (int i = threadIdx.x; i < nr_small; i += blockDim.x) { float v = lds[nt_small + i]; if (v != 0.f) unsafeAtomicAdd(gWr + i, v); }
  }
}

// ---------------------------------------------------------------------------------------
__global__ void k_sumsq(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += x[i] * x[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(out, acc);
}

__global__ void k_sumsq_rows(const float* __restrict__ G, const int32_t* __restrict__ rows, const int32_t* __restrict__ count, int d,
                             float* __restrict__ out) {
  int nrows = *count;
  float acc = 0.f;
  int64_t total = (int64_t)nrows * d;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float v = G[(int64_t)rows[i / d] * d + (i % d)];
    acc += v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(out, acc);
}

// ---- optimisers (optim.adam / optim.adagrad, OneModel.lua:347-361; MyOptimizer.lua:196-218) ----
#pragma clang fp contract(off)
__device__ __forceinline__ float clip_factor(const float* norm2, float clip) {
  if (!norm2) return 1.f;
  float nrm = sqrtf(*norm2);
  return (nrm > clip) ? clip / nrm : 1.f;
}

using kk_dev::adam_elem;   // (adam_rows_dev.h: shared with lstm_fused_prefix.hip's catch-up + prefix launch)

// consume != 0: the gradient is zeroed as it is used (zeroGradParameters of the next trainBatch, MyOptimizer.lua:186, done here);
// [z0,z0+zn0) and [z1,z1+zn1): pad rows of the arena, re-zeroed after the update (zeroPadTokens, MyOptimizer.lua:219);
// tab_slot: this step's entry of the device step-size table (read by the lazy row replay).
struct AdamDenseJob {   // the dense arena's share of an optimiser step (k_adam_dense, or extra workgroups of the row update's launch)
  float *x, *g, *m, *v; int64_t n;
  float step, b1, b2, eps; const float* norm2; float clip, l2; int reg, consume;
  int64_t z0; int zn0; int64_t z1; int zn1; float* tab_slot;
};
__device__ __forceinline__ void adam_dense_block(const AdamDenseJob& a, int64_t block) {
  const int64_t i = block * blockDim.x + threadIdx.x;
  if (i == 0 && a.tab_slot) *a.tab_slot = a.step;
  if (i >= a.n) return;
  float gi = a.g[i];
  float xi = a.x[i];
  if (a.reg) { gi = gi * clip_factor(a.norm2, a.clip); gi = gi + a.l2 * xi; }
  float mi = a.m[i], vi = a.v[i];
  adam_elem(xi, mi, vi, gi, a.step, a.b1, a.b2, a.eps);
  if ((i >= a.z0 && i < a.z0 + a.zn0) || (i >= a.z1 && i < a.z1 + a.zn1)) xi = 0.f;
  a.x[i] = xi; a.m[i] = mi; a.v[i] = vi;
  if (a.consume) a.g[i] = 0.f;
}
__global__ void k_adam_dense(AdamDenseJob a) { adam_dense_block(a, blockIdx.x); }

__global__ void k_adagrad_dense(float* __restrict__ x, float* __restrict__ g, float* __restrict__ G, int64_t n, float clr,
                                const float* __restrict__ norm2, float clip, float l2, int reg, int consume, int64_t z0, int zn0, int64_t z1,
                                int zn1) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i];
  float xi = x[i];
  if (reg) { gi = gi * clip_factor(norm2, clip); gi = gi + l2 * xi; }
  float Gi = G[i] + gi * gi;
  G[i] = Gi;
  xi = xi - clr * gi / (sqrtf(Gi) + 1e-10f);
  if ((i >= z0 && i < z0 + zn0) || (i >= z1 && i < z1 + zn1)) xi = 0.f;
  x[i] = xi;
  if (consume) g[i] = 0.f;
}

// lazy-exact Adam over a list of rows: one wave per row.  Replays the steps the row missed
// (gradient 0) with the same arithmetic as k_adam_dense, then (apply_step) applies step t_now
// with the accumulated gradient and clears it.  Result is bit-identical to updating every
// row at every step, which is what optim.adam does to the flat vector.
__global__ void k_adam_rows(float* __restrict__ W, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            int32_t* __restrict__ last, const int32_t* __restrict__ rows, const int32_t* __restrict__ count, int d,
                            int32_t t_now, int apply_step, const float* __restrict__ step_tab, float b1, float b2, float eps, int64_t pad_row) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= *count) return;
  const int64_t r = rows[wave];
  const int32_t l = last[r];
  const int32_t upto = apply_step ? t_now - 1 : t_now;  // replay (l, upto] with g = 0
  for (int e = lane; e < d; e += 64) {
    const int64_t o = r * d + e;
    float x = W[o], mm = m[o], vv = v[o];
    if (l > 0)
      for (int32_t k = l + 1; k <= upto; ++k) adam_elem(x, mm, vv, 0.f, step_tab[k], b1, b2, eps);
    if (apply_step) {
      adam_elem(x, mm, vv, g[o], step_tab[t_now], b1, b2, eps);
      g[o] = 0.f;
    }
    if (r == pad_row) x = 0.f;  // zeroPadTokens after every step the row lived through (MyOptimizer.lua:219)
    W[o] = x; m[o] = mm; v[o] = vv;
  }
  if (lane == 0) last[r] = t_now;
}

// the same for rows of d = 4 G floats handled by G = 8 / 16 / 32 lanes with 16-byte accesses (d = 32: eight rows per wave instead of one
// half-empty wave per row; the per-element arithmetic is adam_elem's, so the result stays bit-identical to the dense sweep)
// dense (nullable in effect: n_dense_blocks = 0): the dense arena's update rides in the same launch as workgroups [rows_blocks, rows_blocks + n_dense_blocks)
// -- two latency-bound launches of the step's serial tail become one.  The row update then takes THIS step's size from `step_now` (the dense job writes the
// table entry for later replays; inside one launch nothing orders that store before a row's read of it); step_now < 0: read it from the table.
template <int G>
__global__ void k_adam_rows_v(float* __restrict__ W, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                              int32_t* __restrict__ last, const int32_t* __restrict__ rows, const int32_t* __restrict__ count,
                              int32_t t_now, int apply_step, const float* __restrict__ step_tab, float b1, float b2, float eps, int64_t pad_row,
                              int rows_blocks, float step_now, AdamDenseJob dense) {
  if ((int)blockIdx.x >= rows_blocks) { adam_dense_block(dense, (int64_t)blockIdx.x - rows_blocks); return; }
  const kk_dev::AdamRowsArgs ra{W, g, m, v, last, rows, count, t_now, apply_step, step_tab, b1, b2, eps, pad_row, step_now};
  kk_dev::adam_rows_lane_block<G>(ra, (int64_t)blockIdx.x);
}

// Data-parallel exchange, union + update in ONE launch (lazy-exact Adam, no clip / L2).  `all` = the all-gathered packed buffers
// [world][stride] of kprn_sparse_grad_pack: {count, -, -, -, ids[cap] ASCENDING (the batch index's run-length-encoded sorted keys, each
// row once), rows[cap][d]}.  One G-lane group per (rank, entry); lane j finds the row in rank j's list by binary search (world <= G), the
// positions are shared through the group; the entry of the LOWEST rank that touched the row owns it: it adds the other ranks'
// contributions in RANK ORDER (the same addition order on every replica, and the order k_add_rank's launches used), replays the row's
// skipped steps and applies this one.  No flag array over the table, no compaction, no gradient accumulator round trip.
template <int G>
__global__ void k_union_adam(const int32_t* __restrict__ all, int world, int cap, int64_t stride, float* __restrict__ W, float* __restrict__ m,
                             float* __restrict__ v, int32_t* __restrict__ last, int32_t t_now, const float* __restrict__ step_tab, float b1, float b2,
                             float eps, int64_t pad_row) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t slot = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  const int j = threadIdx.x % G;
  const int base = (threadIdx.x & 63) & ~(G - 1);
  const int r = (int)(slot / cap), k = (int)(slot - (int64_t)r * cap);
  const bool in_range = r < world;
  const int32_t* mine = all + (int64_t)(in_range ? r : 0) * stride;
  const bool live = in_range && k < mine[0];
  const int32_t row = live ? mine[4 + k] : 0;
  int pos = -1;
  if (live && j < world && j != r) {
    const int32_t* o = all + (int64_t)j * stride;
    const int n = o[0];
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (o[4 + mid] < row) lo = mid + 1; else hi = mid;
    }
    if (lo < n && o[4 + lo] == row) pos = lo;
  }
  f4 g4 = f4{0.f, 0.f, 0.f, 0.f};
  if (live) g4 = *(const f4*)((const float*)(mine + 4 + cap) + (int64_t)k * (4 * G) + 4 * j);
  bool owner = live;
  for (int q = 0; q < world; ++q) {   // every lane of the wave takes every shuffle (the groups of one wave may belong to two ranks)
    const int p = __shfl(pos, base + q, 64);
    if (p < 0 || !live || q == r) continue;
    if (q < r) owner = false;
    else if (owner) g4 = g4 + *(const f4*)((const float*)(all + (int64_t)q * stride + 4 + cap) + (int64_t)p * (4 * G) + 4 * j);
  }
  if (!owner) return;
  const int32_t l = last[row];
  const int64_t o = (int64_t)row * (4 * G) + 4 * j;
  const f4 x4 = *(const f4*)(W + o), m4 = *(const f4*)(m + o), v4 = *(const f4*)(v + o);
  float x[4] = {x4[0], x4[1], x4[2], x4[3]}, mm[4] = {m4[0], m4[1], m4[2], m4[3]}, vv[4] = {v4[0], v4[1], v4[2], v4[3]};
  if (l > 0)
    for (int32_t kk = l + 1; kk <= t_now - 1; ++kk) {
      const float st = step_tab[kk];
#pragma unroll
      for (int q = 0; q < 4; ++q) adam_elem(x[q], mm[q], vv[q], 0.f, st, b1, b2, eps);
    }
  const float st = step_tab[t_now];
#pragma unroll
  for (int q = 0; q < 4; ++q) adam_elem(x[q], mm[q], vv[q], g4[q], st, b1, b2, eps);
  if (row == pad_row) { x[0] = x[1] = x[2] = x[3] = 0.f; }  // zeroPadTokens (MyOptimizer.lua:219)
  *(f4*)(W + o) = f4{x[0], x[1], x[2], x[3]}; *(f4*)(m + o) = f4{mm[0], mm[1], mm[2], mm[3]}; *(f4*)(v + o) = f4{vv[0], vv[1], vv[2], vv[3]};
  if (j == 0) last[row] = t_now;
}

__global__ void k_adam_flush_all(float* __restrict__ W, float* __restrict__ m, float* __restrict__ v, int32_t* __restrict__ last, int64_t V,
                                 int d, int32_t t_now, const float* __restrict__ step_tab, float b1, float b2, float eps, int64_t pad_row) {
  const int lane = threadIdx.x & 63;
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (r >= V) return;
  const int32_t l = last[r];
  if (l <= 0 || l >= t_now) return;
  for (int e = lane; e < d; e += 64) {
    const int64_t o = r * d + e;
    float x = W[o], mm = m[o], vv = v[o];
    for (int32_t k = l + 1; k <= t_now; ++k) adam_elem(x, mm, vv, 0.f, step_tab[k], b1, b2, eps);
    if (r == pad_row) x = 0.f;
    W[o] = x; m[o] = mm; v[o] = vv;
  }
  if (lane == 0) last[r] = t_now;
}

__global__ void k_adagrad_rows(float* __restrict__ W, float* __restrict__ g, float* __restrict__ G, const int32_t* __restrict__ rows,
                               const int32_t* __restrict__ count, int d, float clr, int64_t pad_row) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= *count) return;
  const int64_t r = rows[wave];
  for (int e = lane; e < d; e += 64) {
    const int64_t o = r * d + e;
    float gi = g[o];
    float Gi = G[o] + gi * gi;
    G[o] = Gi;
    W[o] = (r == pad_row) ? 0.f : W[o] - clr * gi / (sqrtf(Gi) + 1e-10f);
    g[o] = 0.f;
  }
}
#pragma clang fp contract(fast)

__global__ void k_zero_row(float* __restrict__ W, int64_t row, int d) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < d) W[row * d + j] = 0.f;
}

// MyOptimizer:zeroPadTokens (MyOptimizer.lua:74-93): the three pad rows in one launch
__global__ void k_zero_pad3(float* __restrict__ a, int na, float* __restrict__ b, int nb, float* __restrict__ c, int nc) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < na) a[j] = 0.f;
  if (j < nb) b[j] = 0.f;
  if (j < nc) c[j] = 0.f;
}

__global__ void k_fill_i32(int32_t* __restrict__ x, int64_t n, int32_t v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}

// data-parallel exchange helpers -------------------------------------------------------------
__global__ void k_pack_rows(float* __restrict__ G, const int32_t* __restrict__ rows, const int32_t* __restrict__ count, int d,
                            int32_t* __restrict__ ids_out, float* __restrict__ rows_out, int32_t* __restrict__ count_out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int32_t n = *count;
  if (wave == 0 && lane == 0) *count_out = n;
  if (wave >= n) return;
  const int64_t r = rows[wave];
  if (lane == 0) ids_out[wave] = (int32_t)r;
  for (int e = lane; e < d; e += 64) {
    rows_out[wave * d + e] = G[r * d + e];
    G[r * d + e] = 0.f;
  }
}
// the same for d = 4 LG floats with LG lanes per row and 16-byte accesses; the workgroups behind the row part copy `n_tail` floats (the
// dense gradient arena riding behind the rows, kprn_api.hip dp_dense_in_pack) -- one launch for the whole packed buffer
template <int LG>
__global__ void k_pack_rows_v(float* __restrict__ G, const int32_t* __restrict__ rows, const int32_t* __restrict__ count, int32_t* __restrict__ ids_out,
                              float* __restrict__ rows_out, int32_t* __restrict__ count_out, int row_blocks, const float* __restrict__ tail_src,
                              int64_t n_tail, float* __restrict__ tail_dst) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  if ((int)blockIdx.x >= row_blocks) {
    for (int64_t i = (int64_t)(blockIdx.x - row_blocks) * blockDim.x + threadIdx.x; i < n_tail; i += (int64_t)(gridDim.x - row_blocks) * blockDim.x)
      tail_dst[i] = tail_src[i];
    return;
  }
  const int64_t slot = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LG;
  const int j = threadIdx.x % LG;
  const int32_t n = *count;
  if (slot == 0 && j == 0) *count_out = n;
  if (slot >= n) return;
  const int64_t r = rows[slot];
  if (j == 0) ids_out[slot] = (int32_t)r;
  f4* src = (f4*)(G + r * (4 * LG) + 4 * j);
  *(f4*)(rows_out + slot * (4 * LG) + 4 * j) = *src;
  *src = f4{0.f, 0.f, 0.f, 0.f};
}

__global__ void k_clear_rows(float* __restrict__ G, const int32_t* __restrict__ rows, const int32_t* __restrict__ count, int d) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (wave >= *count) return;
  const int64_t r = rows[wave];
  for (int e = lane; e < d; e += 64) G[r * d + e] = 0.f;
}

// per (64-row tile, step): leader[row] = first row of the tile with the same entity id (-1 past N).
// Used by the fused backward to fold duplicate rows before the embedding-gradient atomics.
// uniform(-a, a) init (OneModel.lua:306-309); counter-based splitmix64
__global__ void k_fill_uniform(float* __restrict__ x, int64_t n, float a, uint64_t seed, uint64_t offset) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + offset + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  float u = (float)(z >> 40) * (1.0f / 16777216.0f);  // [0,1)
  x[i] = (2.f * u - 1.f) * a;
}

}  // namespace

#define CHECK_LAUNCH() HIP_TRY(hipGetLastError())

namespace kk {

void validate_indices(hipStream_t s, const int32_t* idx, int64_t nsteps, int F, int nT, int Vt, int Ve, int Vr, int32_t* flag) {
  if (nsteps <= 0) return;
  hipLaunchKernelGGL(k_validate, dim3(nblocks(nsteps)), dim3(TPB), 0, s, idx, nsteps, F, nT, Vt, Ve, Vr, flag);
  CHECK_LAUNCH();
}


void embed_gather(hipStream_t s, const int32_t* idx, int64_t N, int T, int F, int nT, const float* Wt, const float* We, const float* Wr,
                  int dt, int de, int dr, float* X, bool time_major, float* mask) {
  if (N <= 0) return;
  const int64_t NT = N * T;
  const unsigned blocks = nblocks(NT * 64);
  const int all = dt | de | dr;
  const uintptr_t pall = (uintptr_t)Wt | (uintptr_t)We | (uintptr_t)Wr | (uintptr_t)X;
  if ((all % 4 == 0) && !(pall & 15))
    hipLaunchKernelGGL((k_embed_rows<4>), dim3(blocks), dim3(TPB), 0, s, idx, NT, N, T, F, nT, Wt, We, Wr, dt, de, dr, X, time_major ? 1 : 0, mask);
  else if ((all % 2 == 0) && !(pall & 7))
    hipLaunchKernelGGL((k_embed_rows<2>), dim3(blocks), dim3(TPB), 0, s, idx, NT, N, T, F, nT, Wt, We, Wr, dt, de, dr, X, time_major ? 1 : 0, mask);
  else
    hipLaunchKernelGGL((k_embed_rows<1>), dim3(blocks), dim3(TPB), 0, s, idx, NT, N, T, F, nT, Wt, We, Wr, dt, de, dr, X, time_major ? 1 : 0, mask);
  CHECK_LAUNCH();
}

void lstm_gates_fwd(hipStream_t s, float* act, const float* c_prev, float* c, float* h, int64_t N, int H) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_gates_fwd, dim3(nblocks(N * H)), dim3(TPB), 0, s, act, c_prev, c, h, N, H);
  CHECK_LAUNCH();
}

void lstm_gates_bwd(hipStream_t s, const float* act, const float* c, const float* c_prev, const float* dH_up, float* dH, float* dC, float* dA,
                    int64_t N, int H) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_gates_bwd, dim3(nblocks(N * H)), dim3(TPB), 0, s, act, c, c_prev, dH_up, dH, dC, dA, N, H);
  CHECK_LAUNCH();
}

void row_nonzero(hipStream_t s, const float* in, int64_t N, int D, float* mask) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_row_nonzero, dim3(nblocks(N * 64)), dim3(TPB), 0, s, in, N, D, mask);
  CHECK_LAUNCH();
}

void rnn_cell_fwd(hipStream_t s, float* pre, const float* bh, const float* mask, float* h, int64_t N, int H, int relu) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_rnn_cell_fwd, dim3(nblocks(N * H)), dim3(TPB), 0, s, pre, bh, mask, h, N, H, relu);
  CHECK_LAUNCH();
}

void rnn_cell_bwd(hipStream_t s, const float* pre, const float* hcur, const float* mask, const float* dH_up, const float* dH, float* dA, int64_t N,
                  int H, int relu) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_rnn_cell_bwd, dim3(nblocks(N * H)), dim3(TPB), 0, s, pre, hcur, mask, dH_up, dH, dA, N, H, relu);
  CHECK_LAUNCH();
}

void gru_gates_fwd(hipStream_t s, float* a, const float* hp, int64_t N, int H) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_gru_gates_fwd, dim3(nblocks(N * H)), dim3(TPB), 0, s, a, hp, N, H);
  CHECK_LAUNCH();
}
void gru_out_fwd(hipStream_t s, float* a, const float* hp, float* h, int64_t N, int H) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_gru_out_fwd, dim3(nblocks(N * H)), dim3(TPB), 0, s, a, hp, h, N, H);
  CHECK_LAUNCH();
}
void gru_bwd1(hipStream_t s, const float* a, const float* hp, const float* dH, const float* dH_up, float* dA, float* dHdir, int64_t N, int H) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_gru_bwd1, dim3(nblocks(N * H)), dim3(TPB), 0, s, a, hp, dH, dH_up, dA, dHdir, N, H);
  CHECK_LAUNCH();
}
void gru_bwd2(hipStream_t s, const float* a, const float* hp, float* dA, const float* dHdir, float* dH, int64_t N, int H) {
  if (N <= 0) return;
  hipLaunchKernelGGL(k_gru_bwd2, dim3(nblocks(N * H)), dim3(TPB), 0, s, a, hp, dA, dHdir, dH, N, H);
  CHECK_LAUNCH();
}

__global__ void k_add_into(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
void add_into(hipStream_t s, float* dst, const float* src, int64_t n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_add_into, dim3(nblocks(n)), dim3(TPB), 0, s, dst, src, n);
}

void add_bias_rows(hipStream_t s, float* Y, const float* b, int64_t rows, int cols) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(k_add_bias, dim3(nblocks(rows * cols)), dim3(TPB), 0, s, Y, b, rows * cols, cols);
  CHECK_LAUNCH();
}

void col_sum_add(hipStream_t s, const float* A, int64_t rows, int cols, float* out, int64_t ld, float* out2) {
  if (rows <= 0) return;
  if (ld <= 0) ld = cols;
  const int rpb = 512;
  dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((cols + 63) / 64));
  hipLaunchKernelGGL(k_colsum, grid, dim3(256), 0, s, A, rows, cols, ld, out, rpb, out2);
  CHECK_LAUNCH();
}

void pool_sigmoid(hipStream_t s, const float* S, int B, int P, int C, int reducer, int K, float* pooled, float* probs, int cid, float* sel, float* sel_host) {
  if (B <= 0) return;
  if (!pooled) hipLaunchKernelGGL(k_pool_sel, dim3(nblocks((int64_t)B)), dim3(TPB), 0, s, S, B, P, C, reducer, K, cid, sel, sel_host);
  else hipLaunchKernelGGL(k_pool, dim3(nblocks((int64_t)B * C)), dim3(TPB), 0, s, S, B, P, C, reducer, K, pooled, probs, cid, sel, sel_host);
  CHECK_LAUNCH();
}

void loss_stage(hipStream_t s, const float* S, const float* labels, const float* hT, int B, int P, int C, int H, int cid, int reducer, int K,
                int literal, float invB, float* pooled, float* probs, float* sel, float* dS, const int32_t* slot_of, float* gW_row, float* gb_c,
                float* partial, const TransposeJob* tj, float* partial_host, const PoolJob* pj) {
  if (B <= 0) return;
  TransposeJob t;
  memset(&t, 0, sizeof(t));
  if (tj) t = *tj;
  PoolJob pjob;
  memset(&pjob, 0, sizeof(pjob));
  if (pj) pjob = *pj;
  const int nlb = (B + LOSS_PPW - 1) / LOSS_PPW;
  const int npb = (pjob.B + 255) / 256;
  hipLaunchKernelGGL(k_loss_stage, dim3((unsigned)(nlb + 64 * t.n + npb)), dim3(256), 0, s, S, labels, hT, B, P, C, H, cid, reducer, K, literal, invB,
                     pooled, probs, sel, dS, slot_of, gW_row, gb_c, partial, nlb, t, partial_host, pjob);
  CHECK_LAUNCH();
}

int loss_partials(int B) { return (B + LOSS_PPW - 1) / LOSS_PPW; }

void sum_partials(hipStream_t s, const float* partial, int n, float* out, int accumulate) {
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, partial, n, out, accumulate);
  CHECK_LAUNCH();
}



void head_bwd(hipStream_t s, const float* dS, const float* hT, const float* Wout, int64_t N, int H, int cid, float* dH, float* gWout, float* gbout) {
  if (N <= 0) return;
  if (H <= 512 && N >= 4096) {   // the wave-per-row form (below that the 64-row workgroups of the plain kernel fill more of the chip)
    const int rpb = 256;
    const dim3 grid((unsigned)((N + rpb - 1) / rpb));
    const int ng = (H + 63) / 64;
#define KPRN_HB(NG)                                                                                                                        \
    do {                                                                                                                                   \
      if (dH) hipLaunchKernelGGL((k_head_bwd_w<NG, true>), grid, dim3(256), 0, s, dS, hT, Wout, N, H, cid, dH, gWout, gbout, rpb);        \
      else hipLaunchKernelGGL((k_head_bwd_w<NG, false>), grid, dim3(256), 0, s, dS, hT, Wout, N, H, cid, dH, gWout, gbout, rpb);          \
    } while (0)
    if (ng <= 1) KPRN_HB(1); else if (ng == 2) KPRN_HB(2); else if (ng == 3) KPRN_HB(3); else if (ng == 4) KPRN_HB(4);
    else if (ng <= 6) KPRN_HB(6); else KPRN_HB(8);
#undef KPRN_HB
    CHECK_LAUNCH();
    return;
  }
  const int rpb = 64;
  hipLaunchKernelGGL(k_head_bwd, dim3((unsigned)((N + rpb - 1) / rpb)), dim3(H >= 256 ? 256 : (H > 64 ? 128 : 64)), 0, s, dS, hT, Wout, N, H, cid,
                     dH, gWout, gbout, rpb);
  CHECK_LAUNCH();
}

// nn.LookupTable backward for a table of at most 16 rows (the type table: 6 rows; the KKBox relation table: 9) from time-major
// row-major dX [T][N][D]: with so few rows every position of the batch lands on the same handful of accumulators, and atomics (LDS
// or L2) on them serialise.  Here a lane owns one column of the slice and keeps one register per table row; a wave reads 64
// consecutive floats per position and adds them to the row its (wave-uniform) id names; one block-level reduction and a few atomics
// per workgroup at the end.  slots > 1: several id columns feed the same slice (CAddTable over the type slots, FeatureEmbedding.lua:55).
__global__ __launch_bounds__(256) void k_small_table_grad(const int32_t* __restrict__ idx, int64_t N, int T, int F, int idcol, int slots,
                                                           const float* __restrict__ dX, int D, int col0, int dcols, int V, float* __restrict__ gW,
                                                           int pos_per_block) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ngroups = (dcols + 63) >> 6;                      // 64-column groups of the slice
  const int grp = blockIdx.y;                                  // one column group per grid row
  const int col = grp * 64 + lane;
  const bool act = col < dcols;
  const int64_t total = N * T;
  const int64_t p0 = (int64_t)blockIdx.x * pos_per_block;
  const int64_t p1 = (p0 + pos_per_block < total) ? p0 + pos_per_block : total;
  float acc[16];
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  // latency-bound per position (one id, 256 bytes): eight positions in flight per wave
  constexpr int UN = 8;
  for (int64_t pb = p0 + wv; pb < p1; pb += 4 * UN) {         // time-major position p = t N + n
    float x[UN];
    const int32_t* f[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t p = pb + 4 * u;
      const bool live = p < p1;
      const int64_t pc = live ? p : p0;
      const int t = (int)(pc / N);
      const int64_t n = pc - (int64_t)t * N;
      x[u] = (act && live) ? dX[pc * D + col0 + col] : 0.f;
      f[u] = idx + (n * T + t) * F + idcol;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      for (int k = 0; k < slots; ++k) {
        const int id = f[u][k] - 1;                            // (wave-uniform)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] += (id == v) ? x[u] : 0.f;
      }
    }
  }
  (void)ngroups;
#pragma unroll
  for (int v = 0; v < 16; ++v) red[wv][v][lane] = acc[v];
  __syncthreads();
  for (int e = threadIdx.x; e < V * 64; e += 256) {
    const int v = e >> 6, c = e & 63;
    const float sum = (red[0][v][c] + red[1][v][c]) + (red[2][v][c] + red[3][v][c]);
    if (grp * 64 + c < dcols && sum != 0.f) unsafeAtomicAdd(gW + (int64_t)v * dcols + grp * 64 + c, sum);
  }
}

// nn.LookupTable backward for a table that fits LDS (V * dcols floats <= 64 KB: the type / relation tables of every config, up to the 100
// relations x 128 of configs[3]) from time-major dX [T][N][D].  A WAVE owns a position: its lanes are the slice's columns, so the LDS adds of
// one instruction never collide, the id is wave-uniform, the row's slice is read with coalesced loads (eight positions in flight), and the
// only 64-bit division is one per workgroup (positions of a block are consecutive: (t, n) is carried along).  One flush of non-zero
// accumulators per workgroup.  Used for tables of more than 16 rows (configs[3]'s 100 relations) instead of the element-indexed
// k_embed_scatter (four 64-bit divisions per ELEMENT); tiny tables keep the 16-register one-hot kernel, which is faster (no LDS atomics).
// Knock-outs on configs[3] (KPRN_TABLE_GRAD_DBG): reads alone 0.11 ms, + id loads 0.21, + LDS adds 0.55: ds_add_f32 is the cost.
__global__ __launch_bounds__(256) void k_table_grad_lds(const int32_t* __restrict__ idx, int64_t N, int T, int F, int idcol, int slots,
                                                         const float* __restrict__ dX, int D, int col0, int dcols, int V, float* __restrict__ gW,
                                                         int pos_per_block, int dbg) {
  extern __shared__ __attribute__((aligned(16))) float acc[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < V * dcols; i += 256) acc[i] = 0.f;
  __syncthreads();
  const int64_t total = N * T;
  const int64_t p0 = (int64_t)blockIdx.x * pos_per_block;
  const int np = (int)((p0 + pos_per_block < total ? p0 + pos_per_block : total) - p0);
  const int t0 = (int)(p0 / N);
  const int64_t n0 = p0 - (int64_t)t0 * N;
  constexpr int UN = 8;
  for (int qb = wv; qb < np; qb += 4 * UN) {
    float x[UN][2];
    int id[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int q = qb + 4 * u;
      const bool live = q < np;
      int t = t0;
      int64_t n = n0 + (live ? q : 0);
      while (n >= N) { n -= N; ++t; }
      const float* src = dX + (p0 + (live ? q : 0)) * D + col0;
      x[u][0] = (live && lane < dcols) ? src[lane] : 0.f;
      x[u][1] = (live && lane + 64 < dcols) ? src[lane + 64] : 0.f;
      id[u] = live ? ((dbg & 4) ? (int)(q % V) : idx[(n * T + t) * F + idcol] - 1) : -1;
      if (slots > 1 && live) {   // CAddTable over the type slots (FeatureEmbedding.lua:55): the same value goes to every slot's row
        for (int k = 1; k < slots; ++k) {
          const int r = idx[(n * T + t) * F + idcol + k] - 1;
          if (lane < dcols) lds_atomic_add(&acc[r * dcols + lane], x[u][0]);
          if (lane + 64 < dcols) lds_atomic_add(&acc[r * dcols + lane + 64], x[u][1]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (id[u] < 0) continue;
      if (dbg & 1) { if (x[u][0] + x[u][1] == 12345.f) acc[0] = 1.f; continue; }
      if (lane < dcols) lds_atomic_add(&acc[id[u] * dcols + lane], x[u][0]);
      if (lane + 64 < dcols) lds_atomic_add(&acc[id[u] * dcols + lane + 64], x[u][1]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < V * dcols; i += 256) {
    const float v = acc[i];
    if (v != 0.f && !(dbg & 2)) unsafeAtomicAdd(gW + i, v);
  }
}

// The same gradient as a one-hot product on the matrix cores (the form the fused path's passenger job takes, kprn_internal.h
// small_grad_block), for row-major time-major dX and tables of up to 128 rows x 128 columns:
//   gW[v][c] += sum_pos [id(pos) == v] dX[pos][col0 + c]  =  D += A B,  A = one-hot [16 v x 4 pos], B = dX [4 pos x 16 c]
// on v_mfma_f32_16x16x4_f32 (1.0 x is exact, fp32 accumulate).  A workgroup walks pos_per_block consecutive positions, four at a time;
// wave w owns the 16-column blocks w and w + 4 and keeps one accumulator tile per (row tile, column block).  Lane (k, n) loads its B
// element dX[pos + k][16 cb + n] and the id of position pos + k (its A element for row tile rt is [id == 16 rt + n]); a row tile none of
// the four ids falls into is skipped (wave-uniform).  No LDS atomics (0.35 of 0.55 ms on configs[3]), no 16 compare-selects per element.
template <int RT>
__global__ __launch_bounds__(256) void k_table_grad_mfma(const int32_t* __restrict__ idx, int64_t N, int T, int F, int idcol, int slots,
                                                          const float* __restrict__ dX, int D, int col0, int dcols, int V, float* __restrict__ gW,
                                                          int pos_per_block) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kk = lane >> 4, n16 = lane & 15;
  const int ncb = (dcols + 15) >> 4;
  f4 acc[RT][2];
#pragma unroll
  for (int r = 0; r < RT; ++r) { acc[r][0] = f4{0.f, 0.f, 0.f, 0.f}; acc[r][1] = f4{0.f, 0.f, 0.f, 0.f}; }
  const int64_t total = N * T;
  const int64_t p0 = (int64_t)blockIdx.x * pos_per_block;
  const int np = (int)((p0 + pos_per_block < total ? p0 + pos_per_block : total) - p0);
  const int t0 = (int)(p0 / N);
  const int64_t n0 = p0 - (int64_t)t0 * N;
  if (wv >= ncb) return;   // (a slice of fewer than four column blocks: this wave owns none)
  const int c_a = 16 * wv + n16, c_b = 16 * (wv + 4) + n16;          // this lane's columns inside the slice
  const bool has_a = wv < ncb && c_a < dcols, has_b = wv + 4 < ncb && c_b < dcols;
  constexpr int UN = 4;   // k-steps (of four positions) in flight
  for (int q0 = 0; q0 < np; q0 += 4 * UN) {
    float xa[UN], xb[UN];
    int id[UN];
    int64_t ioff[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int q = q0 + 4 * u + kk;
      const bool live = q < np;
      int t = t0;
      int64_t n = n0 + (live ? q : 0);
      while (n >= N) { n -= N; ++t; }
      // (every load unconditional, from a clamped position: under `live ? load : 0` hipcc branches around each load and waits for it inside the
      //  branch -- UN dependent round trips per iteration instead of one: configs[3] 0.267 -> 0.199 ms, dims B 0.156 -> 0.116.  A variant with 16-byte
      //  loads feeding four accumulator tiles per lane and the waves splitting the positions measured 0.73 / 1.15 ms: its extra position groups
      //  multiply the epilogue's atomics on the same few table rows, which is what bounds this kernel on small tables)
      const float* src = dX + (p0 + (live ? q : 0)) * D + col0;
      const float va = src[has_a ? c_a : 0], vb = src[has_b ? c_b : 0];
      ioff[u] = (n * T + t) * F + idcol;
      const int idr = idx[ioff[u]];
      xa[u] = (live && has_a) ? va : 0.f;
      xb[u] = (live && has_b) ? vb : 0.f;
      id[u] = live ? idr - 1 : -1;
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      for (int sl = 0; sl < slots; ++sl) {   // CAddTable over the type slots (FeatureEmbedding.lua:55): the same dx goes to every slot's row
        const int idv = (sl == 0) ? id[u] : ((id[u] >= 0) ? idx[ioff[u] + sl] - 1 : -1);
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          if (RT > 1 && __ballot((idv >> 4) == r) == 0) continue;   // (wave-uniform) nobody's id lies in this row tile
          const float onehot = (idv == 16 * r + n16) ? 1.f : 0.f;
          acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(onehot, xa[u], acc[r][0], 0, 0, 0);
          if (wv + 4 < ncb) acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(onehot, xb[u], acc[r][1], 0, 0, 0);   // (wave-uniform)
        }
      }
    }
  }
  // D: lane (g, n) holds rows v = 16 rt + 4 g + i of column 16 cb + n
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = 16 * r + 4 * kk + i;
      if (v < V) {
        if (has_a && acc[r][0][i] != 0.f) unsafeAtomicAdd(gW + (int64_t)v * dcols + c_a, acc[r][0][i]);
        if (has_b && acc[r][1][i] != 0.f) unsafeAtomicAdd(gW + (int64_t)v * dcols + c_b, acc[r][1][i]);
      }
    }
}

static bool table_grad_mfma(hipStream_t s, const int32_t* idx, int64_t N, int T, int F, int idcol, int slots, const float* dX, int D, int col0, int dcols,
                            int V, float* gW) {
  if (dcols <= 0 || dcols > 128 || V <= 0 || V > 128) return false;
  static const int ppb_env = KPRN_DEV_ENV("KPRN_TABLE_GRAD_PPB") ? atoi(KPRN_DEV_ENV("KPRN_TABLE_GRAD_PPB")) : 0;   // (measurement)
  const int ppb = ppb_env > 0 ? ppb_env : 512;   // (round 4, after the loads went unconditional: 512 beats 256 on configs[3] 0.175 : 0.198 ms, shipped 0.147 : 0.198, D = 192
                                                 //  0.106 : 0.113 -- half the workgroups = half the epilogue atomics on the same table rows; 1 024 loses parallelism: 0.253)
  const int64_t total = N * T;
  const dim3 grid((unsigned)((total + ppb - 1) / ppb));
#define KPRN_TG(RT_) hipLaunchKernelGGL(k_table_grad_mfma<RT_>, grid, dim3(256), 0, s, idx, N, T, F, idcol, slots, dX, D, col0, dcols, V, gW, ppb)
  if (V <= 16) KPRN_TG(1);
  else if (V <= 32) KPRN_TG(2);
  else if (V <= 64) KPRN_TG(4);
  else KPRN_TG(8);
#undef KPRN_TG
  return true;
}

static bool table_grad_lds(hipStream_t s, const int32_t* idx, int64_t N, int T, int F, int idcol, int slots, const float* dX, int D, int col0, int dcols,
                           int V, float* gW) {
  const size_t bytes = (size_t)V * dcols * sizeof(float);
  if (dcols <= 0 || dcols > 128 || bytes > 64 * 1024) return false;
  static PerDeviceOnce attr_set;
  if (bytes > 48 * 1024 && attr_set.need()) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_table_grad_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  }
  // positions per workgroup: enough that the flush (V * dcols atomics) stays small beside the reads, few enough to fill the chip
  const int64_t total = N * T;
  int ppb = 512;   // (128 .. 512 measure alike; 1 024 and up lose parallelism)
  while (ppb < 8192 && (int64_t)V * dcols * 8 > (int64_t)ppb * dcols) ppb *= 2;
  static const int ppb_env = KPRN_DEV_ENV("KPRN_TABLE_GRAD_PPB") ? atoi(KPRN_DEV_ENV("KPRN_TABLE_GRAD_PPB")) : 0;   // (measurement)
  if (ppb_env > 0) ppb = ppb_env;
  static const int dbg_env = KPRN_DEV_ENV("KPRN_TABLE_GRAD_DBG") ? atoi(KPRN_DEV_ENV("KPRN_TABLE_GRAD_DBG")) : 0;   // (knock-outs: 1 no LDS adds, 2 no flush, 4 no id loads)
  hipLaunchKernelGGL(k_table_grad_lds, dim3((unsigned)((total + ppb - 1) / ppb)), dim3(256), bytes, s, idx, N, T, F, idcol, slots, dX, D, col0, dcols, V, gW, ppb, dbg_env);
  return true;
}

// ---- small tables (kprn_internal.h has the identity) -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_onehot_cols(const int32_t* __restrict__ idx, int64_t NT, int64_t N, int T, int F, int Vr, int Vt, float* __restrict__ X, int64_t ldx,
                                                     int col0, int ns) {
  const int nq = ns >> 2;   // 16-byte pieces per position
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= NT * nq) return;
  const int64_t pos = g / nq;   // time-major position t N + n
  const int q = (int)(g - pos * nq);
  const int64_t t = pos / N, n = pos - t * N;
  const int32_t* id = idx + (n * T + t) * F;
  const int r = id[F - 1] - 1, y = Vr + id[F - 3] - 1;
  float4 v;
  v.x = (4 * q == r || 4 * q == y) ? 1.f : 0.f;
  v.y = (4 * q + 1 == r || 4 * q + 1 == y) ? 1.f : 0.f;
  v.z = (4 * q + 2 == r || 4 * q + 2 == y) ? 1.f : 0.f;
  v.w = (4 * q + 3 == r || 4 * q + 3 == y) ? 1.f : 0.f;
  float* dst = X + pos * ldx + col0 + 4 * q;
  dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;   // (col0 need not be 16-byte aligned: dt - ns for any dt)
}
void onehot_cols(hipStream_t s, const int32_t* idx, int64_t N, int T, int F, int Vr, int Vt, float* X, int64_t ldx, int col0, int ns) {
  const int64_t work = N * T * (ns >> 2);
  if (work <= 0) return;
  hipLaunchKernelGGL(k_onehot_cols, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, idx, N * T, N, T, F, Vr, Vt, X, ldx, col0, ns);
  HIP_TRY(hipGetLastError());
}
// workgroups [0, GH): one gate row each; behind them one (table row, K slice) each (see lstm_bf16.hip k_small_tables_finish: the same kernel on fp32 weights)
__global__ __launch_bounds__(256) void k_small_tables_finish_f32(const float* __restrict__ Ct, int ns, int GH, int Din, int dt, int de, int dr, int Vt, int Vr,
                                                                 const float* __restrict__ Wt, const float* __restrict__ Wr, const float* __restrict__ Wi,
                                                                 float* __restrict__ gWi, float* __restrict__ gWt, float* __restrict__ gWr) {
  __shared__ float g[128];
  const int tid = threadIdx.x, NZ = ns + de;
  if ((int)blockIdx.x < GH) {
    const int k = blockIdx.x;
    const float* row = Ct + (int64_t)k * NZ;
    if (tid < ns) g[tid] = row[tid];
    __syncthreads();
    for (int j = tid; j < Din; j += 256) {
      float v = 0.f;
      if (j < dt) {
        for (int y = 0; y < Vt; ++y) v += g[Vr + y] * Wt[y * dt + j];
      } else if (j < dt + de) {
        v = row[ns + j - dt];
      } else {
        const int jj = j - dt - de;
        for (int r = 0; r < Vr; ++r) v += g[r] * Wr[r * dr + jj];
      }
      gWi[(int64_t)k * Din + j] += v;
    }
    return;
  }
  // one workgroup per table row, no atomics: two interleaved K sub-sums per column, added in a fixed order (the optimiser tests compare runs BITWISE)
  __shared__ float red[256];
  const int r = (int)blockIdx.x - GH;
  const bool rel = r < Vr;
  const int w = rel ? dr : dt, col0 = rel ? dt + de : 0;
  const int j = tid & 127, sb = tid >> 7;
  for (int j0 = 0; j0 < w; j0 += 128) {
    const int jj = j0 + j, jc = jj < w ? jj : w - 1;
    float acc = 0.f;
    for (int k = sb; k < GH; k += 2) acc += Ct[(int64_t)k * NZ + r] * Wi[(int64_t)k * Din + col0 + jc];
    red[tid] = acc;
    __syncthreads();
    if (sb == 0 && jj < w) (rel ? gWr + (int64_t)r * dr : gWt + (int64_t)(r - Vr) * dt)[jj] += red[tid] + red[tid + 128];
    __syncthreads();
  }
}
void small_tables_finish(hipStream_t s, const float* Ct, int ns, int GH, int Din, int dt, int de, int dr, int Vt, int Vr, const float* Wt, const float* Wr,
                         const float* Wi, float* gWi, float* gWt, float* gWr) {
  hipLaunchKernelGGL(k_small_tables_finish_f32, dim3((unsigned)(GH + Vr + Vt)), dim3(256), 0, s, Ct, ns, GH, Din, dt, de, dr, Vt, Vr, Wt, Wr, Wi, gWi, gWt, gWr);
  HIP_TRY(hipGetLastError());
}

void embed_scatter(hipStream_t s, const int32_t* idx, int64_t N, int T, int F, int nT, const float* dX, int dt, int de, int dr, int Vt, int Vr,
                   float* gWt, float* gWe, float* gWr, bool skip_entity) {
  if (N <= 0) return;
  // tables that fit LDS (every type / relation table of the named configs): one wave per position, LDS accumulators (k_table_grad_lds);
  // KPRN_TABLE_GRAD=old keeps the 16-register one-hot kernel for tiny tables and the element-indexed scatter for the rest
  const int D_ = dt + de + dr;
  static const bool old_path = KPRN_DEV_ENV("KPRN_TABLE_GRAD") && strcmp(KPRN_DEV_ENV("KPRN_TABLE_GRAD"), "old") == 0;
  bool type_small = false, rel_small = false;   // (= handled here)
  // KPRN_TABLE_GRAD: mfma (default) = the one-hot product on the matrix cores for tables up to 128 x 128; lds = LDS accumulators above 16 rows
  // (0.54 ms on configs[3], 0.35 of it ds_add_f32) + the 16-register one-hot kernel below; old = that kernel + the element-indexed scatter
  static const bool lds_path = KPRN_DEV_ENV("KPRN_TABLE_GRAD") && strcmp(KPRN_DEV_ENV("KPRN_TABLE_GRAD"), "lds") == 0;
  if (!old_path && !lds_path) {
    if (dt > 0) type_small = table_grad_mfma(s, idx, N, T, F, F - nT - 2, nT, dX, D_, 0, dt, Vt, gWt);
    if (dr > 0) rel_small = table_grad_mfma(s, idx, N, T, F, F - 1, 1, dX, D_, dt + de, dr, Vr, gWr);
  } else if (lds_path) {
    if (dt > 0 && Vt > 16) type_small = table_grad_lds(s, idx, N, T, F, F - nT - 2, nT, dX, D_, 0, dt, Vt, gWt);
    if (dr > 0 && Vr > 16) rel_small = table_grad_lds(s, idx, N, T, F, F - 1, 1, dX, D_, dt + de, dr, Vr, gWr);
  }
  const int ppb = 512;
  if (!type_small && dt > 0 && Vt <= 16) {
    hipLaunchKernelGGL(k_small_table_grad, dim3((unsigned)((N * T + ppb - 1) / ppb), (unsigned)((dt + 63) / 64)), dim3(256), 0, s, idx, N, T, F, F - nT - 2, nT,
                       dX, D_, 0, dt, Vt, gWt, ppb);
    type_small = true;
  }
  if (!rel_small && dr > 0 && Vr <= 16) {
    hipLaunchKernelGGL(k_small_table_grad, dim3((unsigned)((N * T + ppb - 1) / ppb), (unsigned)((dr + 63) / 64)), dim3(256), 0, s, idx, N, T, F, F - 1, 1, dX,
                       D_, dt + de, dr, Vr, gWr, ppb);
    rel_small = true;
  }
  if ((type_small || dt == 0) && (rel_small || dr == 0) && (skip_entity || de == 0)) { CHECK_LAUNCH(); return; }
  if (type_small) { Vt = 0; }   // (their LDS share is not needed)
  size_t small = (size_t)((int64_t)(type_small ? 0 : Vt) * dt + (int64_t)(rel_small ? 0 : Vr) * dr) * sizeof(float);
  int use_lds = small <= 96 * 1024 ? 1 : 0;
  const int spb = 256;  // steps per block
  int64_t total = N * T;
  if (use_lds && small > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)k_embed_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)small));
  hipLaunchKernelGGL(k_embed_scatter, dim3((unsigned)((total + spb - 1) / spb)), dim3(TPB), use_lds ? small : 0, s, idx, N, T, F, nT, dX, dt, de,
                     dr, type_small ? 0 : Vt, rel_small ? 0 : Vr, gWt, gWe, gWr, use_lds, spb, skip_entity ? 1 : 0, type_small ? 1 : 0, rel_small ? 1 : 0);
  CHECK_LAUNCH();
}

void sumsq(hipStream_t s, const float* x, int64_t n, float* out) {
  if (n <= 0) return;
  unsigned g = nblocks(n);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_sumsq, dim3(g), dim3(TPB), 0, s, x, n, out);
  CHECK_LAUNCH();
}

void sumsq_rows(hipStream_t s, const float* G, const int32_t* rows, const int32_t* count, int d, float* out) {
  hipLaunchKernelGGL(k_sumsq_rows, dim3(1024), dim3(TPB), 0, s, G, rows, count, d, out);
  CHECK_LAUNCH();
}

static AdamDenseJob dense_job(float* x, float* g, float* m, float* v, int64_t n, float step, float b1, float b2, float eps,
                              const float* norm2, float clip, float l2, int consume, int64_t z0, int zn0, int64_t z1, int zn1, float* tab_slot) {
  AdamDenseJob a;
  a.x = x; a.g = g; a.m = m; a.v = v; a.n = n; a.step = step; a.b1 = b1; a.b2 = b2; a.eps = eps; a.norm2 = norm2; a.clip = clip; a.l2 = l2;
  a.reg = (norm2 != nullptr || l2 != 0.f) ? 1 : 0; a.consume = consume; a.z0 = z0; a.zn0 = zn0; a.z1 = z1; a.zn1 = zn1; a.tab_slot = tab_slot;
  return a;
}
void adam_dense(hipStream_t s, float* x, float* g, float* m, float* v, int64_t n, float step, float b1, float b2, float eps,
                const float* norm2, float clip, float l2, int consume, int64_t z0, int zn0, int64_t z1, int zn1, float* tab_slot) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_adam_dense, dim3(nblocks(n)), dim3(TPB), 0, s, dense_job(x, g, m, v, n, step, b1, b2, eps, norm2, clip, l2, consume, z0, zn0, z1, zn1, tab_slot));
  CHECK_LAUNCH();
}

void adagrad_dense(hipStream_t s, float* x, float* g, float* G, int64_t n, float clr, const float* norm2, float clip, float l2, int consume,
                   int64_t z0, int zn0, int64_t z1, int zn1) {
  if (n <= 0) return;
  int reg = (norm2 != nullptr || l2 != 0.f) ? 1 : 0;
  hipLaunchKernelGGL(k_adagrad_dense, dim3(nblocks(n)), dim3(TPB), 0, s, x, g, G, n, clr, norm2, clip, l2, reg, consume, z0, zn0, z1, zn1);
  CHECK_LAUNCH();
}

void adam_rows(hipStream_t s, float* W, float* g, float* m, float* v, int32_t* last, const int32_t* rows, const int32_t* count, int64_t max_rows,
               int d, int32_t t_now, int apply_step, const float* step_tab, float b1, float b2, float eps, int64_t pad_row) {
  if (max_rows <= 0) return;
  const bool al = !(((uintptr_t)W | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15);
  AdamDenseJob none;
  memset(&none, 0, sizeof(none));
  if (al && d == 32) { const int nb = (int)nblocks(max_rows * 8); hipLaunchKernelGGL(k_adam_rows_v<8>, dim3(nb), dim3(TPB), 0, s, W, g, m, v, last, rows, count, t_now, apply_step, step_tab, b1, b2, eps, pad_row, nb, -1.f, none); }
  else if (al && d == 64) { const int nb = (int)nblocks(max_rows * 16); hipLaunchKernelGGL(k_adam_rows_v<16>, dim3(nb), dim3(TPB), 0, s, W, g, m, v, last, rows, count, t_now, apply_step, step_tab, b1, b2, eps, pad_row, nb, -1.f, none); }
  else if (al && d == 128) { const int nb = (int)nblocks(max_rows * 32); hipLaunchKernelGGL(k_adam_rows_v<32>, dim3(nb), dim3(TPB), 0, s, W, g, m, v, last, rows, count, t_now, apply_step, step_tab, b1, b2, eps, pad_row, nb, -1.f, none); }
  else hipLaunchKernelGGL(k_adam_rows, dim3(nblocks(max_rows * 64)), dim3(TPB), 0, s, W, g, m, v, last, rows, count, d, t_now, apply_step, step_tab, b1,
                          b2, eps, pad_row);
  CHECK_LAUNCH();
}

// the optimiser step of the lazy-exact row update AND of the dense arena as ONE launch (rows of 32 / 64 / 128 floats, 16-byte aligned tables); false: the
// caller launches the two separately.  Same per-element arithmetic as the separate kernels (adam_elem): bit-identical results.
bool adam_step_merged(hipStream_t s, float* W, float* g, float* m, float* v, int32_t* last, const int32_t* rows, const int32_t* count, int64_t max_rows, int d,
                      int32_t t_now, const float* step_tab, int64_t pad_row, float* dx, float* dg, float* dm, float* dv, int64_t dn, float step, float b1,
                      float b2, float eps, const float* norm2, float clip, float l2, int64_t z0, int zn0, int64_t z1, int zn1, float* tab_slot) {
  const bool al = !(((uintptr_t)W | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15);
  if (max_rows <= 0 || dn <= 0 || !al || !(d == 32 || d == 64 || d == 128)) return false;
  const AdamDenseJob dj = dense_job(dx, dg, dm, dv, dn, step, b1, b2, eps, norm2, clip, l2, /*consume=*/1, z0, zn0, z1, zn1, tab_slot);
  const int nd = (int)nblocks(dn);
  if (d == 32) { const int nb = (int)nblocks(max_rows * 8); hipLaunchKernelGGL(k_adam_rows_v<8>, dim3(nb + nd), dim3(TPB), 0, s, W, g, m, v, last, rows, count, t_now, 1, step_tab, b1, b2, eps, pad_row, nb, step, dj); }
  else if (d == 64) { const int nb = (int)nblocks(max_rows * 16); hipLaunchKernelGGL(k_adam_rows_v<16>, dim3(nb + nd), dim3(TPB), 0, s, W, g, m, v, last, rows, count, t_now, 1, step_tab, b1, b2, eps, pad_row, nb, step, dj); }
  else { const int nb = (int)nblocks(max_rows * 32); hipLaunchKernelGGL(k_adam_rows_v<32>, dim3(nb + nd), dim3(TPB), 0, s, W, g, m, v, last, rows, count, t_now, 1, step_tab, b1, b2, eps, pad_row, nb, step, dj); }
  CHECK_LAUNCH();
  return true;
}

void adam_flush_all(hipStream_t s, float* W, float* m, float* v, int32_t* last, int64_t V, int d, int32_t t_now, const float* step_tab, float b1,
                    float b2, float eps, int64_t pad_row) {
  hipLaunchKernelGGL(k_adam_flush_all, dim3(nblocks(V * 64)), dim3(TPB), 0, s, W, m, v, last, V, d, t_now, step_tab, b1, b2, eps, pad_row);
  CHECK_LAUNCH();
}

void adagrad_rows(hipStream_t s, float* W, float* g, float* G, const int32_t* rows, const int32_t* count, int64_t max_rows, int d, float clr,
                  int64_t pad_row) {
  if (max_rows <= 0) return;
  hipLaunchKernelGGL(k_adagrad_rows, dim3(nblocks(max_rows * 64)), dim3(TPB), 0, s, W, g, G, rows, count, d, clr, pad_row);
  CHECK_LAUNCH();
}

void zero_pad3(hipStream_t s, float* a, int na, float* b, int nb, float* c, int nc) {
  const int n = na > nb ? (na > nc ? na : nc) : (nb > nc ? nb : nc);
  hipLaunchKernelGGL(k_zero_pad3, dim3(nblocks(n)), dim3(TPB), 0, s, a, na, b, nb, c, nc);
  CHECK_LAUNCH();
}

void zero_rows(hipStream_t s, float* W, int64_t row, int d) {
  hipLaunchKernelGGL(k_zero_row, dim3(nblocks(d)), dim3(TPB), 0, s, W, row, d);
  CHECK_LAUNCH();
}

void pack_rows(hipStream_t s, float* G, const int32_t* rows, const int32_t* count, int64_t max_rows, int d, int32_t* ids_out, float* rows_out,
               int32_t* count_out, const float* tail_src, int64_t n_tail, float* tail_dst) {
  int64_t waves = max_rows > 0 ? max_rows : 1;
  const bool al = !(((uintptr_t)G | (uintptr_t)rows_out | (uintptr_t)tail_dst) & 15);
  const int tail_blocks = n_tail > 0 ? (int)std::min<int64_t>((n_tail + 255) / 256, 256) : 0;
#define KPRN_PACK_V(LG)                                                                                                                      \
  {                                                                                                                                          \
    const int rb = (int)nblocks(waves * LG);                                                                                                 \
    hipLaunchKernelGGL(k_pack_rows_v<LG>, dim3((unsigned)(rb + tail_blocks)), dim3(TPB), 0, s, G, rows, count, ids_out, rows_out, count_out, rb, \
                       tail_src, n_tail, tail_dst);                                                                                          \
  }
  if (al && d == 32) KPRN_PACK_V(8)
  else if (al && d == 64) KPRN_PACK_V(16)
  else if (al && d == 128) KPRN_PACK_V(32)
  else {
    hipLaunchKernelGGL(k_pack_rows, dim3(nblocks(waves * 64)), dim3(TPB), 0, s, G, rows, count, d, ids_out, rows_out, count_out);
    if (n_tail > 0) HIP_TRY(hipMemcpyAsync(tail_dst, tail_src, (size_t)n_tail * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
#undef KPRN_PACK_V
  CHECK_LAUNCH();
}

// union of all ranks' packed rows + the lazy-exact Adam step on it (k_union_adam); false = shape not covered (the caller materialises
// the union with bidx::merge_rows and takes the ordinary row update)
bool union_adam(hipStream_t s, const void* all, int world, int cap, int64_t stride, int d, float* W, float* m, float* v, int32_t* last, int32_t t_now,
                const float* step_tab, float b1, float b2, float eps, int64_t pad_row) {
  const bool al = !(((uintptr_t)W | (uintptr_t)m | (uintptr_t)v | (uintptr_t)all) & 15) && (stride & 3) == 0 && (cap & 3) == 0;
  const int lg = d >> 2;
  if (!al || !(d == 32 || d == 64 || d == 128) || world > lg || world <= 0 || cap <= 0) return false;
  const int64_t n = (int64_t)world * cap;
  if (d == 32) hipLaunchKernelGGL(k_union_adam<8>, dim3(nblocks(n * 8)), dim3(TPB), 0, s, (const int32_t*)all, world, cap, stride, W, m, v, last, t_now, step_tab, b1, b2, eps, pad_row);
  else if (d == 64) hipLaunchKernelGGL(k_union_adam<16>, dim3(nblocks(n * 16)), dim3(TPB), 0, s, (const int32_t*)all, world, cap, stride, W, m, v, last, t_now, step_tab, b1, b2, eps, pad_row);
  else hipLaunchKernelGGL(k_union_adam<32>, dim3(nblocks(n * 32)), dim3(TPB), 0, s, (const int32_t*)all, world, cap, stride, W, m, v, last, t_now, step_tab, b1, b2, eps, pad_row);
  CHECK_LAUNCH();
  return true;
}


void fill_uniform(hipStream_t s, float* x, int64_t n, float a, uint64_t seed, uint64_t offset) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_fill_uniform, dim3(nblocks(n)), dim3(TPB), 0, s, x, n, a, seed, offset);
  CHECK_LAUNCH();
}

void clear_rows(hipStream_t s, float* G, const int32_t* rows, const int32_t* count, int64_t max_rows, int d) {
  if (max_rows <= 0) return;
  hipLaunchKernelGGL(k_clear_rows, dim3(nblocks(max_rows * 64)), dim3(TPB), 0, s, G, rows, count, d);
  CHECK_LAUNCH();
}


void fill_i32(hipStream_t s, int32_t* x, int64_t n, int32_t v) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_fill_i32, dim3(nblocks(n)), dim3(TPB), 0, s, x, n, v);
  CHECK_LAUNCH();
}

}  // namespace kk
