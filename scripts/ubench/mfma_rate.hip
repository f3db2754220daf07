// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 on gfx950 with one wave per SIMD (the fused LSTM
// kernels' regime).  Prints cycles per MFMA (s_memtime ticks) for a few accumulator / operand patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 68];
  for (int i = threadIdx.x; i < 64 * 68; i += 256) lds[i] = 0.001f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float b[16];
  for (int i = 0; i < 16; ++i) b[i] = 0.5f + i + lane;
  f32x4 a4 = *(const f32x4*)(lds + (lane & 15) * 68 + (lane >> 4) * 4);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    f32x4 an = a4;
    if (MODE == 1) an = *(const f32x4*)(lds + ((lane + it) & 15) * 68 + (lane >> 4) * 4 + (it & 3) * 16);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[(jj * 4 + q) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], b[(q * 4 + jj) & 15], acc[(jj * 4 + q) % NACC], 0, 0, 0);
    if (MODE == 3) {  // ping-pong: second group consumes `an`, prefetches a4 again (no register copies)
      a4 = *(const f32x4*)(lds + ((lane + it + 1) & 15) * 68 + (lane >> 4) * 4 + (it & 3) * 16);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[(jj * 4 + q) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[jj], b[(q * 4 + jj) & 15], acc[(jj * 4 + q) % NACC], 0, 0, 0);
    } else if (MODE == 4) {  // VALU-produced next fragment instead of an LDS read
      a4 = an * 1.0001f;
    } else if (MODE == 5) {  // 4 x ds_read_b32 instead of one b128
      const float* q4 = lds + ((lane + it) & 15) * 68 + (lane >> 4) * 4 + (it & 3) * 16;
      a4 = f32x4{q4[0 + 17 * 0], q4[1 + 68], q4[2 + 136], q4[3 + 204]};
    } else if (MODE == 6) {  // conflict-free b128: consecutive lanes read consecutive 16 B
      a4 = *(const f32x4*)(lds + lane * 4 + (it & 15) * 256);
    } else {
      a4 = an;
    }
    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = f32x4{0, 0, 0, 0};
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the fused forward's half_unit pattern: 8 k-groups per iteration, A fragment of the next group always in flight,
// immediate LDS offsets (no VALU in the loop), B operands register-stationary (128 VGPRs)
template <int VAR>
__global__ __launch_bounds__(256, 1) void k2(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 68 * 2];
  for (int i = threadIdx.x; i < 64 * 68 * 2; i += 256) lds[i] = 0.001f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f32x4 w[8][4];
  for (int i = 0; i < 8; ++i) for (int q = 0; q < 4; ++q) w[i][q] = f32x4{0.5f + i + lane, 1.f + q, 2.f + i * q, 3.f};
  const float* base = lds + (lane & 15) * 68 + (lane >> 4) * 4;
  f32x4 apre = *(const f32x4*)(base);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const f32x4 a4 = apre;
      if (VAR != 9) apre = *(const f32x4*)(base + ((g + 1) & 7) * 16 + (g >= 4 ? 64 * 68 : 0));
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], w[g][q][jj], acc[q], 0, 0, 0);
      if (VAR == 1) __builtin_amdgcn_sched_barrier(0);
      if (VAR == 2) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 16, 0); __builtin_amdgcn_sched_barrier(0); }
      if (VAR == 3) { __builtin_amdgcn_sched_group_barrier(0x008, 8, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 8, 0); __builtin_amdgcn_sched_barrier(0); }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = apre;
  for (int i = 0; i < 4; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// inline-asm MFMA: B operand pinned to AGPRs ("a"), accumulators in architectural VGPRs ("v"); 256 weight
// registers stationary like the fused forward at L = 2.  VAR 1 adds an independent cell-like VALU stream
// (2 transcendental-heavy ops per MFMA), VAR 2 stages every B operand through v_accvgpr_read first.
#define MFMA_VA(acc, a_, b_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a_), "a"(b_))
#define MFMA_VV(acc, a_, b_) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a_), "v"(b_))
template <int VAR>
__global__ __launch_bounds__(256, 1) void k3(float* out, unsigned long long* cyc, int iters, const float* wsrc) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 68 * 2];
  for (int i = threadIdx.x; i < 64 * 68 * 2; i += 256) lds[i] = 0.001f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f32x4 w[16][4];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { w[i][q] = *(const f32x4*)(wsrc + ((i * 4 + q) * 64 + lane) * 4); asm volatile("" : "+a"(w[i][q])); }
  const float* base = lds + (lane & 15) * 68 + (lane >> 4) * 4;
  f32x4 apre = *(const f32x4*)(base);
  float x0 = lane * 0.01f, x1 = 0.3f, x2 = 0.7f, x3 = 1.1f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const f32x4 a4 = apre;
      apre = *(const f32x4*)(base + ((g + 1) & 7) * 16 + ((g & 8) ? 64 * 68 : 0));
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (VAR == 2) { float t = w[g][q][jj]; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(w[g][q][jj])); MFMA_VV(acc[q], a4[jj], t); }
          else MFMA_VA(acc[q], a4[jj], w[g][q][jj]);
          if (VAR == 1) {
            if (((jj * 4 + q) & 3) == 0) x0 = __builtin_amdgcn_rcpf(1.0f + __expf(-x0));
            if (((jj * 4 + q) & 3) == 1) x1 = __builtin_amdgcn_rcpf(1.0f + __expf(-x1));
            if (((jj * 4 + q) & 3) == 2) x2 = x2 * x0 + x1;
            if (((jj * 4 + q) & 3) == 3) x3 = __builtin_amdgcn_rcpf(1.0f + __expf(-x3 * x2));
          }
        }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15");
  f32x4 s = apre;
  for (int i = 0; i < 4; ++i) s += acc[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// co-issue calibration: asm MFMA stream (B in AGPRs) with VAR-selected INDEPENDENT VALU work behind every MFMA
// (8 rotating registers: an op depends on the result produced 8 MFMAs earlier), order pinned by sched_barrier.
template <int VAR>
__global__ __launch_bounds__(256, 1) void k4(float* out, unsigned long long* cyc, int iters, const float* wsrc) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 68 * 2];
  for (int i = threadIdx.x; i < 64 * 68 * 2; i += 256) lds[i] = 0.001f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f32x4 w[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { w[i][q] = *(const f32x4*)(wsrc + ((i * 4 + q) * 64 + lane) * 4); asm volatile("" : "+a"(w[i][q])); }
  const float* base = lds + (lane & 15) * 68 + (lane >> 4) * 4;
  f32x4 a4 = *(const f32x4*)(base);
  float x[8], y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = lane * 0.01f + i; y[i] = 0.5f * i; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        MFMA_VA(acc[k & 3], a4[k >> 2], w[g][k & 3][k >> 2]);
        const int r = k & 7;
        if (VAR == 1) x[r] = x[r] * 1.0001f + 0.5f;
        if (VAR == 2) x[r] = __builtin_amdgcn_exp2f(x[r]);
        if (VAR == 3) { x[r] = __builtin_amdgcn_exp2f(x[r]); y[r] = y[r] * 1.0001f + 0.5f; }
        if (VAR == 4) { x[r] = x[r] * 1.0001f + 0.5f; y[r] = y[r] * 1.0001f + 0.5f; }
        if (VAR == 5) { x[r] = __builtin_amdgcn_rcpf(x[r]); y[r] = y[r] * 1.0001f + 0.5f; y[(r + 4) & 7] = y[(r + 4) & 7] * 0.999f + 0.25f; }
        if (VAR == 6) { x[r] = x[r] * 1.0001f + 0.5f; y[r] = y[r] * 1.0001f + 0.5f; x[(r + 4) & 7] *= 0.999f; y[(r + 4) & 7] *= 1.001f; }
        if (VAR == 7) lds[64 * 68 + threadIdx.x + k * 256] = x[r];
        if (VAR == 8) { x[r] = __builtin_amdgcn_exp2f(x[r]); y[r] = __builtin_amdgcn_rcpf(y[r]); }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15");
  f32x4 s = a4;
  for (int i = 0; i < 4; ++i) s += acc[i];
  float e = 0;
  for (int i = 0; i < 8; ++i) e += x[i] + y[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + e;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// two waves per SIMD: waves 0-3 run the VALU-heavy stream (VARA), waves 4-7 the stream VARB; reports the
// cycles until the LAST wave of the workgroup is done, per MFMA issued on one SIMD (2 * 64 * iters)
template <int VARA, int VARB>
__global__ __launch_bounds__(512, 1) void k5(float* out, unsigned long long* cyc, int iters, const float* wsrc) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 68 * 2];
  __shared__ unsigned long long tend[8];
  for (int i = threadIdx.x; i < 64 * 68 * 2; i += 512) lds[i] = 0.001f * i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  f32x4 w[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) { w[i][q] = *(const f32x4*)(wsrc + ((i * 4 + q) * 64 + lane) * 4); asm volatile("" : "+a"(w[i][q])); }
  const float* base = lds + (lane & 15) * 68 + (lane >> 4) * 4;
  f32x4 a4 = *(const f32x4*)(base);
  float x[8], y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = lane * 0.01f + i; y[i] = 0.5f * i; }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  auto body = [&](auto var_tag) {
    constexpr int VAR = decltype(var_tag)::value;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          MFMA_VA(acc[k & 3], a4[k >> 2], w[g][k & 3][k >> 2]);
          const int r = k & 7;
          if (VAR == 8) { x[r] = __builtin_amdgcn_exp2f(x[r]); y[r] = __builtin_amdgcn_rcpf(y[r]); }
          if (VAR == 6) { x[r] = x[r] * 1.0001f + 0.5f; y[r] = y[r] * 1.0001f + 0.5f; x[(r + 4) & 7] *= 0.999f; y[(r + 4) & 7] *= 1.001f; }
          if (VAR == 9 && k >= 8) { x[r] = __builtin_amdgcn_exp2f(x[r]); y[r] = __builtin_amdgcn_rcpf(y[r]); x[(r + 4) & 7] *= 0.999f; y[(r + 4) & 7] *= 1.001f; }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  };
  if (wv < 4) body(std::integral_constant<int, VARA>{}); else body(std::integral_constant<int, VARB>{});
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15");
  if (lane == 0) tend[wv] = t1 - t0;
  __syncthreads();
  f32x4 s = a4;
  for (int i = 0; i < 4; ++i) s += acc[i];
  float e = 0;
  for (int i = 0; i < 8; ++i) e += x[i] + y[i];
  out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + e;
  if (threadIdx.x == 0) { unsigned long long m = 0; for (int i = 0; i < 8; ++i) m = tend[i] > m ? tend[i] : m; cyc[blockIdx.x] = m; }
}

template <int VARA, int VARB>
void run5(const char* name, int grid) {
  float* out; unsigned long long* cyc; float* wsrc;
  hipMalloc(&out, grid * 512 * 4); hipMalloc(&cyc, grid * 8); hipMalloc(&wsrc, 64 * 64 * 4 * 4); hipMemset(wsrc, 0, 64 * 64 * 4 * 4);
  const int iters = 1000;
  hipLaunchKernelGGL((k5<VARA, VARB>), dim3(grid), dim3(512), 0, 0, out, cyc, iters, wsrc);
  hipDeviceSynchronize();
  unsigned long long h[1024]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= grid;
  printf("%-36s grid %4d: %.2f ticks per MFMA per SIMD (2 waves)\n", name, grid, avg / (128.0 * iters));
  hipFree(out); hipFree(cyc); hipFree(wsrc);
}

template <int VAR>
void run4(const char* name, int grid) {
  float* out; unsigned long long* cyc; float* wsrc;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8); hipMalloc(&wsrc, 64 * 64 * 4 * 4); hipMemset(wsrc, 0, 64 * 64 * 4 * 4);
  const int iters = 1000;
  hipLaunchKernelGGL((k4<VAR>), dim3(grid), dim3(256), 0, 0, out, cyc, iters, wsrc);
  hipDeviceSynchronize();
  unsigned long long h[1024]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= grid;
  printf("%-36s grid %4d: %.2f ticks/MFMA\n", name, grid, avg / (64.0 * iters));
  hipFree(out); hipFree(cyc); hipFree(wsrc);
}

template <int VAR>
void run3(const char* name, int grid) {
  float* out; unsigned long long* cyc; float* wsrc;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8); hipMalloc(&wsrc, 64 * 64 * 4 * 4); hipMemset(wsrc, 0, 64 * 64 * 4 * 4);
  const int iters = 300;
  hipLaunchKernelGGL((k3<VAR>), dim3(grid), dim3(256), 0, 0, out, cyc, iters, wsrc);
  hipDeviceSynchronize();
  unsigned long long h[1024]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= grid;
  printf("%-36s grid %4d: %.2f ticks/MFMA\n", name, grid, avg / (256.0 * iters));
  hipFree(out); hipFree(cyc); hipFree(wsrc);
}

template <int VAR>
void run2(const char* name, int grid) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
  const int iters = 500;
  hipLaunchKernelGGL((k2<VAR>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  unsigned long long h[1024]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= grid;
  printf("%-36s grid %4d: %.2f ticks/MFMA\n", name, grid, avg / (128.0 * iters));
  hipFree(out); hipFree(cyc);
}

template <int NACC, int MODE>
void run(const char* name, int grid) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1024]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= grid;
  const double n = 16.0 * iters;
  printf("%-28s grid %4d: %.2f memtime-ticks/MFMA, kernel %.3f ms -> %.1f TFLOP/s, ticks/us %.1f\n", name, grid, avg / n, ms,
         grid * 4 * n * 2048 / (ms * 1e-3) / 1e12, avg / (ms * 1e3));
  hipFree(out); hipFree(cyc);
}

int main() {
  run5<0, 0>("k5 2 waves: mfma | mfma", 256);
  run5<8, 0>("k5 2 waves: mfma+exp+rcp | mfma", 256);
  run5<8, 8>("k5 2 waves: both mfma+exp+rcp", 256);
  run5<6, 6>("k5 2 waves: both mfma+4 valu", 256);
  run5<9, 9>("k5 2 waves: both, bursty valu", 256);
  run4<0>("k4 mfma only", 256);
  run4<1>("k4 + 1 fma", 256);
  run4<2>("k4 + 1 exp", 256);
  run4<3>("k4 + 1 exp + 1 fma", 256);
  run4<4>("k4 + 2 fma", 256);
  run4<5>("k4 + 1 rcp + 2 fma", 256);
  run4<6>("k4 + 2 fma + 2 mul", 256);
  run4<7>("k4 + 1 ds_write_b32", 256);
  run4<8>("k4 + exp + rcp", 256);
  run3<0>("k3 asm mfma, B in AGPR", 256);
  run3<1>("k3 + interleaved cell-like VALU", 256);
  run3<2>("k3 B staged via accvgpr_read", 256);
  run2<0>("k2 free schedule", 256);
  run2<1>("k2 sched_barrier per group", 256);
  run2<2>("k2 read first then 16 mfma", 256);
  run2<3>("k2 8 mfma, read, 8 mfma", 256);
  run2<9>("k2 no LDS reads", 256);
  run<4, 0>("4 acc, regs only", 256);
  run<16, 0>("16 acc, regs only", 256);
  run<4, 1>("4 acc + ds_read_b128/16", 256);
  run<4, 2>("4 acc + sched_barrier", 256);
  run<4, 3>("4 acc + b128 ping-pong x2", 256);
  run<4, 4>("4 acc + VALU a4 update", 256);
  run<4, 5>("4 acc + 4 ds_read_b32", 256);
  run<4, 6>("4 acc + b128 linear", 256);
  run<4, 0>("4 acc, regs only, 1 WG", 1);
  run<4, 0>("4 acc, regs only, 32 WG", 32);
  return 0;
}
