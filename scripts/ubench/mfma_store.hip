// Micro-benchmark: what a global store costs a wave that is issuing fp32 MFMAs back to back (one wave per SIMD, all CUs: the fused training forward's
// regime).  Prints s_memtime ticks per 16-MFMA group for: no store, one dwordx4 / two dwordx2 / four dword stores per group, a non-temporal dwordx4, a
// dwordx4 LOAD (consumed 8 groups later), four stores back to back per 64 MFMAs, and the stores of the four waves skewed against each other.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_store scripts/ubench/mfma_store.hip && /tmp/mfma_store
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA16()                                                                                                   \
  _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) _Pragma("unroll") for (int q = 0; q < 4; ++q)                    \
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[jj], b[q * 4 + jj], acc[q], 0, 0, 0);

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, float* big, size_t wg_floats, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float b[16];
  for (int i = 0; i < 16; ++i) b[i] = 0.5f + i + lane;
  f32x4 a4 = f32x4{1.f + lane, 2.f, 3.f, 4.f};
  f32x4 sv = f32x4{(float)lane, 1.f, 2.f, 3.f};
  f32x4 ld = f32x4{0, 0, 0, 0};
  f32x4 ring[8];
  for (int i = 0; i < 8; ++i) ring[i] = f32x4{0, 0, 0, 0};
  float* mine = big + (size_t)blockIdx.x * wg_floats;
  const size_t mask = wg_floats / 1024 - 1;   // 4 KiB slots per workgroup (power of two)
  if (MODE == 8) for (int i = 0; i < wave; ++i) __builtin_amdgcn_s_sleep(1);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
  for (int it = 0; it < iters; ++it) {
    float* p = mine + ((size_t)it & mask) * 1024 + threadIdx.x * 4;
    if (MODE == 6) {  // 4 stores back to back, then 64 MFMAs
      if ((it & 3) == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *(f32x4*)(mine + ((size_t)(it + u) & mask) * 1024 + threadIdx.x * 4) = sv;
      }
    } else if (MODE == 1 || MODE == 8) *(f32x4*)p = sv;
    else if (MODE == 2) { *(f32x2*)(p) = f32x2{sv[0], sv[1]}; *(f32x2*)(mine + ((size_t)it & mask) * 1024 + 512 + threadIdx.x * 2) = f32x2{sv[2], sv[3]}; }
    else if (MODE == 3) { for (int u = 0; u < 4; ++u) mine[((size_t)it & mask) * 1024 + u * 256 + threadIdx.x] = sv[u]; }
    else if (MODE == 4) { ld += ring[it & 7]; ring[it & 7] = *(const f32x4*)p; }   // (consumed 8 groups = 4 k cycles later)
    else if (MODE == 9) { ld += ring[it & 7]; ring[it & 7] = *(const f32x4*)(mine + ((size_t)it & mask) * 1024 + (threadIdx.x & 15) * 64 + (threadIdx.x >> 4) * 4); }
    else if (MODE == 5) __builtin_nontemporal_store(sv, (f32x4*)p);
    else if (MODE == 7) {  // the strided shape: consecutive lanes 256 B apart
      *(f32x4*)(mine + ((size_t)it & mask) * 1024 + (threadIdx.x & 15) * 64 + (threadIdx.x >> 4) * 4) = sv;
    }
    __builtin_amdgcn_sched_barrier(0);
    MFMA16()
    __builtin_amdgcn_sched_barrier(0);
    sv[0] += 1.0f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3] + ld;
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, float* big, size_t wg_floats, unsigned long long* cyc, int grid) {
  const int iters = 4096;
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, out, big, wg_floats, cyc, iters);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, out, big, wg_floats, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> c(grid);
  hipMemcpy(c.data(), cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s = 0; for (auto v : c) s += (double)v;
  printf("%-44s %8.1f ticks / 16 MFMAs   %7.3f ms  (%.1f ns per group)\n", name, s / grid / iters, ms, ms * 1e6 / iters);
}

int main() {
  int grid = 256;
  float *out, *big; unsigned long long* cyc;
  (void)hipMalloc(&out, grid * 256 * sizeof(float)); (void)hipMalloc(&big, grid * ((size_t)1 << 22) + (1 << 20)); (void)hipMalloc(&cyc, grid * sizeof(unsigned long long));
  (void)hipMemset(big, 0, grid * ((size_t)1 << 22));
  for (size_t wg_floats : {(size_t)1 << 20, (size_t)1 << 14}) {   // 4 MiB per workgroup (streams to HBM) | 64 KiB (stays in the L2)
    printf("-- %zu KiB per workgroup\n", wg_floats * 4 / 1024);
    run<0>("no store", out, big, wg_floats, cyc, grid);
    run<1>("1 x global_store_dwordx4", out, big, wg_floats, cyc, grid);
    run<2>("2 x dwordx2", out, big, wg_floats, cyc, grid);
    run<3>("4 x dword", out, big, wg_floats, cyc, grid);
    run<5>("1 x dwordx4 non-temporal", out, big, wg_floats, cyc, grid);
    run<4>("1 x global_load_dwordx4", out, big, wg_floats, cyc, grid);
    run<9>("1 x global_load_dwordx4, lanes 256 B apart", out, big, wg_floats, cyc, grid);
    run<6>("4 x dwordx4 back to back per 64 MFMAs", out, big, wg_floats, cyc, grid);
    run<7>("1 x dwordx4, lanes 256 B apart", out, big, wg_floats, cyc, grid);
    run<8>("1 x dwordx4, waves skewed by 64 cycles", out, big, wg_floats, cyc, grid);
  }
  return 0;
}
