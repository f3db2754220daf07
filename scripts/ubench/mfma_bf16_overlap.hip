// Micro-benchmark (gfx950): do VALU instructions overlap with bf16 MFMA the way they do NOT with fp32 MFMA?
// One wave per SIMD (the fused kernels' regime).  Streams of independent MFMAs with optional VALU work behind each:
//   fp32: v_mfma_f32_16x16x4_f32        (2 KFLOP per 32-cycle issue slot)
//   bf16: v_mfma_f32_16x16x32_bf16      (16 KFLOP per issue)
// Prints s_memtime ticks per MFMA.  Feeds DESIGN.md section 7 ("fp32 products on the matrix cores").
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int VAR>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
  float af = 0.5f + lane * 0.001f, bfv = 0.25f + lane * 0.002f;
  bf16x8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(0.5f + 0.01f * i + lane * 0.001f); b8[i] = (__bf16)(0.25f + 0.02f * i); }
  bf16x8 bw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { bw[i] = b8; bw[i][0] = (__bf16)(0.1f * i); if (KIND == 2) asm volatile("" : "+a"(bw[i])); else asm volatile("" : "+v"(bw[i])); }
  float x[8], y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = lane * 0.01f + i; y[i] = 0.5f * i; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) {
      if (KIND == 0) acc[k2 & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bfv, acc[k2 & 3], 0, 0, 0);
      else if (KIND == 1) acc[k2 & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[k2 & 3], 0, 0, 0);
      else if (KIND == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k2 & 3]) : "v"(a8), "a"(bw[k2 & 7]));   // B from the AGPR half, 8 different registers
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[k2 & 3]) : "v"(a8), "v"(bw[k2 & 7]));                   // same, B in VGPRs
      const int r = k2 & 7;
      if (VAR == 1) { x[r] = x[r] * 1.0001f + 0.5f; y[r] = y[r] * 1.0001f + 0.5f; }
      if (VAR == 2) { x[r] = __builtin_amdgcn_exp2f(x[r]); y[r] = __builtin_amdgcn_rcpf(y[r]); }
      if (VAR == 3) { x[r] = x[r] * 1.0001f + 0.5f; y[r] = y[r] * 1.0001f + 0.5f; x[(r + 4) & 7] *= 0.999f; y[(r + 4) & 7] *= 1.001f;
                      x[(r + 2) & 7] += 0.25f; y[(r + 2) & 7] += 0.125f; x[(r + 6) & 7] *= 1.01f; y[(r + 6) & 7] *= 0.99f; }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
  float e = 0;
  for (int i = 0; i < 8; ++i) e += x[i] + y[i];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + e;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int VAR>
void run(const char* name) {
  const int grid = 256, iters = 4000;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, grid * 8);
  hipLaunchKernelGGL((k<KIND, VAR>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, VAR>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < grid; ++i) avg += h[i]; avg /= grid;
  const double n = 16.0 * iters;
  const double flops_per = KIND == 0 ? 2048.0 : 16384.0;  // (KIND 1..3: bf16)
  printf("%-44s %.2f ticks/MFMA, kernel %.3f ms -> %.1f TFLOP/s (ticks/us %.0f)\n", name, avg / n, ms, grid * 4 * n * flops_per / (ms * 1e-3) / 1e12,
         avg / (ms * 1e3));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0>("fp32 16x16x4 alone");
  run<0, 1>("fp32 16x16x4 + 2 fma per MFMA");
  run<0, 2>("fp32 16x16x4 + exp2 + rcp per MFMA");
  run<1, 0>("bf16 16x16x32 alone");
  run<1, 1>("bf16 16x16x32 + 2 fma per MFMA");
  run<1, 2>("bf16 16x16x32 + exp2 + rcp per MFMA");
  run<1, 3>("bf16 16x16x32 + 8 simple VALU per MFMA");
  run<3, 0>("bf16 asm, B in VGPRs (8 regs rotating)");
  run<2, 0>("bf16 asm, B in AGPRs (8 regs rotating)");
  run<2, 2>("bf16 asm, B in AGPRs + exp2 + rcp");
  return 0;
}
