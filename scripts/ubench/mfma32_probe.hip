// Probe of the two hardware facts kprn_amd/csrc/lstm_bf16_persist.hip is built on (run on the GPU box before reading its parity tests):
//  1. v_mfma_f32_32x32x16_bf16 operand / result layout: A lane (m, kg) holds A[m][8 kg .. + 7], B lane (n, kg) holds B[8 kg .. + 7][n],
//     D lane (n, half) register r holds D[(r & 3) + 8 (r >> 2) + 4 half][n]
//  2. global_load_lds_dwordx4 with M0 = LDS byte address: lane l's 16 bytes land at M0 + 16 l, also above 64 KB
// hipcc --offload-arch=gfx950 -O2 scripts/ubench/mfma32_probe.hip -o scripts/ubench/build/mfma32_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_mfma(const bf16* A /*[32][16]*/, const bf16* B /*[16][32]*/, float* D /*[32][32]*/) {
  const int lane = threadIdx.x, i = lane & 31, kg = lane >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = A[i * 16 + 8 * kg + j]; b[j] = B[(8 * kg + j) * 32 + i]; }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + i] = acc[r];
}

__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }
__global__ void k_dma(const int* src /*[64 lanes][4]*/, int* out /*[2][256]*/) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x;
  for (int which = 0; which < 2; ++which) {
    char* dst = smem + (which ? 150 * 1024 : 1024);
    // lane l fetches record (63 - l): the LDS image must come out reversed
    const int* g = src + (63 - lane) * 4;
    unsigned keep;
    unsigned d = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_off(dst));
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(d) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int q = 0; q < 4; ++q) out[which * 256 + lane * 4 + q] = ((const int*)dst)[lane * 4 + q];
    __syncthreads();
  }
}

int main() {
  std::vector<bf16> A(32 * 16), B(16 * 32);
  std::vector<float> Af(32 * 16), Bf(16 * 32), D(32 * 32), R(32 * 32, 0.f);
  srand(3);
  for (int i = 0; i < 32 * 16; ++i) { Af[i] = (float)(rand() % 9 - 4); A[i] = (bf16)Af[i]; }
  for (int i = 0; i < 16 * 32; ++i) { Bf[i] = (float)(rand() % 7 - 3); B[i] = (bf16)Bf[i]; }
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) R[m * 32 + n] += Af[m * 16 + k] * Bf[k * 32 + n];
  bf16 *dA, *dB; float* dD;
  hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, D.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32 * 32; ++i) bad += D[i] != R[i];
  printf("mfma_f32_32x32x16_bf16 layout: %s (%d of 1024 wrong)\n", bad ? "MISMATCH" : "ok", bad);
  // DMA
  std::vector<int> src(256), out(512);
  for (int i = 0; i < 256; ++i) src[i] = 1000 + i;
  int *dS, *dO;
  hipMalloc(&dS, 1024); hipMalloc(&dO, 2048);
  hipMemcpy(dS, src.data(), 1024, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
  hipLaunchKernelGGL(k_dma, dim3(1), dim3(64), 156 * 1024, 0, dS, dO);
  hipError_t e = hipDeviceSynchronize();
  hipMemcpy(out.data(), dO, 2048, hipMemcpyDeviceToHost);
  for (int which = 0; which < 2; ++which) {
    int wrong = 0;
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) wrong += out[which * 256 + l * 4 + q] != 1000 + (63 - l) * 4 + q;
    printf("global_load_lds_dwordx4 -> LDS + %d KB: %s (%d wrong) [%s]\n", which ? 150 : 1, wrong ? "MISMATCH" : "ok", wrong, hipGetErrorString(e));
    bad += wrong;
  }
  return bad ? 1 : 0;
}
