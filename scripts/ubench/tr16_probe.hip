// Probe of ds_read_b64_tr_b16 as the A-operand loader of v_mfma_f32_32x32x16_bf16 from a K-MAJOR LDS tile (lstm_bf16.hip gx::k_gemm16xt:
// dx = dA W_i2g with dA handed over TRANSPOSED, [gate column][path]).  Model under test (cdna_hip_programming.md, LDS section): every lane
// supplies the address of 4 contiguous bf16; inside a 16-lane group the 4 x 16 block (lane p holds block[p >> 2][4 (p & 3) .. + 3]) comes back
// transposed, lane l element j = block[j][l & 15].  The kernel loads a [16 k][32 m] tile (pitch 64 B... any), forms the A fragment with two
// transpose reads per lane and checks it against A[m][8 kg + e] directly, then multiplies by a B fragment and checks D.
// hipcc --offload-arch=gfx950 -O2 scripts/ubench/tr16_probe.hip -o scripts/ubench/build/tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p; }
__device__ __forceinline__ bf16x4 tr_read(unsigned addr) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// AT: [16 k][pitch elements] (k-major image of A[32 m][16 k]); out: [64 lanes][8] the fragment each lane formed
__global__ void k_tr(const bf16* AT, int pitch, bf16* out) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  bf16* t = (bf16*)smem;
  for (int i = threadIdx.x; i < 16 * pitch; i += 64) t[i] = AT[i];
  __syncthreads();
  const int lane = threadIdx.x, r = lane & 31, kg = lane >> 5, p = lane & 15, mbase = r & 16;
  bf16x8 a;
  for (int h = 0; h < 2; ++h) {
    const int k = 8 * kg + 4 * h + (p >> 2), m = mbase + 4 * (p & 3);
    const bf16x4 v = tr_read(lds_off(t + k * pitch + m));
    for (int j = 0; j < 4; ++j) a[4 * h + j] = v[j];
  }
  for (int j = 0; j < 8; ++j) out[lane * 8 + j] = a[j];
}

int main() {
  const int pitch = 40;   // (elements: 80-byte rows, 8-byte aligned pieces)
  std::vector<bf16> AT(16 * pitch), out(64 * 8);
  for (int k = 0; k < 16; ++k) for (int m = 0; m < pitch; ++m) AT[k * pitch + m] = (bf16)(float)((k * 37 + m * 5 + 3) % 251);   // integers below 256: exact in bf16
  bf16 *dA, *dO;
  hipMalloc(&dA, AT.size() * 2); hipMalloc(&dO, out.size() * 2);
  hipMemcpy(dA, AT.data(), AT.size() * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 16 * pitch * 2, 0, dA, pitch, dO);
  hipMemcpy(out.data(), dO, out.size() * 2, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int e = 0; e < 8; ++e) {
      const int m = lane & 31, k = 8 * (lane >> 5) + e;
      const float want = (float)AT[k * pitch + m], got = (float)out[lane * 8 + e];
      if (want != got) { if (bad < 12) printf("lane %d e %d: got %g want %g\n", lane, e, got, want); ++bad; }
    }
  printf("tr16_probe: A fragment from a k-major tile via ds_read_b64_tr_b16: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
  if (bad) {   // dump what lane 0..19 got, to read the real layout off
    for (int lane = 0; lane < 20; ++lane) { printf("lane %2d:", lane); for (int e = 0; e < 8; ++e) printf(" %5g", (float)out[lane * 8 + e]); printf("\n"); }
  }
  return bad ? 1 : 0;
}
