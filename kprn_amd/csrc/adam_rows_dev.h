// The lazy-exact Adam row update of ONE row-lane group, as a device function two kernels share: kernels_basic.hip k_adam_rows_v (the optimiser step's row
// update, and a plain catch-up) and lstm_fused_prefix.hip k_catchup_prefix (a catch-up with the identical-prefix forward as one more workgroup).  The
// per-element arithmetic is adam_elem's wherever it is compiled: the result stays bit-identical to the dense sweep (tests/test_gpu_parity.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kk_dev {

// optim.adam's element update (MyOptimizer.lua:218).  The fused multiply-adds are WRITTEN OUT and contraction is off: left to hipcc, which products of
// `m b1 + (1 - b1) g` / `v b2 + (1 - b2) g g` / `x - step (m / denom)` it fuses depends on the code around the inlined call -- and the lazy row replay, the
// dense sweep, the union kernel of the data-parallel step and the catch-up + prefix launch must round alike (bit-identical replicas, lazy == dense:
// tests/test_gpu_parity.py).  Round 6: moving this function into a header changed one of those choices.
__device__ __forceinline__ void adam_elem(float& x, float& m, float& v, float g, float step, float b1, float b2, float eps) {
#pragma clang fp contract(off)
  m = __builtin_fmaf(m, b1, (1.f - b1) * g);
  v = __builtin_fmaf(v, b2, ((1.f - b2) * g) * g);
  const float denom = sqrtf(v) + eps;
  x = __builtin_fmaf(-step, m / denom, x);
}

struct AdamRowsArgs {
  float *W, *g, *m, *v; int32_t* last; const int32_t* rows; const int32_t* count;
  int32_t t_now; int apply_step; const float* step_tab; float b1, b2, eps; int64_t pad_row; float step_now;
};

// rows of d = 4 G floats handled by G = 8 / 16 / 32 lanes with 16-byte accesses; `block`: index among the launch's row workgroups
template <int G>
__device__ __forceinline__ void adam_rows_lane_block(const AdamRowsArgs& a, int64_t block) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t slot = (block * blockDim.x + threadIdx.x) / G;
  const int j = threadIdx.x % G;
  if (slot >= *a.count) return;
  const int64_t r = a.rows[slot];
  const int32_t l = a.last[r];
  const int32_t upto = a.apply_step ? a.t_now - 1 : a.t_now;  // replay (l, upto] with g = 0
  const int64_t o = r * (4 * G) + 4 * j;
  const f4 x4 = *(const f4*)(a.W + o), m4 = *(const f4*)(a.m + o), v4 = *(const f4*)(a.v + o);
  f4 g4 = f4{0.f, 0.f, 0.f, 0.f};
  if (a.apply_step) g4 = *(const f4*)(a.g + o);
  float x[4] = {x4[0], x4[1], x4[2], x4[3]}, mm[4] = {m4[0], m4[1], m4[2], m4[3]}, vv[4] = {v4[0], v4[1], v4[2], v4[3]};
  if (l > 0)
    for (int32_t k = l + 1; k <= upto; ++k) {
      const float st = a.step_tab[k];
#pragma unroll
      for (int q = 0; q < 4; ++q) adam_elem(x[q], mm[q], vv[q], 0.f, st, a.b1, a.b2, a.eps);
    }
  if (a.apply_step) {
    const float st = a.step_now >= 0.f ? a.step_now : a.step_tab[a.t_now];
#pragma unroll
    for (int q = 0; q < 4; ++q) adam_elem(x[q], mm[q], vv[q], g4[q], st, a.b1, a.b2, a.eps);
    *(f4*)(a.g + o) = f4{0.f, 0.f, 0.f, 0.f};
  }
  if (r == a.pad_row) { x[0] = x[1] = x[2] = x[3] = 0.f; }  // zeroPadTokens after every step the row lived through (MyOptimizer.lua:219)
  // a catch-up that finds the row current (touched by the step before: every row of a batch that comes again, most rows of consecutive minibatches) has
  // changed nothing: its three 16-byte stores per lane are skipped (round 6: they were half of the catch-up launch's traffic)
  const bool changed = a.apply_step || (l > 0 && l < upto) || r == a.pad_row;
  if (changed) { *(f4*)(a.W + o) = f4{x[0], x[1], x[2], x[3]}; *(f4*)(a.m + o) = f4{mm[0], mm[1], mm[2], mm[3]}; *(f4*)(a.v + o) = f4{vv[0], vv[1], vv[2], vv[3]}; }
  if (j == 0) a.last[r] = a.t_now;
}

}  // namespace kk_dev
