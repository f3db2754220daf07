// Entity-occurrence index of a batch: all (path, step) positions sorted by entity id (rocPRIM radix sort), and
// the sorted list of distinct entity rows.  Built once per batch on the device (kprn_batch_create); consumers:
//   * the lazy-exact optimiser and the data-parallel row exchange (distinct rows of the batch),
//   * the embedding backward (lstm_fused_bwd.hip k_entity_grad): a gather-reduce over the sorted positions
//     replaces the scatter-add of nn.LookupTable (FeatureEmbedding.lua:86) -- no atomics for rows whose
//     occurrences fall into one 64-position segment, i.e. a deterministic sum for all but the hub rows.
// New design: the reference re-derives nothing of the sort (LookupTable:accGradParameters walks the indices serially).
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>

#include "kprn_internal.h"

namespace bidx {

namespace {
__global__ void k_keys(const int32_t* __restrict__ idx, int64_t nsteps, int F, int32_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsteps) return;
  keys[i] = idx[i * F + F - 2] - 1;
  vals[i] = (int32_t)i;
}
int bits_for(int64_t v) { int b = 1; while (((int64_t)1 << b) < v) ++b; return b; }
}  // namespace

// scratch layout inside `scratch` (bytes): keys | vals | counts | rocPRIM temp
size_t scratch_bytes(int64_t nsteps, int Ve) {
  size_t t1 = 0, t2 = 0;
  int32_t* p = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, t1, p, p, p, p, (size_t)nsteps, 0, bits_for(Ve), (hipStream_t)0);
  (void)rocprim::run_length_encode(nullptr, t2, p, (size_t)nsteps, p, p, p, (hipStream_t)0);
  const size_t tmp = (t1 > t2 ? t1 : t2);
  return (size_t)nsteps * 3 * sizeof(int32_t) + ((tmp + 255) & ~(size_t)255) + 1024;
}

void build(hipStream_t s, const int32_t* idx, int64_t nsteps, int F, int Ve, int32_t* key_sorted, int32_t* pos_sorted, int32_t* uniq,
           int32_t* n_uniq_dev, void* scratch, size_t scratch_sz) {
  if (nsteps <= 0) return;
  int32_t* keys = (int32_t*)scratch;
  int32_t* vals = keys + nsteps;
  int32_t* counts = vals + nsteps;
  char* tmp = (char*)(counts + nsteps);
  tmp = (char*)(((uintptr_t)tmp + 255) & ~(uintptr_t)255);
  size_t tmp_bytes = scratch_sz - (size_t)(tmp - (char*)scratch);
  hipLaunchKernelGGL(k_keys, dim3((unsigned)((nsteps + 255) / 256)), dim3(256), 0, s, idx, nsteps, F, keys, vals);
  HIP_TRY(hipGetLastError());
  size_t need = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, keys, key_sorted, vals, pos_sorted, (size_t)nsteps, 0, bits_for(Ve), s));
  KPRN_REQUIRE(need <= tmp_bytes, KPRN_E_DEVICE, "batch index scratch too small (sort)");
  HIP_TRY(rocprim::radix_sort_pairs(tmp, need, keys, key_sorted, vals, pos_sorted, (size_t)nsteps, 0, bits_for(Ve), s));
  HIP_TRY(rocprim::run_length_encode(nullptr, need, key_sorted, (size_t)nsteps, uniq, counts, n_uniq_dev, s));
  KPRN_REQUIRE(need <= tmp_bytes, KPRN_E_DEVICE, "batch index scratch too small (rle)");
  HIP_TRY(rocprim::run_length_encode(tmp, need, key_sorted, (size_t)nsteps, uniq, counts, n_uniq_dev, s));
}

}  // namespace bidx
