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
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>

#include "kprn_internal.h"

namespace bidx {

namespace {
// Index entries: one per (path, step) position that the fused kernels really execute, plus -- when the batch has an
// identical-prefix plan -- one VIRTUAL position per prefix step (position Npad*T + t, row 0 of a tile past the last one)
// where the prefix backward (lstm_fused_prefix.hip) leaves the summed dx of all the skipped occurrences of that step.
// Skipped positions and unused virtual slots get the sentinel key Ve (sorted last, dropped from the distinct rows).
__global__ void k_keys(const int32_t* __restrict__ idx, int64_t N, int T, int F, const int32_t* __restrict__ tile_k,
                       const int32_t* __restrict__ meta, int kcap, int sentinel, int32_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t nsteps = N * T;
  if (i < nsteps) {
    const int64_t n = i / T;
    const int t = (int)(i - n * T);
    const bool skipped = tile_k && t < tile_k[n >> 6];
    keys[i] = skipped ? sentinel : idx[i * F + F - 2] - 1;
    vals[i] = (int32_t)i;
  } else if (i < nsteps + kcap) {
    const int t = (int)(i - nsteps);
    const int64_t npad = (N + 63) / 64 * 64;
    keys[i] = (t < meta[0]) ? meta[8 + F - 2] - 1 : sentinel;
    vals[i] = (int32_t)(npad * T + t);
  }
}
__global__ void k_drop_sentinel(const int32_t* __restrict__ uniq, const int32_t* __restrict__ runs, int sentinel, int32_t* __restrict__ count_out) {
  const int n = *runs;
  *count_out = (n > 0 && uniq[n - 1] == sentinel) ? n - 1 : n;
}
int bits_for(int64_t v) { int b = 1; while (((int64_t)1 << b) < v) ++b; return b; }

// ---- identical-prefix plan ---------------------------------------------------------------------------------
// A path set padded on the left (movie_data_format.py:250-254) feeds every padded path the SAME id tuple for its first
// steps, so the recurrent state after k such steps is one vector per layer, not one per path.  The reference tuple is
// taken from the data, not from a vocabulary convention: step 0 of the first path whose steps 0 and 1 carry the same
// ids (a real path never repeats a (type, entity, relation) tuple back to back).  k_n = leading steps of path n equal
// to that tuple, capped so that at least two real steps remain.
__global__ void k_find_ref(const int32_t* __restrict__ idx, int64_t N, int T, int F, int c0, int32_t* __restrict__ meta) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int32_t* r = idx + n * T * F;
  bool same = true;
  for (int c = c0; c < F; ++c) same &= (r[c] == r[F + c]);
  if (same) atomicMin(meta + 1, (int32_t)n);
}
__global__ void k_prefix_len(const int32_t* __restrict__ idx, int64_t N, int T, int F, int c0, int kcap, int32_t* __restrict__ meta,
                             int32_t* __restrict__ keys, int32_t* __restrict__ vals) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const int ref = meta[1];
  int k = 0;
  if (ref != 0x7fffffff) {
    const int32_t* q = idx + (int64_t)ref * T * F;
    const int32_t* r = idx + n * T * F;
    const int kmax = (T - 2 < kcap) ? T - 2 : kcap;
    while (k < kmax) {
      bool same = true;
      for (int c = c0; c < F; ++c) same &= (r[k * F + c] == q[c]);
      if (!same) break;
      ++k;
    }
    if (n == 0) for (int c = 0; c < F; ++c) meta[8 + c] = q[c];
  }
  keys[n] = k;
  vals[n] = (int32_t)n;
}
__global__ void k_prefix_apply(const int32_t* __restrict__ idx, int64_t N, int T, int F, const int32_t* __restrict__ ksorted,
                               const int32_t* __restrict__ perm, int32_t* __restrict__ idx_s, int32_t* __restrict__ slot_of, int32_t* __restrict__ tile_k,
                               int32_t* __restrict__ meta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t row = (int64_t)T * F;
  if (i < N * row) {
    const int64_t n = i / row;
    idx_s[i] = idx[(int64_t)perm[n] * row + (i - n * row)];
  }
  if (i < N) slot_of[perm[i]] = (int32_t)i;
  if (i < (N + 63) / 64) tile_k[i] = ksorted[i * 64];  // ascending order: the tile's first path has its shortest prefix
  if (i == 0) meta[0] = ksorted[N - 1];
}
}  // namespace

size_t prefix_scratch_bytes(int64_t N, int kcap) {
  size_t t1 = 0;
  int32_t* p = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, t1, p, p, p, p, (size_t)N, 0, bits_for(kcap + 1), (hipStream_t)0);
  return (size_t)N * 3 * sizeof(int32_t) + ((t1 + 255) & ~(size_t)255) + 1024;
}

// paths reordered by prefix length (stable, ascending: the longest tiles first), per-tile shared prefix length, the
// reference tuple.  meta: [0] longest prefix in the batch, [1] reference path (INT_MAX: none), [8..8+F) its step-0 ids.
void prefix_plan(hipStream_t s, const int32_t* idx, int64_t N, int T, int F, int nT, int kcap, int32_t* idx_s, int32_t* perm, int32_t* slot_of, int32_t* tile_k,
                 int32_t* meta, void* scratch, size_t scratch_sz) {
  int32_t* keys = (int32_t*)scratch;
  int32_t* vals = keys + N;
  int32_t* ksorted = vals + N;
  char* tmp = (char*)(ksorted + N);
  tmp = (char*)(((uintptr_t)tmp + 255) & ~(uintptr_t)255);
  const size_t tmp_bytes = scratch_sz - (size_t)(tmp - (char*)scratch);
  const int c0 = F - nT - 2;
  // (no host-to-device copy from pageable memory here: it would make the host wait for the stream -- the feed stream runs
  //  this a whole step ahead of the consumer)
  HIP_TRY(hipMemsetAsync(meta, 0, (size_t)(8 + F) * sizeof(int32_t), s));
  HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(meta + 1), 0x7fffffff, 1, s));
  const dim3 gn((unsigned)((N + 255) / 256));
  hipLaunchKernelGGL(k_find_ref, gn, dim3(256), 0, s, idx, N, T, F, c0, meta);
  hipLaunchKernelGGL(k_prefix_len, gn, dim3(256), 0, s, idx, N, T, F, c0, kcap, meta, keys, vals);
  HIP_TRY(hipGetLastError());
  size_t need = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, keys, ksorted, vals, perm, (size_t)N, 0, bits_for(kcap + 1), s));
  KPRN_REQUIRE(need <= tmp_bytes, KPRN_E_DEVICE, "prefix plan scratch too small");
  HIP_TRY(rocprim::radix_sort_pairs(tmp, need, keys, ksorted, vals, perm, (size_t)N, 0, bits_for(kcap + 1), s));
  const int64_t work = N * T * F;
  hipLaunchKernelGGL(k_prefix_apply, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, idx, N, T, F, ksorted, perm, idx_s, slot_of, tile_k, meta);
  HIP_TRY(hipGetLastError());
}

// scratch layout inside `scratch` (bytes): keys | vals | counts | rocPRIM temp;  n_index = N*T (+ kcap virtual entries)
size_t scratch_bytes(int64_t n_index, int Ve) {
  size_t t1 = 0, t2 = 0;
  int32_t* p = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, t1, p, p, p, p, (size_t)n_index, 0, bits_for((int64_t)Ve + 1), (hipStream_t)0);
  (void)rocprim::run_length_encode(nullptr, t2, p, (size_t)n_index, p, p, p, (hipStream_t)0);
  const size_t tmp = (t1 > t2 ? t1 : t2);
  return (size_t)n_index * 3 * sizeof(int32_t) + ((tmp + 255) & ~(size_t)255) + 1024 + 256;
}

// tile_k / meta / kcap: the batch's identical-prefix plan (null / null / 0: every position is indexed)
void build(hipStream_t s, const int32_t* idx, int64_t N, int T, int F, int Ve, const int32_t* tile_k, const int32_t* meta, int kcap,
           int32_t* key_sorted, int32_t* pos_sorted, int32_t* uniq, int32_t* n_uniq_dev, void* scratch, size_t scratch_sz) {
  const int64_t n_index = N * T + (tile_k ? kcap : 0);
  if (n_index <= 0) return;
  int32_t* keys = (int32_t*)scratch;
  int32_t* vals = keys + n_index;
  int32_t* counts = vals + n_index;
  char* tmp = (char*)(counts + n_index);
  tmp = (char*)(((uintptr_t)tmp + 255) & ~(uintptr_t)255);
  int32_t* runs = (int32_t*)tmp;  // first 256 bytes of the temp area: the run count
  tmp += 256;
  size_t tmp_bytes = scratch_sz - (size_t)(tmp - (char*)scratch);
  const int sentinel = Ve;
  const int bits = bits_for((int64_t)Ve + 1);
  hipLaunchKernelGGL(k_keys, dim3((unsigned)((n_index + 255) / 256)), dim3(256), 0, s, idx, N, T, F, tile_k, meta, tile_k ? kcap : 0, sentinel, keys, vals);
  HIP_TRY(hipGetLastError());
  size_t need = 0;
  HIP_TRY(rocprim::radix_sort_pairs(nullptr, need, keys, key_sorted, vals, pos_sorted, (size_t)n_index, 0, bits, s));
  KPRN_REQUIRE(need <= tmp_bytes, KPRN_E_DEVICE, "batch index scratch too small (sort)");
  HIP_TRY(rocprim::radix_sort_pairs(tmp, need, keys, key_sorted, vals, pos_sorted, (size_t)n_index, 0, bits, s));
  HIP_TRY(rocprim::run_length_encode(nullptr, need, key_sorted, (size_t)n_index, uniq, counts, runs, s));
  KPRN_REQUIRE(need <= tmp_bytes, KPRN_E_DEVICE, "batch index scratch too small (rle)");
  HIP_TRY(rocprim::run_length_encode(tmp, need, key_sorted, (size_t)n_index, uniq, counts, runs, s));
  hipLaunchKernelGGL(k_drop_sentinel, dim3(1), dim3(1), 0, s, uniq, runs, sentinel, n_uniq_dev);
  HIP_TRY(hipGetLastError());
}

}  // namespace bidx

// ---------------------------------------------------------------------------------------------------------
// Entity-table gradient as a gather-reduce over the batch's occurrence index: positions sorted by entity id are
// cut into 64-position segments, one wave per segment, lane = column of the entity slice.  All 64 dx values of a
// lane are requested up front (independent loads), then summed run by run.  A run that lies inside its segment is
// written with a plain store: no atomics, a fixed summation order.  Only runs that straddle segments (hub
// entities, the pad row) add their per-segment partial sums atomically.
//   FRAG = 1: dx in the fused backward's fragment order [(N/16)][T][D/16 waves][64 lanes][4]
//             (lane (ag, arow), register r <-> row 4 ag + r of the 16-row block, col 16 w + arow);
//   FRAG = 0: dx time-major row-major [T][N][D] (generic pipeline).
namespace bidx {
namespace {
template <int FRAG>
__global__ __launch_bounds__(256) void k_entity_grad(const float* __restrict__ DX, const int32_t* __restrict__ key_sorted,
                                                     const int32_t* __restrict__ pos_sorted, int64_t nsteps, int64_t N, int T, int D, int dt, int de,
                                                     int sentinel, float* __restrict__ gWe, int n_ent_blocks, int n_red_blocks, SlabReduce red,
                                                     SmallGrad sg, int dbg) {
  // Which workgroup does what (workgroup-uniform).  The passenger jobs (weight-gradient slab reduce, small-table gradients) stream
  // coalesced data and start at once; a gather-reduce workgroup first walks key -> position -> row (three dependent round trips) with
  // little in flight.  order 1 dispatches the passengers FIRST so that their traffic fills the time the gathers spend waiting
  // (order 0: entity workgroups first, as before; 2: alternating while both kinds last).
  const int order = (dbg >> 4) & 3;
  const int n_pass = gridDim.x - n_ent_blocks;
  int bid = blockIdx.x, rb = -1;   // entity workgroup bid, or passenger workgroup rb
  if (order == 1) { if (bid < n_pass) { rb = bid; } else bid -= n_pass; }
  else if (order == 2) {
    const int both = 2 * (n_pass < n_ent_blocks ? n_pass : n_ent_blocks);
    if (bid < both) { if (bid & 1) rb = bid >> 1; else bid >>= 1; }
    else if (n_pass > n_ent_blocks) rb = bid - n_ent_blocks;
    else bid -= n_pass;
  } else if (bid >= n_ent_blocks) rb = bid - n_ent_blocks;
  if (rb >= 0) {
    if (dbg & 2) return;
    if (rb >= n_red_blocks) { if (!(dbg & 8)) small_grad_block(sg, rb - n_red_blocks); return; }
    if (dbg & 4) return;
    const int nbx = (red.n_elem + 255) / 256;
    slab_reduce_block(red, rb % nbx, (rb / nbx) % red.ny, rb / (nbx * red.ny));
    return;
  }
  const int lane = threadIdx.x & 63;
  const int64_t seg = (int64_t)bid * 4 + (threadIdx.x >> 6);
  const int64_t base = seg * 64;
  if (base >= nsteps) return;
  const int cnt = (int)((nsteps - base < 64) ? (nsteps - base) : 64);
  // all four index reads are issued unconditionally from clamped addresses and masked afterwards: hipcc turns `cond ? load : x`
  // into a branch with the wait inside it, which made these four dependent round trips instead of one
  const int64_t mine = base + (lane < cnt ? lane : cnt - 1);
  const int key_ld = key_sorted[mine], pos_ld = pos_sorted[mine];
  const int kb_ld = key_sorted[base > 0 ? base - 1 : 0], ka_ld = key_sorted[base + cnt < nsteps ? base + cnt : nsteps - 1];
  const int my_key = (lane < cnt) ? key_ld : -1;
  if (__builtin_amdgcn_readlane(my_key, 0) == sentinel) return;  // sorted last: nothing but skipped positions from here on
  const int my_pos = (lane < cnt) ? pos_ld : 0;
  const int key_before = (base > 0) ? kb_ld : -1;
  const int key_after = (base + cnt < nsteps) ? ka_ld : -1;
  const int waves_per_group = D >> 4;  // FRAG: 16-column blocks per (16-row block, t)
  // Where the runs end, found once and lane-parallel (round 6): lane i compares its key with lane i + 1's, a ballot gives the 64-bit mask the walk below
  // tests one bit of per position.  (The walk used to find the ends itself -- two readlanes, compares and two branches for every one of the 64 positions
  // and columns' chunk -- and with six waves a SIMD that instruction stream, not the gather, was most of the launch: rows fetched in sorted order instead of
  // at random did not change its time, `KPRN_EGRAD_DBG` 64.)
  const int key_next = __shfl_down(my_key, 1, 64);
  const bool is_end = (lane < cnt) && (lane == cnt - 1 || key_next != my_key);
  const unsigned long long endmask = __ballot(is_end);
  const bool continues = __builtin_amdgcn_readlane(my_key, (cnt - 1) & 63) == key_after;   // the segment's last run goes on in the next segment
  for (int c0 = 0; c0 < de; c0 += 64) {
    const bool act = c0 + lane < de;
    const int ecol = act ? c0 + lane : de - 1;  // column inside the entity slice (idle lanes re-read the last one: loads stay unconditional)
    const int col = dt + ecol;        // column of dx
    const int coff = (col >> 4) * 256 + (col & 15) * 4;
    // the segment's 64 positions in two halves of 32: 32 gathers in flight per lane (76 VGPRs, 6 waves per SIMD; capping the kernel at 64
    // VGPRs for 8 waves makes the passenger jobs' code spill and measured 8 us slower)
    float acc = 0.f;
    bool opened_here = __builtin_amdgcn_readlane(my_key, 0) != key_before;  // the first run starts in this segment
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (hf * 32 >= cnt) break;   // (wave-uniform)
      float v[32];
#pragma unroll
      for (int i2 = 0; i2 < 32; ++i2) {
        const int i = 32 * hf + i2;
        const int p = __builtin_amdgcn_readlane(my_pos, i);
        const int n = p / T, t = p - n * T;
        int64_t off;
        if (FRAG == 1) {
          const int rr = n & 15;
          off = ((int64_t)(n >> 4) * T + t) * ((int64_t)waves_per_group * 256) + (rr >> 2) * 64 + (rr & 3) + coff;
        } else if (FRAG == 2) {
          off = (int64_t)p * de + ecol;   // compact entity slice: 4 de contiguous bytes per position
        } else {
          off = ((int64_t)t * N + n) * D + col;
        }
        // Unconditional, and unconditionally used below: 32 loads in flight.  (`cond ? DX[off] : 0` compiles to a branch per load with
        // the wait inside it.)  What the guard used to zero is harmless: idle lanes never store, a run of skipped (sentinel) positions is
        // dropped at its end, and positions past cnt (they re-read position 0) are only added after the segment's last run was written.
        v[i2] = DX[off];
      }
#pragma unroll
      for (int i2 = 0; i2 < 32; ++i2) {
        const int i = 32 * hf + i2;
        acc += v[i2];   // (positions past cnt add to a sum that is never stored: position cnt - 1 is always an end)
        if (endmask & (1ull << i)) {  // (wave-uniform) the run ends here, or the segment does
          const int k = __builtin_amdgcn_readlane(my_key, i);
          const bool whole = opened_here && !(i == cnt - 1 && continues);  // every occurrence of row k was in this segment
          if (act && k != sentinel) {
            float* dst = gWe + (int64_t)k * de + ecol;
            if (whole) *dst = acc; else if (!(dbg & 1)) unsafeAtomicAdd(dst, acc);
          }
          acc = 0.f;
          opened_here = true;
        }
      }
    }
  }
}
}  // namespace

void entity_grad(hipStream_t s, const float* DX, int frag_order, const int32_t* key_sorted, const int32_t* pos_sorted, int64_t n_index, int64_t N,
                 int T, int D, int dt, int de, int Ve, float* gWe, const SlabReduce* red, const SmallGrad* sg) {
  if (n_index <= 0 && !red && !sg) return;
  const int64_t segs = (n_index + 63) / 64;
  const int n_ent = (int)((segs + 3) / 4);
  SlabReduce r;
  memset(&r, 0, sizeof(r));
  SmallGrad g;
  memset(&g, 0, sizeof(g));
  int n_red = 0, n_sg = 0;
  if (red) { r = *red; n_red = ((r.n_elem + 255) / 256) * r.ny * r.L; }
  if (sg) { g = *sg; n_sg = g.nblocks; }
  static const int dbg = KPRN_DEV_ENV("KPRN_EGRAD_DBG") ? atoi(KPRN_DEV_ENV("KPRN_EGRAD_DBG")) : 0;   // (measurement: 1 no atomics, 2 no passenger work; 16 x workgroup order)
  const dim3 grid((unsigned)(n_ent + n_red + n_sg));
  if (frag_order == 1) hipLaunchKernelGGL(k_entity_grad<1>, grid, dim3(256), 0, s, DX, key_sorted, pos_sorted, n_index, N, T, D, dt, de, Ve, gWe, n_ent, n_red, r, g, dbg);
  else if (frag_order == 2) hipLaunchKernelGGL(k_entity_grad<2>, grid, dim3(256), 0, s, DX, key_sorted, pos_sorted, n_index, N, T, D, dt, de, Ve, gWe, n_ent, n_red, r, g, dbg);
  else hipLaunchKernelGGL(k_entity_grad<0>, grid, dim3(256), 0, s, DX, key_sorted, pos_sorted, n_index, N, T, D, dt, de, Ve, gWe, n_ent, n_red, r, g, dbg);
  HIP_TRY(hipGetLastError());
}
}  // namespace bidx

// ---------------------------------------------------------------------------------------------------------
// Data-parallel exchange of the row-sparse entity gradients (new design): every rank's packed
// buffer [count | ids | rows] has been all-gathered; the union of the rows and their sum IN RANK ORDER is built
// by one stable sort of (row id, source slot) + one gather-reduce -- no per-row atomics on a shared counter, and
// the same bits on every rank: a row has at most `world` occurrences, so a run touches at most two 64-entry
// segments and the two partial sums commute.
namespace bidx {
// ---- union by marking (the exchange's merge since round 2) -------------------------------------------------------------------
// Every rank's id list has each row at most once, so neither the union nor the sum needs a sort: rank by rank (W small launches in
// stream order = RANK ORDER, the same addition order on every replica, no atomics: inside one rank's buffer every row is unique)
// the rows are added into the gradient accumulator and flagged in a persistent [Ve] array; one rocprim::select over a counting
// iterator compacts the flagged ids into the sorted union; the touched flags are reset (the array is clean between steps).
// Replaces a stable radix sort of (row, source) pairs + run-length encode + gather-reduce (0.24 ms at 8 x 71 k rows).
namespace {
typedef float f32x4m __attribute__((ext_vector_type(4)));
// G[row][:] += rows[k][:] for the k < count rows of ONE rank's packed buffer; mark[row] = 1
__global__ void k_add_rank(const int32_t* __restrict__ buf, int cap, int de, float* __restrict__ G, int32_t* __restrict__ mark) {
  const int count = buf[0];
  const int per = de >> 2;
  const float* rows = (const float*)(buf + 4 + cap);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)count * per; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / per), c = (int)(i - (int64_t)k * per) * 4;
    const int row = buf[4 + k];
    f32x4m* dst = (f32x4m*)(G + (int64_t)row * de + c);
    *dst = *dst + *(const f32x4m*)(rows + (int64_t)k * de + c);
    if (c == 0) mark[row] = 1;
  }
}
__global__ void k_add_rank_scalar(const int32_t* __restrict__ buf, int cap, int de, float* __restrict__ G, int32_t* __restrict__ mark) {
  const int count = buf[0];
  const float* rows = (const float*)(buf + 4 + cap);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)count * de; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i / de), c = (int)(i - (int64_t)k * de);
    const int row = buf[4 + k];
    G[(int64_t)row * de + c] += rows[(int64_t)k * de + c];
    if (c == 0) mark[row] = 1;
  }
}
__global__ void k_unmark(const int32_t* __restrict__ all, int world, int cap, int64_t stride, int32_t* __restrict__ mark) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)world * cap) return;
  const int r = (int)(i / cap), k = (int)(i - (int64_t)r * cap);
  const int32_t* buf = all + (int64_t)r * stride;
  if (k < buf[0]) mark[buf[4 + k]] = 0;
}
}  // namespace

size_t merge_scratch_bytes(int64_t n, int Ve) {
  size_t t1 = 0;
  int32_t* p = nullptr;
  (void)rocprim::select(nullptr, t1, rocprim::counting_iterator<int32_t>(0), p, p, p, (size_t)Ve, (hipStream_t)0);
  return ((t1 + 255) & ~(size_t)255) + 1024;
}

// all: [world][stride] 32-bit words, each rank's buffer = {count, -, -, -, ids[cap] (each row once), rows[cap*de] (fp32)}.
// mark: persistent int32 [Ve], all zero on entry and on exit.
// Out: G rows of the union += sum over ranks (rank order); union_rows (sorted) and *union_count.
void merge_rows(hipStream_t s, const void* all, int world, int cap, int de, int Ve, float* G, int32_t* union_rows, int32_t* union_count,
                int32_t* mark, void* scratch, size_t scratch_sz, int64_t tail_words) {
  const int64_t n = (int64_t)world * cap;
  if (n <= 0) return;
  const int64_t stride = 4 + (int64_t)cap * (1 + de) + tail_words;   // (tail: the dense gradient arena riding behind the rows, kprn_api.hip)
  for (int r = 0; r < world; ++r) {   // stream order = rank order
    const int32_t* buf = (const int32_t*)all + (int64_t)r * stride;
    if ((de & 3) == 0) {
      const int64_t work = (int64_t)cap * (de >> 2);
      hipLaunchKernelGGL(k_add_rank, dim3((unsigned)std::min<int64_t>((work + 255) / 256, 16384)), dim3(256), 0, s, buf, cap, de, G, mark);
    } else {
      const int64_t work = (int64_t)cap * de;
      hipLaunchKernelGGL(k_add_rank_scalar, dim3((unsigned)std::min<int64_t>((work + 255) / 256, 16384)), dim3(256), 0, s, buf, cap, de, G, mark);
    }
  }
  size_t need = 0;
  HIP_TRY(rocprim::select(nullptr, need, rocprim::counting_iterator<int32_t>(0), mark, union_rows, union_count, (size_t)Ve, s));
  KPRN_REQUIRE(need <= scratch_sz, KPRN_E_DEVICE, "merge scratch too small (select)");
  HIP_TRY(rocprim::select(scratch, need, rocprim::counting_iterator<int32_t>(0), mark, union_rows, union_count, (size_t)Ve, s));
  hipLaunchKernelGGL(k_unmark, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const int32_t*)all, world, cap, stride, mark);
  HIP_TRY(hipGetLastError());
}

}  // namespace bidx
