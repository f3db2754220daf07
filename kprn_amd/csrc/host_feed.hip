// Host side of the streaming batch feed (kprn_batch_feed_async, "feed_build" = host): everything the engine derives from a
// minibatch's ids BEFORE any kernel can run on it -- id validation, the identical-prefix plan, the entity-occurrence index --
// computed by worker threads on the host cores, so that the GPU sees the next batch only as DMA traffic (copy engines, no CUs)
// while the persistent kernels of the current step own every CU.  (The same derivation as ~45 small dependent kernels on a
// side stream -- batch_index.hip, "feed_build" = device -- finds free CUs only in the tails of the persistent kernels and cost
// 0.28 ms of a 1.6 ms step; measured, profiles/r02.)
//
// Stands where BatcherFileList:populateGPUTensor stands in the reference (model/batcher/BatcherFileList.lua:78-96: the loader
// thread prepares the minibatch tensors, :copy() moves them).  The arrays are bit-identical to the device builders' (both sorts
// are stable; tests/test_gpu_feed.py compares them).  Pure host code: no kernels in this file.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <thread>

#include "kprn_internal.h"

namespace hostfeed {

// ---- a small worker pool (jobs = whole batches; a job fans out over helper threads for its data-parallel phases) ----
class Pool {
 public:
  explicit Pool(int n) {
    for (int i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> g(m_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  std::future<void> submit(std::function<void()> fn) {
    auto task = std::make_shared<std::packaged_task<void()>>(std::move(fn));
    std::future<void> f = task->get_future();
    {
      std::lock_guard<std::mutex> g(m_);
      q_.emplace_back([task] { (*task)(); });
    }
    cv_.notify_one();
    return f;
  }

 private:
  void run() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return stop_ || !q_.empty(); });
        if (q_.empty()) return;
        job = std::move(q_.front());
        q_.pop_front();
      }
      job();
    }
  }
  std::vector<std::thread> workers_;
  std::deque<std::function<void()>> q_;
  std::mutex m_;
  std::condition_variable cv_;
  bool stop_ = false;
};

// Data-parallel phases of a job run on a second pool of persistent helper threads shared by all jobs (threads spawned per
// phase cost more than the phases in a process with gigabytes of page-locked memory mapped; helper tasks never wait on other
// tasks, so sharing cannot deadlock).  fn(part, nparts); the caller runs part 0.
static Pool* g_helpers = nullptr;
static std::once_flag g_helpers_once;
static Pool* helpers() {
  std::call_once(g_helpers_once, [] {
    const unsigned hc = std::thread::hardware_concurrency();
    g_helpers = new Pool(hc >= 64 ? 24 : (hc >= 16 ? 8 : 3));
  });
  return g_helpers;
}
template <class Fn>
static void parallel(int nparts, Fn fn) {
  if (nparts <= 1) { fn(0, 1); return; }
  Pool* hp = helpers();
  std::vector<std::future<void>> fs;
  fs.reserve((size_t)nparts - 1);
  for (int p = 1; p < nparts; ++p) fs.push_back(hp->submit([&fn, p, nparts] { fn(p, nparts); }));
  fn(0, nparts);
  for (auto& f : fs) f.get();
}
static inline void span(int64_t n, int part, int nparts, int64_t* lo, int64_t* hi) {
  const int64_t per = (n + nparts - 1) / nparts;
  *lo = std::min<int64_t>(n, per * part);
  *hi = std::min<int64_t>(n, *lo + per);
}

void gather_rows(int32_t* dst, const int32_t* src, int64_t row_words, const int64_t* rows, int64_t n, int nth) {
  parallel(std::max(1, nth), [&](int part, int nparts) {
    int64_t lo, hi;
    span(n, part, nparts, &lo, &hi);
    for (int64_t i = lo; i < hi; ++i) memcpy(dst + i * row_words, src + rows[i] * row_words, (size_t)row_words * sizeof(int32_t));
  });
}

// printf("%.5f") of a float32 in [0, 2^20), exactly: v = M 2^e with a 24-bit M, so M * 10^5 (< 2^41) is an exact integer and the
// 5-decimal rounding is an integer shift with round-half-to-even on the exact remainder -- what glibc does on the exact binary
// value (ties do occur: 1/64 = 0.015625 -> "0.01562").  Anything else (negative, huge, inf, nan) goes through snprintf.
static int fmt_5f(char* out, float v) {
  if (!(v >= 0.0f) || v >= 1048576.0f) return snprintf(out, 64, "%.5f", (double)v);
  int e;
  const float fr = frexpf(v, &e);                                  // v = fr 2^e, fr in [0.5, 1)
  const unsigned long long M = (unsigned long long)ldexpf(fr, 24); // exact: 24-bit integer
  const int sh = 24 - e;                                           // v = M / 2^sh
  unsigned long long q;
  const unsigned long long num = M * 100000ULL;
  if (sh <= 0) q = num << (-sh);
  else if (sh >= 64) q = 0;
  else {
    q = num >> sh;
    const unsigned long long rem = num & ((1ULL << sh) - 1), half = 1ULL << (sh - 1);
    if (rem > half || (rem == half && (q & 1ULL))) ++q;
  }
  const unsigned long long ip = q / 100000ULL;
  unsigned fp = (unsigned)(q % 100000ULL);
  char tmp[24]; int k = 0, m = 0;
  unsigned long long c = ip;
  do { tmp[k++] = (char)('0' + c % 10); c /= 10; } while (c);
  while (k) out[m++] = tmp[--k];
  out[m++] = '.';
  for (int d = 4; d >= 0; --d) { out[m + d] = (char)('0' + fp % 10); fp /= 10; }
  return m + 5;
}

// eval/test_from_checkpoint.lua:110-118: one line per pair, counter \t string.format("%.5f", score) \t label (Lua 5.1's number
// concatenation = "%.14g").  Chunks are formatted in parallel into per-thread strings and laid out in order.
int64_t format_scores(int64_t counter0, const float* probs, const float* labels, int64_t n, char* out, int64_t cap, int nth) {
  nth = (int)std::max<int64_t>(1, std::min<int64_t>(nth, n / 4096));
  std::vector<std::string> parts((size_t)nth);
  parallel(nth, [&](int part, int nparts) {
    int64_t lo, hi;
    span(n, part, nparts, &lo, &hi);
    std::string& s = parts[(size_t)part];
    s.reserve((size_t)(hi - lo) * 24);
    char buf[96];
    for (int64_t i = lo; i < hi; ++i) {
      int m = 0;
      {  // counter
        char tmp[24]; int k = 0;
        unsigned long long c = (unsigned long long)(counter0 + i);
        if (counter0 + i < 0) { m = snprintf(buf, sizeof(buf), "%lld", (long long)(counter0 + i)); }
        else { do { tmp[k++] = (char)('0' + c % 10); c /= 10; } while (c); while (k) buf[m++] = tmp[--k]; }
      }
      buf[m++] = '\t';
      m += fmt_5f(buf + m, probs[i]);
      buf[m++] = '\t';
      const float lb = labels[i];
      if (lb == 1.0f) buf[m++] = '1';
      else if (lb == 0.0f && !std::signbit(lb)) buf[m++] = '0';
      else m += snprintf(buf + m, sizeof(buf) - (size_t)m, "%.14g", (double)lb);
      buf[m++] = '\n';
      s.append(buf, (size_t)m);
    }
  });
  int64_t total = 0;
  for (const std::string& s : parts) total += (int64_t)s.size();
  if (total > cap) return -total;   // (the caller sizes the buffer from this and calls again)
  int64_t at = 0;
  for (const std::string& s : parts) { memcpy(out + at, s.data(), s.size()); at += (int64_t)s.size(); }
  return total;
}

static int bits_for(int64_t v) { int b = 1; while (((int64_t)1 << b) < v) ++b; return b; }

// stable LSD radix sort of (key, val) pairs, 11-bit digits, parallel over contiguous chunks: chunk c counts its digits, the
// write offsets run over (digit, chunk) in that order, and every chunk scatters its elements in their order -- equal keys keep
// their relative order, like rocprim::radix_sort_pairs on the device
template <int RB>
static void radix_sort_pairs_rb(int32_t* k0, int32_t* v0, int32_t* k1, int32_t* v1, int64_t n, int bits, int nth, int32_t** kout, int32_t** vout) {
  constexpr int NB = 1 << RB;
  int32_t *ks = k0, *vs = v0, *kd = k1, *vd = v1;
  std::vector<int64_t> hist((size_t)nth * NB);
  for (int shift = 0; shift < bits; shift += RB) {
    std::fill(hist.begin(), hist.end(), 0);
    parallel(nth, [&](int p, int np) {
      int64_t lo, hi;
      span(n, p, np, &lo, &hi);
      int64_t* h = hist.data() + (size_t)p * NB;
      for (int64_t i = lo; i < hi; ++i) ++h[((uint32_t)ks[i] >> shift) & (NB - 1)];
    });
    int64_t run = 0;
    for (int d = 0; d < NB; ++d)
      for (int p = 0; p < nth; ++p) {
        const int64_t c = hist[(size_t)p * NB + d];
        hist[(size_t)p * NB + d] = run;
        run += c;
      }
    parallel(nth, [&](int p, int np) {
      int64_t lo, hi;
      span(n, p, np, &lo, &hi);
      int64_t* h = hist.data() + (size_t)p * NB;
      for (int64_t i = lo; i < hi; ++i) {
        const int64_t o = h[((uint32_t)ks[i] >> shift) & (NB - 1)]++;
        kd[o] = ks[i];
        vd[o] = vs[i];
      }
    });
    std::swap(ks, kd);
    std::swap(vs, vd);
  }
  *kout = ks;
  *vout = vs;
}
// (a stable LSD radix sort gives the same order whatever the digit width: a 128-pair minibatch -- 1.5 k positions -- spends its time clearing and
//  scanning the 2 048-entry histograms of the 11-bit passes, so small inputs take 8-bit digits: 28 -> 9 us of the drop-in step's host time)
static void radix_sort_pairs(int32_t* k0, int32_t* v0, int32_t* k1, int32_t* v1, int64_t n, int bits, int nth, int32_t** kout, int32_t** vout) {
  if (n < 16384) radix_sort_pairs_rb<8>(k0, v0, k1, v1, n, bits, nth, kout, vout);
  else radix_sort_pairs_rb<11>(k0, v0, k1, v1, n, bits, nth, kout, vout);
}

// ids of one batch -> validation, plan, index.  All outputs are caller-provided (page-locked staging of the slot); w0..w3 are
// work arrays of n_index int32 each.  Mirrors kk::validate_indices, bidx::prefix_plan and bidx::build line by line.
void build(const Shape& g, const int32_t* idx, int kcap, int nth, bool want_index, Result* r, int32_t* idx_s, int32_t* perm, int32_t* slot_of, int32_t* tile_k,
           int32_t* pmeta /*[8+16]*/, int32_t* key_sorted, int32_t* pos_sorted, int32_t* uniq, int32_t* w0, int32_t* w1, int32_t* w2, int32_t* w3) {
  const int B = g.B, P = g.P, T = g.T, F = g.F, nT = g.nT;
  const int64_t N = (int64_t)B * P, nsteps = N * T;
  const int c0 = F - nT - 2;
  static const bool timing = KPRN_DEV_ENV("KPRN_FEED_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto tlast = now();
  const auto t_begin = tlast;
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto t = now();
    fprintf(stderr, "[kprn feed] %-10s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(t - tlast).count());
    tlast = t;
  };
  const int64_t row = (int64_t)T * F;
  // ---- validation (kernels_basic.hip k_validate): type slots in 1..Vt, entity in 1..Ve, relation in 1..Vr
  std::vector<int> badp((size_t)nth, 0);
  parallel(nth, [&](int p, int np) {
    int64_t lo, hi;
    span(nsteps, p, np, &lo, &hi);
    int bad = 0;
    if (nT == 1 && F == 3) {  // the shipped layout: (type, entity, relation); unsigned compare = both bounds at once
      const uint32_t vt = (uint32_t)g.Vt, ve = (uint32_t)g.Ve, vr = (uint32_t)g.Vr;
      for (int64_t i = lo; i < hi; ++i) {
        const int32_t* f = idx + i * 3;
        bad |= ((uint32_t)(f[0] - 1) >= vt) | ((uint32_t)(f[1] - 1) >= ve) | ((uint32_t)(f[2] - 1) >= vr);
      }
    } else {
      for (int64_t i = lo; i < hi; ++i) {
        const int32_t* f = idx + i * F;
        for (int k = 0; k < nT; ++k) bad |= (f[c0 + k] < 1) | (f[c0 + k] > g.Vt);
        bad |= (f[F - 2] < 1) | (f[F - 2] > g.Ve) | (f[F - 1] < 1) | (f[F - 1] > g.Vr);
      }
    }
    badp[(size_t)p] = bad;
  });
  r->bad = false;
  for (int b : badp) r->bad |= (b != 0);
  lap("validate");
  r->kmax = 0; r->n_uniq = 0; r->exec_steps = nsteps;
  memset(r->ref, 0, sizeof(r->ref));
  if (r->bad) return;
  const int32_t* src = idx;  // the id array the index is built over
  // ---- identical-prefix plan (batch_index.hip k_find_ref / k_prefix_len / stable sort by k / k_prefix_apply)
  if (kcap > 0) {
    memset(pmeta, 0, (size_t)(8 + 16) * sizeof(int32_t));
    int64_t ref = -1;
    for (int64_t n = 0; n < N && ref < 0; ++n) {  // first path whose steps 0 and 1 carry the same ids
      const int32_t* q = idx + n * row;
      bool same = true;
      for (int c = c0; c < F; ++c) same &= (q[c] == q[F + c]);
      if (same) ref = n;
    }
    pmeta[1] = ref < 0 ? 0x7fffffff : (int32_t)ref;
    int32_t* kn = w0;
    if (ref >= 0) {
      const int32_t* q = idx + ref * row;
      for (int c = 0; c < F; ++c) pmeta[8 + c] = q[c];
      const int kmax = std::min(T - 2, kcap);
      parallel(nth, [&](int p, int np) {
        int64_t lo, hi;
        span(N, p, np, &lo, &hi);
        for (int64_t n = lo; n < hi; ++n) {
          const int32_t* rr = idx + n * row;
          int k = 0;
          while (k < kmax) {
            bool same = true;
            for (int c = c0; c < F; ++c) same &= (rr[k * F + c] == q[c]);
            if (!same) break;
            ++k;
          }
          kn[n] = k;
        }
      });
    } else {
      for (int64_t n = 0; n < N; ++n) kn[n] = 0;
    }
    lap("prefix_len");
    // stable counting sort by k ascending (longest work first)
    int64_t cnt[64] = {0};
    for (int64_t n = 0; n < N; ++n) ++cnt[kn[n]];
    int64_t off[64];
    int64_t run = 0;
    for (int k = 0; k <= kcap; ++k) { off[k] = run; run += cnt[k]; }
    int32_t* ksorted = w1;
    for (int64_t n = 0; n < N; ++n) {
      const int64_t o = off[kn[n]]++;
      perm[o] = (int32_t)n;
      ksorted[o] = kn[n];
    }
    parallel(nth, [&](int p, int np) {
      int64_t lo, hi;
      span(N, p, np, &lo, &hi);
      for (int64_t i = lo; i < hi; ++i) {
        memcpy(idx_s + i * row, idx + (int64_t)perm[i] * row, (size_t)row * sizeof(int32_t));
        slot_of[perm[i]] = (int32_t)i;
      }
    });
    lap("reorder");
    const int64_t ntiles = (N + 63) / 64;
    for (int64_t tl = 0; tl < ntiles; ++tl) {
      tile_k[tl] = ksorted[tl * 64];  // ascending: the tile's first path has its shortest prefix
      r->exec_steps -= (int64_t)tile_k[tl] * std::min<int64_t>(64, N - tl * 64);
    }
    pmeta[0] = ksorted[N - 1];
    r->kmax = pmeta[0];
    for (int c = 0; c < F && c < 16; ++c) r->ref[c] = pmeta[8 + c];
    src = idx_s;
  }
  if (!want_index) return;   // a scoring-only batch: nothing downstream walks its entity rows
  // ---- occurrence index (batch_index.hip k_keys / radix sort / run-length encode / k_drop_sentinel)
  const int64_t n_index = nsteps + kcap;
  const int sentinel = g.Ve;
  int32_t *keys = w0, *vals = w1;
  parallel(nth, [&](int p, int np) {
    int64_t lo, hi;
    span(N, p, np, &lo, &hi);
    for (int64_t n = lo; n < hi; ++n) {
      const int k0 = kcap > 0 ? tile_k[n >> 6] : 0;
      const int32_t* e = src + n * row + F - 2;
      int32_t* kk = keys + n * T;
      int32_t* vv = vals + n * T;
      for (int t = 0; t < T; ++t) {
        kk[t] = (t < k0) ? sentinel : e[t * F] - 1;
        vv[t] = (int32_t)(n * T + t);
      }
    }
    if (p == 0) {
      const int64_t npad = (N + 63) / 64 * 64;
      for (int t = 0; t < kcap; ++t) {  // the virtual positions of the prefix steps
        keys[nsteps + t] = (t < pmeta[0]) ? pmeta[8 + F - 2] - 1 : sentinel;
        vals[nsteps + t] = (int32_t)(npad * T + t);
      }
    }
  });
  lap("keys");
  int32_t *ks, *vs;
  radix_sort_pairs(keys, vals, w2, w3, n_index, bits_for((int64_t)g.Ve + 1), nth, &ks, &vs);
  lap("sort");
  memcpy(key_sorted, ks, (size_t)n_index * sizeof(int32_t));
  memcpy(pos_sorted, vs, (size_t)n_index * sizeof(int32_t));
  int64_t nu = 0;
  for (int64_t i = 0; i < n_index; ++i)
    if (i == 0 || ks[i] != ks[i - 1]) uniq[nu++] = ks[i];
  if (nu > 0 && uniq[nu - 1] == sentinel) --nu;
  r->n_uniq = (int32_t)nu;
  lap("copy+rle");
  if (timing) fprintf(stderr, "[kprn feed] job total %7.3f ms (threads %d)\n", std::chrono::duration<double, std::milli>(now() - t_begin).count(), nth);
}

Pool* make_pool(int n) { return new Pool(n); }
void free_pool(Pool* p) { delete p; }
std::future<void> submit(Pool* p, std::function<void()> fn) { return p->submit(std::move(fn)); }

}  // namespace hostfeed
