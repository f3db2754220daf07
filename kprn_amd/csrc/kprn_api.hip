// libkprn.so -- C ABI (include/kprn.h) and step orchestration.
//
// One handle = the reference's training_net (MapReduce(predictor_net, reducer) + Sigmoid,
// release/songPathRnn/model/OneModel.lua:204-294) plus MyOptimizer's state
// (model/optimizer/MyOptimizer.lua:13-72), resident in HBM.
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <functional>
#include <thread>

#include "kprn_internal.h"

namespace fused {
bool fwd_supported(const kprn_handle* h, int T);
void forward(kprn_handle* h, const kprn_batch* b, bool save, int64_t tile_begin = 0, int64_t tile_end = -1, bool ignore_reserve = false);
bool small_tiles(const kprn_handle* h, int64_t N, bool has_plan);
bool bwd_supported(const kprn_handle* h, int T);
void backward(kprn_handle* h, const kprn_batch* b, int cid);
void params_changed(kprn_handle* h);
bool transpose_job(kprn_handle* h, kk::TransposeJob* tj);
void prefix_forward(kprn_handle* h, const kprn_batch* b);
bool forward_dual(kprn_handle* h, const kprn_batch* bt, const kprn_batch* bs, float* S_score);
bool catch_up_with_prefix(kprn_handle* h, const kprn_batch* b, float* W, float* g, float* m, float* v, int32_t* last, int32_t t_now, const float* step_tab,
                          float b1, float b2, float eps);
void mc_prepare(kprn_handle* h);
void release(kprn_handle* h);
void handover_stats(kprn_handle* h, const kprn_batch* b, int64_t* out);
}  // namespace fused

static thread_local std::string g_create_error;

// ---------------------------------------------------------------------------------------
// profiling scopes
ProfScope::ProfScope(kprn_handle* h_, const char* n, hipStream_t on) : h(h_), name(n), strm(on ? on : h_->stream) {
  if (!h->prof_on) return;
  // an event pair costs ~4 us of stream time: a filter keeps the measurement of ONE kernel family from taxing all the others
  if (!h->prof_filter.empty() && strncmp(n, h->prof_filter.c_str(), h->prof_filter.size()) != 0) return;
  auto get = [&]() {
    hipEvent_t e;
    if (!h->event_pool.empty()) { e = h->event_pool.back(); h->event_pool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e;
  };
  a = get(); b = get();
  if (a) hipEventRecord(a, strm);
}
ProfScope::~ProfScope() {
  if (!h->prof_on || !a || !b) return;
  hipEventRecord(b, strm);
  h->prof_pending.push_back({name, a, b, launches});
  if (h->prof_pending.size() > 4096) prof_drain(h);
}
void prof_drain(kprn_handle* h) {
  if (h->prof_pending.empty()) return;
  if (h->score_stream) hipStreamSynchronize(h->score_stream);
  if (h->rest_stream) hipStreamSynchronize(h->rest_stream);
  hipStreamSynchronize(h->stream);
  for (auto& p : h->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      auto& e = h->prof[p.name];
      e.total_ms += ms;
      e.launches += p.launches;
    }
    h->event_pool.push_back(p.a);
    h->event_pool.push_back(p.b);
  }
  h->prof_pending.clear();
}

// ---------------------------------------------------------------------------------------
template <typename T>
static T* dalloc(int64_t n) {
  void* p = nullptr;
  if (n <= 0) n = 1;
  // 64 bytes of slack: the tiled GEMMs fetch 16-byte vectors that may straddle the end of a matrix's last row (gemm_tiled.hip)
  hipError_t e = kprn_dev_malloc(&p, (size_t)n * sizeof(T) + 64);
  if (e != hipSuccess) throw KprnError{KPRN_E_NOMEM, std::string("hipMalloc failed: ") + hipGetErrorString(e)};
  return (T*)p;
}
template <typename T>
static void dfree(T*& p) { if (p) { hipFree(p); p = nullptr; } }

static void build_layout(kprn_handle* h) {
  const kprn_config& c = h->cfg;
  h->D = c.dt + c.de + c.dr;
  int64_t flat = 0, dn = 0;
  auto add = [&](const std::string& nm, int64_t rows, int64_t cols, int where, int64_t dev_off) {
    h->params.push_back({nm, flat, rows, cols, where, dev_off});
    flat += rows * cols;
  };
  h->off_Wt = dn; add("type_emb", c.Vt, c.dt, 0, dn); dn += (int64_t)c.Vt * c.dt;
  add("entity_emb", c.Ve, c.de, 1, 0);
  h->off_Wr = dn; add("relation_emb", c.Vr, c.dr, 0, dn); dn += (int64_t)c.Vr * c.dr;
  h->G = (c.rnn_type == 1) ? 1 : ((c.rnn_type == 2) ? 2 : 4);
  for (int l = 0; l < c.L; ++l) {
    const int Din = (l == 0) ? h->D : c.H;
    h->layer[l].Din = Din;
    const std::string ln = std::to_string(l + 1);
    if (c.rnn_type == 1) {  // nn.Recurrence(nn.MaskZero(...)): input2hidden / hidden2hidden nn.Linear, both with bias (OneModel.lua:231-232)
      h->layer[l].Wi = dn; add("rnn" + ln + ".i2h.weight", c.H, Din, 0, dn); dn += (int64_t)c.H * Din;
      h->layer[l].bi = dn; add("rnn" + ln + ".i2h.bias", c.H, 1, 0, dn); dn += c.H;
      h->layer[l].Wo = dn; add("rnn" + ln + ".h2h.weight", c.H, c.H, 0, dn); dn += (int64_t)c.H * c.H;
      h->layer[l].bo = dn; add("rnn" + ln + ".h2h.bias", c.H, 1, 0, dn); dn += c.H;
    } else if (c.rnn_type == 2) {  // nn.GRU: gates r, z from i2g (nn.Linear) + o2g (nn.LinearNoBias); candidate from its own pair of maps
      h->layer[l].Wi = dn; add("gru" + ln + ".i2g.weight", 2 * c.H, Din, 0, dn); dn += (int64_t)2 * c.H * Din;
      h->layer[l].bi = dn; add("gru" + ln + ".i2g.bias", 2 * c.H, 1, 0, dn); dn += (int64_t)2 * c.H;
      h->layer[l].Wo = dn; add("gru" + ln + ".o2g.weight", 2 * c.H, c.H, 0, dn); dn += (int64_t)2 * c.H * c.H;
      h->layer[l].Wc = dn; add("gru" + ln + ".c_i2h.weight", c.H, Din, 0, dn); dn += (int64_t)c.H * Din;
      h->layer[l].bc = dn; add("gru" + ln + ".c_i2h.bias", c.H, 1, 0, dn); dn += c.H;
      h->layer[l].Uc = dn; add("gru" + ln + ".c_h2h.weight", c.H, c.H, 0, dn); dn += (int64_t)c.H * c.H;
      h->layer[l].bo = -1;
    } else {
      h->layer[l].Wi = dn; add("lstm" + ln + ".i2g.weight", 4 * c.H, Din, 0, dn); dn += (int64_t)4 * c.H * Din;
      h->layer[l].bi = dn; add("lstm" + ln + ".i2g.bias", 4 * c.H, 1, 0, dn); dn += (int64_t)4 * c.H;
      h->layer[l].Wo = dn; add("lstm" + ln + ".o2g.weight", 4 * c.H, c.H, 0, dn); dn += (int64_t)4 * c.H * c.H;
      h->layer[l].bo = -1;
    }
  }
  h->off_outW = dn; add("out.weight", c.C, c.H, 0, dn); dn += (int64_t)c.C * c.H;
  h->off_outb = dn; add("out.bias", c.C, 1, 0, dn); dn += c.C;
  h->n_dense = dn;
  h->n_ent = (int64_t)c.Ve * c.de;
  h->n_params = flat;
}

static const ParamInfo* find_param(kprn_handle* h, const char* name) {
  if (!name) return nullptr;
  for (auto& p : h->params) if (p.name == name) return &p;
  return nullptr;
}

static void step_tab_reserve(kprn_handle* h, int64_t need) {
  if (need < h->step_tab_cap) return;
  int64_t cap = std::max<int64_t>(4096, h->step_tab_cap * 2);
  while (cap <= need) cap *= 2;
  float* nd = dalloc<float>(cap);
  float* nh = nullptr;
  HIP_TRY(hipHostMalloc((void**)&nh, (size_t)cap * sizeof(float)));
  memset(nh, 0, (size_t)cap * sizeof(float));
  if (h->step_tab) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    memcpy(nh, h->step_tab_host, (size_t)h->step_tab_cap * sizeof(float));
    HIP_TRY(hipMemcpy(nd, nh, (size_t)h->step_tab_cap * sizeof(float), hipMemcpyHostToDevice));
    hipFree(h->step_tab);
    hipHostFree(h->step_tab_host);
  }
  h->step_tab = nd; h->step_tab_host = nh; h->step_tab_cap = cap;
}

// scoring overlap: the main stream waits for the pass running on the side stream (before anything that changes what that pass
// reads -- parameters, the prefix table -- and before the backward kernels, which want the chip to themselves)
static void launch_score_rest(kprn_handle* h);
void join_score(kprn_handle* h) {
  if (h->score_rest_batch) launch_score_rest(h);   // (a split pass whose second part nobody placed: it goes now)
  if (!h->score_pending) return;
  HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_score_done, 0));
  h->score_pending = false;
}

// bring every entity row up to opt_step (needed before anything reads the whole table)
static void flush_lazy(kprn_handle* h) {
  join_score(h);
  if (!h->lazy_pending) return;
  ProfScope ps(h, "adam_flush_all");
  kk::adam_flush_all(h->stream, h->We, h->s1_We, h->s2_We, h->We_last, h->cfg.Ve, h->cfg.de, (int32_t)h->opt_step, h->step_tab,
                     h->last_b1, h->last_b2, h->last_eps, (int64_t)h->cfg.Ve - 1);
  h->lazy_pending = false;
  bf16p::params_changed(h, false);   // the replay rewrote rows no row list names: the bf16 shadow of the whole table is stale
}

static void zero_pad_tokens(kprn_handle* h) {
  const kprn_config& c = h->cfg;
  join_score(h);
  kk::zero_pad3(h->stream, h->dense + h->off_Wt + (int64_t)(c.Vt - 1) * c.dt, c.dt, h->dense + h->off_Wr + (int64_t)(c.Vr - 1) * c.dr, c.dr,
                h->We + (int64_t)(c.Ve - 1) * c.de, c.de);
  h->pad_clean = true;
}

// parameters were written from outside the optimiser: every steady-state shortcut is off
static void params_touched(kprn_handle* h) {
  join_score(h);
  h->pad_clean = false;
  h->caught_serial = -1;
  fused::params_changed(h);
  bf16p::params_changed(h, false);
}

// the optimiser's row list: a view of the batch's distinct-row list until something needs an owned copy
static void view_step_rows(kprn_handle* h, const kprn_batch* b) {
  h->rows_view = b->uniq; h->count_view = b->uniq + b->uniq_cap; h->view_batch = b;
  h->step_rows_ub = b->n_uniq;
}

static void materialize_step_rows(kprn_handle* h) {
  if (!h->view_batch) { h->rows_view = h->step_rows; h->count_view = h->step_count; return; }
  const kprn_batch* b = h->view_batch;
  int64_t need = std::max<int64_t>(b->n_uniq, 1);
  if (need > h->step_rows_cap) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    dfree(h->step_rows);
    h->step_rows_cap = need * 2;
    h->step_rows = dalloc<int32_t>(h->step_rows_cap);
  }
  if (b->n_uniq > 0)
    HIP_TRY(hipMemcpyAsync(h->step_rows, b->uniq, (size_t)b->n_uniq * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->step_count, b->uniq + b->uniq_cap, sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
  h->step_rows_ub = b->n_uniq;
  h->rows_view = h->step_rows; h->count_view = h->step_count; h->view_batch = nullptr;
}

static void ensure_ws_common(kprn_handle* h, int64_t N, int64_t B) {
  if (kk::loss_partials((int)B) > h->loss_partial_cap) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    dfree(h->loss_partial);
    h->loss_partial_cap = (int64_t)kk::loss_partials((int)B) * 2;
    h->loss_partial = dalloc<float>(h->loss_partial_cap);
  }
  Workspace& w = h->ws;
  const kprn_config& c = h->cfg;
  if (N > w.cap_Nc) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    dfree(w.S); dfree(w.dS);
    w.S = dalloc<float>(N * c.C);
    w.dS = dalloc<float>(N * 2 + 16);  // [N] grads + [B<=N] per-pair loss terms
    w.cap_Nc = N;
  }
  if (B > w.cap_B) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    dfree(w.pooled); dfree(w.probs); dfree(w.sel); dfree(w.dy);
    w.pooled = dalloc<float>(B * c.C);
    w.probs = dalloc<float>(B * c.C);
    w.sel = dalloc<float>(B);
    w.dy = dalloc<float>(B);
    w.cap_B = B;
  }
  h->score_buf = w.S;
}

static void ensure_ws_generic(kprn_handle* h, int64_t N, int T) {
  Workspace& w = h->ws;
  const kprn_config& c = h->cfg;
  const int H = c.H, L = c.L, D = h->D;
  if (N > w.cap_N || T > w.cap_T) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    int64_t cn = std::max<int64_t>(N, w.cap_N);
    int ct = std::max(T, w.cap_T);
    dfree(w.X); dfree(w.Hs); dfree(w.Cs); dfree(w.ACT); dfree(w.dA); dfree(w.dIn); dfree(w.dH); dfree(w.dC); dfree(w.mask);
    w.mask = dalloc<float>((int64_t)L * ct * cn);
    w.X = dalloc<float>(cn * ct * D);
    w.Hs = dalloc<float>((int64_t)L * ct * cn * H);
    w.Cs = dalloc<float>((int64_t)L * ct * cn * H);
    w.ACT = dalloc<float>((int64_t)L * ct * cn * 4 * H);
    w.dA = dalloc<float>((int64_t)ct * cn * 4 * H);
    w.dIn = dalloc<float>((int64_t)ct * cn * std::max(D, H));
    w.dH = dalloc<float>(cn * H);
    w.dC = dalloc<float>(cn * H);
    w.cap_N = cn; w.cap_T = ct;
  }
}

// rows of this batch that are behind opt_step are replayed before the forward reads them
static void flush_lazy(kprn_handle* h);
static bool use_fused(kprn_handle* h, const kprn_batch* b, bool save_for_backward);
static void catch_up(kprn_handle* h, const kprn_batch* b) {
  if (h->lazy_pending && !b->has_index) { flush_lazy(h); return; }   // (no row list: bring the whole table up to date instead)
  if (!h->lazy_pending || b->n_uniq == 0) return;
  if (h->caught_serial == b->serial && h->caught_step == h->opt_step) return;  // this batch's rows are already current
  join_score(h);
  ProfScope ps(h, "adam_rows_catchup");
  // (fused path, a batch with an identical-prefix plan: its prefix table is one more workgroup of this launch -- fused::catch_up_with_prefix)
  const bool with_prefix = h->cfg.compute_dtype == 0 && use_fused(h, b, false) &&
                           fused::catch_up_with_prefix(h, b, h->We, h->g_We, h->s1_We, h->s2_We, h->We_last, (int32_t)h->opt_step, h->step_tab, h->last_b1, h->last_b2, h->last_eps);
  // count lives at the tail of the list buffer
  if (!with_prefix)
  kk::adam_rows(h->stream, h->We, h->g_We, h->s1_We, h->s2_We, h->We_last, b->uniq, b->uniq + b->uniq_cap, b->n_uniq, h->cfg.de,
                (int32_t)h->opt_step, 0, h->step_tab, h->last_b1, h->last_b2, h->last_eps, (int64_t)h->cfg.Ve - 1);
  bf16p::rows_updated(h, b->uniq, b->uniq + b->uniq_cap, b->n_uniq);
  h->caught_serial = b->serial; h->caught_step = h->opt_step;
}

// ---------------------------------------------------------------------------------------
// generic (unfused) forward: gather -> per layer {input GEMM, per step recurrent GEMM + gates} -> head
static void forward_generic(kprn_handle* h, const kprn_batch* b, bool save) {
  const kprn_config& c = h->cfg;
  const bool bf = c.compute_dtype == 1;  // bf16 MFMA products, fp32 accumulation (gemm_f32.hip)
  static const bool no_step = getenv("KPRN_NO_STEP_KERNEL") != nullptr;  // (measurement: GEMM + element-wise kernels per step)
  Workspace& w = h->ws;
  const int H = c.H, L = c.L, T = b->T;
  const int64_t N = (int64_t)b->B * b->P;
  hipStream_t s = h->stream;
  {
    ProfScope ps(h, "embed_gather");
    // (rnnType rnn: MaskZero's mask of the bottom layer's input rows comes out of the same pass)
    kk::embed_gather(s, b->idx, N, T, b->F, c.num_types, h->dense + h->off_Wt, h->We, h->dense + h->off_Wr, c.dt, c.de, c.dr, w.X, true,
                     c.rnn_type == 1 ? w.mask : nullptr);
  }
  if (c.rnn_type == 2) {
    // nn.Sequencer(nn.GRU(D, H)) x L (OneModel.lua:237-238,268-273); step record a[n][4H] = [r | z | n | r*h']
    for (int l = 0; l < L; ++l) {
      const int Din = h->layer[l].Din;
      const float* in = (l == 0) ? w.X : w.Hs + (int64_t)(l - 1) * T * N * H;
      float* act = w.ACT + (int64_t)l * T * N * 4 * H;
      float* hs = w.Hs + (int64_t)l * T * N * H;
      const float* Wo = h->dense + h->layer[l].Wo;
      const float* Uc = h->dense + h->layer[l].Uc;
      if (!bf && h->persist_layers && lp32::supported(2, N, Din, H, h->persist_layers == 2)) {
        // all T steps of the layer -- both dependent products of a step -- in ONE persistent launch (layer_f32_persist.hip, CELL 2)
        ProfScope ps(h, "gru_layer_fwd");
        lp32::forward_layer(s, 2, in, N, T, Din, H, h->dense + h->layer[l].Wi, Wo, h->dense + h->layer[l].bi, nullptr, hs, nullptr, act, nullptr, 0, save,
                            /*write_all_h=*/l < L - 1, h->dense + h->layer[l].Wc, Uc, h->dense + h->layer[l].bc);
        continue;
      }
      {
        ProfScope ps(h, "gemm_i2g_fwd");
        gemm::run(s, in, Din, 1, h->dense + h->layer[l].Wi, 1, Din, act, 4 * H, (int64_t)T * N, 2 * H, Din, false, h->dense + h->layer[l].bi, 1, bf);
        gemm::run(s, in, Din, 1, h->dense + h->layer[l].Wc, 1, Din, act + 2 * H, 4 * H, (int64_t)T * N, H, Din, false, h->dense + h->layer[l].bc, 1, bf);
      }
      for (int t = 0; t < T; ++t) {
        float* a_t = act + (int64_t)t * N * 4 * H;
        const float* hp = t > 0 ? hs + (int64_t)(t - 1) * N * H : nullptr;
        if (t > 0) {
          ProfScope ps(h, "gemm_o2g_fwd");
          gemm::run(s, hp, H, 1, Wo, 1, H, a_t, 4 * H, N, 2 * H, H, true, nullptr, 1, bf);
        }
        {
          ProfScope ps(h, "gru_cell_fwd");
          kk::gru_gates_fwd(s, a_t, hp, N, H);
        }
        if (t > 0) {
          ProfScope ps(h, "gemm_o2g_fwd");
          gemm::run(s, a_t + 3 * H, 4 * H, 1, Uc, 1, H, a_t + 2 * H, 4 * H, N, H, H, true, nullptr, 1, bf);
        }
        ProfScope ps(h, "gru_cell_fwd");
        kk::gru_out_fwd(s, a_t, hp, hs + (int64_t)t * N * H, N, H);
      }
    }
  } else if (c.rnn_type == 1) {
    // nn.Sequencer(nn.Recurrence(nn.MaskZero(act(i2h x_t + h2h h_{t-1}), 1))) x L (OneModel.lua:240-266,268-273)
    for (int l = 0; l < L; ++l) {
      const int Din = h->layer[l].Din;
      const float* in = (l == 0) ? w.X : w.Hs + (int64_t)(l - 1) * T * N * H;
      float* pre = w.ACT + (int64_t)l * T * N * H;
      float* hs = w.Hs + (int64_t)l * T * N * H;
      float* mask = w.mask + (int64_t)l * T * N;
      const bool stepk = !bf && !no_step && gemm::step_supported(in, Din, Din, hs, H, H, h->dense + h->layer[l].Wi, h->dense + h->layer[l].Wo, N);
      if (!stepk) {
        ProfScope ps(h, "gemm_i2g_fwd");
        gemm::run(s, in, Din, 1, h->dense + h->layer[l].Wi, 1, Din, pre, H, (int64_t)T * N, H, Din, false, h->dense + h->layer[l].bi, 1, bf);
      }
      if (l > 0) {
        ProfScope ps(h, "rnn_mask");
        kk::row_nonzero(s, in, (int64_t)T * N, Din, mask);  // layer l > 1: the mask follows the ACTUAL input rows (h^{l-1}_t), as MaskZero does
      }
      if (stepk && h->persist_layers && lp32::supported(1, N, Din, H, h->persist_layers == 2)) {
        // all T steps of the layer in ONE persistent launch: h never leaves the CU, weights stream L2 -> LDS by DMA (layer_f32_persist.hip)
        ProfScope ps(h, "rnn_layer_fwd");
        lp32::forward_layer(s, 1, in, N, T, Din, H, h->dense + h->layer[l].Wi, h->dense + h->layer[l].Wo, h->dense + h->layer[l].bi, h->dense + h->layer[l].bo, hs,
                            nullptr, pre, mask, c.use_relu == 1 ? 1 : 0, save, /*write_all_h=*/l < L - 1);   // (a scoring pass needs the top layer's last step only; the training forward writes every step: forward_layer)
        continue;
      }
      if (stepk) {
        // i2h, h2h, both biases, the activation and MaskZero in one launch per step (gemm_tiled.hip)
        ProfScope ps(h, "rnn_step_fwd");
        ps.launches = T;
        for (int t = 0; t < T; ++t)
          gemm::rnn_step(s, in + (int64_t)t * N * Din, Din, Din, h->dense + h->layer[l].Wi, h->dense + h->layer[l].bi,
                         t > 0 ? hs + (int64_t)(t - 1) * N * H : nullptr, h->dense + h->layer[l].Wo, h->dense + h->layer[l].bo, mask + (int64_t)t * N,
                         pre + (int64_t)t * N * H, hs + (int64_t)t * N * H, H, N, H, c.use_relu == 1 ? 1 : 0);
        continue;
      }
      for (int t = 0; t < T; ++t) {
        float* pre_t = pre + (int64_t)t * N * H;
        if (t > 0) {
          ProfScope ps(h, "gemm_o2g_fwd");
          gemm::run(s, hs + (int64_t)(t - 1) * N * H, H, 1, h->dense + h->layer[l].Wo, 1, H, pre_t, H, N, H, H, true, nullptr, 1, bf);
        }
        ProfScope ps(h, "rnn_cell_fwd");
        kk::rnn_cell_fwd(s, pre_t, h->dense + h->layer[l].bo, mask + (int64_t)t * N, hs + (int64_t)t * N * H, N, H, c.use_relu == 1 ? 1 : 0);
      }
      // the mask of the NEXT layer reads hs: it is complete here
    }
  } else
  for (int l = 0; l < L; ++l) {
    const int Din = h->layer[l].Din;
    const float* in = (l == 0) ? w.X : w.Hs + (int64_t)(l - 1) * T * N * H;
    float* act = w.ACT + (int64_t)l * T * N * 4 * H;
    float* hs = w.Hs + (int64_t)l * T * N * H;
    float* cs = w.Cs + (int64_t)l * T * N * H;
    const float* Wi = h->dense + h->layer[l].Wi;
    const float* bi = h->dense + h->layer[l].bi;
    const float* Wo = h->dense + h->layer[l].Wo;
    if (!bf && !no_step && h->persist_layers && lp32::supported(0, N, Din, H, h->persist_layers == 2)) {
      // all T steps of the layer in ONE persistent launch: h and c never leave the CU (layer_f32_persist.hip)
      ProfScope ps(h, "lstm_layer_fwd");
      lp32::forward_layer(s, 0, in, N, T, Din, H, Wi, Wo, bi, nullptr, hs, cs, act, nullptr, 0, save, /*write_all_h=*/l < L - 1);
      continue;
    }
    if (!bf && !no_step && gemm::step_supported(in, Din, Din, hs, H, H, Wi, Wo, N)) {
      // one launch per step: [x_t | h_{t-1}] [W_i2g | W_o2g]^T + b with the FastLSTM cell in the epilogue (gemm_tiled.hip);
      // gate values are written only when a backward follows
      ProfScope ps(h, "lstm_step_fwd");
      ps.launches = T;
      for (int t = 0; t < T; ++t)
        gemm::lstm_step(s, in + (int64_t)t * N * Din, Din, Din, Wi, bi, t > 0 ? hs + (int64_t)(t - 1) * N * H : nullptr, Wo,
                        t > 0 ? cs + (int64_t)(t - 1) * N * H : nullptr, cs + (int64_t)t * N * H, hs + (int64_t)t * N * H, H,
                        save ? act + (int64_t)t * N * 4 * H : nullptr, N, H);
      continue;
    }
    {
      ProfScope ps(h, "gemm_i2g_fwd");
      gemm::run(s, in, Din, 1, Wi, 1, Din, act, 4 * H, (int64_t)T * N, 4 * H, Din, false, bi, 1, bf);
    }
    for (int t = 0; t < T; ++t) {
      float* act_t = act + (int64_t)t * N * 4 * H;
      if (t > 0) {
        ProfScope ps(h, "gemm_o2g_fwd");
        gemm::run(s, hs + (int64_t)(t - 1) * N * H, H, 1, Wo, 1, H, act_t, 4 * H, N, 4 * H, H, true, nullptr, 1, bf);
      }
      ProfScope ps(h, "lstm_gates_fwd");
      kk::lstm_gates_fwd(s, act_t, t > 0 ? cs + (int64_t)(t - 1) * N * H : nullptr, cs + (int64_t)t * N * H, hs + (int64_t)t * N * H, N, H);
    }
  }
  {
    ProfScope ps(h, "gemm_head_fwd");
    const float* hT = w.Hs + ((int64_t)(L - 1) * T + (T - 1)) * N * H;
    gemm::run(s, hT, H, 1, h->dense + h->off_outW, 1, H, w.S, c.C, N, c.C, H, false, h->dense + h->off_outb, 1, bf);
  }
}

// every_class: also pooled / probs of all C classes (what kprn_forward_batch can hand out); else the selected class only
static void pool_stage(kprn_handle* h, const kprn_batch* b, int cid, bool every_class) {
  const kprn_config& c = h->cfg;
  Workspace& w = h->ws;
  ProfScope ps(h, "pool_sigmoid");
  kk::pool_sigmoid(h->stream, w.S, b->B, b->P, c.C, c.reducer, c.K, every_class ? w.pooled : nullptr, every_class ? w.probs : nullptr, cid, w.sel, h->sel_host_armed);
}

static void batch_ready(kprn_handle* h, const kprn_batch* cb);
static void check_batch(kprn_handle* h, const kprn_batch* b, int class_id) {
  KPRN_REQUIRE(b != nullptr, KPRN_E_ARG, "batch is NULL");
  batch_ready(h, b);  // (a slot filled by kprn_batch_feed_async: its feed was issued a step ago)
  KPRN_REQUIRE(class_id >= 1 && class_id <= h->cfg.C, KPRN_E_ARG, "classId must be in 1..C (nn.Select(2,classId), MyOptimizer.lua:126)");
}

static bool use_fused(kprn_handle* h, const kprn_batch* b, bool save_for_backward) {
  if (h->impl != 0 || h->cfg.rnn_type != 0 || !fused::fwd_supported(h, b->T)) return false;
  if (h->cfg.dt == 0 || h->cfg.de == 0) return false;  // embedding ablations run on the generic pipeline
  // compute_dtype 0 (f32 MFMA) and 2 (f32x6: exact fp32 products from bf16 pieces on the matrix cores): fused forward + backward;
  // 1 (bf16 products): the fused matrix-core forward for scoring, the generic pipeline for training
  if (h->cfg.compute_dtype == 1) return !save_for_backward;  // (2, 3: forward on the matrix cores, fp32 backward)
  return !save_for_backward || fused::bwd_supported(h, b->T);
}

static void forward_impl(kprn_handle* h, const kprn_batch* b, int class_id, bool save_for_backward, bool do_pool = true, bool every_class = true) {
  check_batch(h, b, class_id);
  const int64_t N = (int64_t)b->B * b->P;
  catch_up(h, b);
  ensure_ws_common(h, N, b->B);
  if (use_fused(h, b, save_for_backward)) {
    bool dual = false;
    if (save_for_backward && h->score_dual && h->score_rest_batch && h->score_rest_tile0 == 0 && h->ev_score_done) {
      // a deferred scoring pass ("score_dual") rides in this training forward's launch; its pooling stage follows on this stream, and the event the
      // pass's readers wait for is recorded here
      const kprn_batch* sb = h->score_rest_batch;
      dual = fused::forward_dual(h, b, sb, h->S2);
      if (dual) {
        h->score_rest_batch = nullptr;
        h->pool_defer_batch = sb; h->pool_defer_cid = h->score_rest_cid - 1;   // (its pooling stage: more workgroups of the loss stage's launch, backward_impl)
        // the pass is now part of THIS stream's order: nothing is pending on the side stream, and no event is recorded for it here (an event record is
        // 6 us of idle queue in front of the loss stage at 256 paths) -- kprn_read_probs orders its copy behind this stream when somebody reads
        h->score_pending = false;
        h->score_on_main = true;
      }
    }
    if (!dual) fused::forward(h, b, save_for_backward);
  } else if (!b->idx_valid) {
    throw KprnError{KPRN_E_ARG, "this label-less batch was fed for the fused kernels (plan only); feed it again after changing impl"};
  } else if (bf16p::supported(h, b)) {
    ensure_ws_generic(h, N, b->T);   // (fp32 cell state, dx, h_T and the head's buffers are shared with the generic pipeline)
    bf16p::forward(h, b, save_for_backward);
  } else {
    ensure_ws_generic(h, N, b->T);
    forward_generic(h, b, save_for_backward);
  }
  if (do_pool) pool_stage(h, b, class_id - 1, every_class);
  h->last_B = b->B;
}

static void scratch_reserve(void** scratch, size_t* bytes, size_t need);
static void dp_release(kprn_handle* h);
// words of the dense gradient arena riding behind the packed rows (padded so that every rank's slice of the gathered buffer keeps
// 16-byte alignment)
static inline int64_t dp_tail_words(const kprn_handle* h) { return h->dp_dense_in_pack ? ((h->n_dense + 3) & ~(int64_t)3) : 0; }

// a union recorded by kprn_sparse_grad_merge (dp_fused_update) -> the summed rows in g_We + the sorted union list, as the unfused merge
// leaves them
static void materialize_union(kprn_handle* h) {
  if (!h->dp_union_pending) return;
  h->dp_union_pending = false;
  const int64_t n = (int64_t)h->dp_world * h->dp_cap;
  const size_t need = bidx::merge_scratch_bytes(n, h->cfg.Ve);
  if (need > h->bidx_scratch_bytes) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    scratch_reserve(&h->bidx_scratch, &h->bidx_scratch_bytes, need);
  }
  if (!h->dp_mark) {
    h->dp_mark = dalloc<int32_t>(h->cfg.Ve);
    HIP_TRY(hipMemsetAsync(h->dp_mark, 0, (size_t)h->cfg.Ve * sizeof(int32_t), h->stream));
  }
  ProfScope ps(h, "dp_merge_rows");
  bidx::merge_rows(h->stream, h->dp_all, h->dp_world, h->dp_cap, h->cfg.de, h->cfg.Ve, h->g_We, h->step_rows, h->step_count, h->dp_mark, h->bidx_scratch,
                   h->bidx_scratch_bytes, dp_tail_words(h));
  h->view_batch = nullptr; h->rows_view = h->step_rows; h->count_view = h->step_count;
  h->step_rows_ub = std::min<int64_t>(n, h->cfg.Ve);
  h->ent_grads_dirty = true;
}

// zeroGradParameters (MyOptimizer.lua:186): dense arena memset; entity rows cleared by list
static void zero_grads(kprn_handle* h) {
  if (h->dp_union_pending) {   // a gathered gradient nobody applied: the pack moved the rows out of g_We, so there is nothing to clear
    h->dp_union_pending = false; h->ent_grads_dirty = false; h->step_rows_ub = 0;
  }
  if (!h->dense_grads_clean) HIP_TRY(hipMemsetAsync(h->g_dense, 0, (size_t)h->n_dense * sizeof(float), h->stream));
  if (!h->view_batch) { h->rows_view = h->step_rows; h->count_view = h->step_count; }
  if (h->ent_grads_dirty && h->step_rows_ub > 0 && h->rows_view)
    kk::clear_rows(h->stream, h->g_We, h->rows_view, h->count_view, h->step_rows_ub, h->cfg.de);
  h->ent_grads_dirty = false;
  h->dense_grads_clean = true;
}

// Layer 0 of the generic LSTM / rnn backward through the small-table identity (kprn_internal.h kk::onehot_cols): dx for the entity slice only, ONE dW
// product over [S | x_e] (the one-hot selectors are written over the last ns type columns of the saved step input, next to the entity columns),
// the type / relation blocks of gW_i2g and both table gradients from G.  GH = rows of W_i2g (4H FastLSTM, H rnn).  Returns false when the shape is not
// covered (the caller then takes the dx product + table-gradient route).
static int small_tables_ns(const kprn_handle* h, const kprn_batch* b) {
  const kprn_config& c = h->cfg;
  const int ns = (c.Vr + c.Vt + 3) & ~3;
  const bool ok = h->small_tables && c.num_types == 1 && c.rnn_type != 2 && ns <= c.dt && ns <= 128 && c.de > 0 && c.dr > 0 && b->key_sorted != nullptr && !b->tile_k &&
                  b->F >= 3;
  return ok ? ns : 0;
}
static void backward_layer0_small_tables(kprn_handle* h, const kprn_batch* b, int ns, int GH, int split, bool bf) {
  const kprn_config& c = h->cfg;
  Workspace& w = h->ws;
  const int D = h->D, T = b->T;
  const int64_t N = (int64_t)b->B * b->P, TN = (int64_t)T * N;
  hipStream_t s = h->stream;
  float* gd = h->g_dense;
  const float* Wi = h->dense + h->layer[0].Wi;
  const int NZ = ns + c.de;
  if ((int64_t)GH * NZ > h->st_ctmp_cap) {
    HIP_TRY(hipStreamSynchronize(s));
    dfree(h->st_ctmp);
    h->st_ctmp = dalloc<float>((int64_t)GH * NZ);
    h->st_ctmp_cap = (int64_t)GH * NZ;
  }
  {
    ProfScope ps(h, "gemm_bwd_dw_merged");   // Ct [GH][ns + de] = dA^T [S | x_e]
    kk::onehot_cols(s, b->idx, N, T, b->F, c.Vr, c.Vt, w.X, D, c.dt - ns, ns);
    HIP_TRY(hipMemsetAsync(h->st_ctmp, 0, (size_t)GH * NZ * sizeof(float), s));
    gemm::run(s, w.dA, 1, GH, w.X + (c.dt - ns), D, 1, h->st_ctmp, NZ, GH, NZ, TN, true, nullptr, split, bf);
  }
  {
    ProfScope ps(h, "gemm_i2g_bwd_dx_e");    // dx_e [T N][de] = dA W_i2g[:, entity columns], compact
    gemm::run(s, w.dA, GH, 1, Wi + c.dt, D, 1, w.dIn, c.de, TN, c.de, GH, false, nullptr, 1, bf);
  }
  {
    ProfScope ps(h, "small_tables_finish");
    kk::small_tables_finish(s, h->st_ctmp, ns, GH, D, c.dt, c.de, c.dr, c.Vt, c.Vr, h->dense + h->off_Wt, h->dense + h->off_Wr, Wi, gd + h->layer[0].Wi, gd + h->off_Wt,
                            gd + h->off_Wr);
  }
  {
    ProfScope ps(h, "entity_grad");
    bidx::entity_grad(s, w.dIn, /*frag_order=*/0, b->key_sorted, b->pos_sorted, b->n_index, N, T, c.de, 0, c.de, c.Ve, h->g_We);
  }
}

// scratch for W_o2g^T of the persistent BPTT launch (grow-only)
static float* lp_wot_buffer(kprn_handle* h, int H, int GH) {
  const int64_t need = (int64_t)lp32::bptt_scratch_floats(H, GH);
  if (need > h->lp_wot_cap) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    dfree(h->lp_wot);
    h->lp_wot = dalloc<float>(need);
    h->lp_wot_cap = need;
  }
  return h->lp_wot;
}

static void backward_generic(kprn_handle* h, const kprn_batch* b, int cid) {
  const kprn_config& c = h->cfg;
  const bool bf = c.compute_dtype == 1;
  Workspace& w = h->ws;
  const int H = c.H, L = c.L, D = h->D, T = b->T;
  const int64_t N = (int64_t)b->B * b->P;
  hipStream_t s = h->stream;
  float* gd = h->g_dense;
  const int st_ns = small_tables_ns(h, b);   // > 0: layer 0's input gradients through the small-table identity
  {
    ProfScope ps(h, "head_bwd");
    const float* hT = w.Hs + ((int64_t)(L - 1) * T + (T - 1)) * N * H;
    kk::head_bwd(s, w.dS, hT, h->dense + h->off_outW, N, H, cid, w.dH, gd + h->off_outW, gd + h->off_outb);
  }
  HIP_TRY(hipMemsetAsync(w.dC, 0, (size_t)N * H * sizeof(float), s));
  const int split = (int)std::min<int64_t>(1024, std::max<int64_t>(1, (T * N) / 2048));
  if (c.rnn_type == 2) {
    for (int l = L - 1; l >= 0; --l) {
      const int Din = h->layer[l].Din;
      const float* in = (l == 0) ? w.X : w.Hs + (int64_t)(l - 1) * T * N * H;
      const float* act = w.ACT + (int64_t)l * T * N * 4 * H;
      const float* hs = w.Hs + (int64_t)l * T * N * H;
      const float* Wi = h->dense + h->layer[l].Wi;
      const float* Wo = h->dense + h->layer[l].Wo;
      const float* Wc = h->dense + h->layer[l].Wc;
      const float* Uc = h->dense + h->layer[l].Uc;
      const bool has_up = (l < L - 1);
      const bool bptt2 = !bf && h->persist_layers && lp32::bptt_supported(2, N, H, h->persist_layers == 2);
      if (bptt2) {
        // the cell backward of all T steps and both recurrent products of a step (d(r h') = d pre_n c_h2h; dh' += [d pre_r | d pre_z] o2g) in ONE launch
        ProfScope ps(h, "gru_layer_bwd");
        lp32::bptt_layer(s, 2, act, nullptr, hs, nullptr, has_up ? w.dIn : w.dH, has_up, Wo, lp_wot_buffer(h, H, 3 * H), w.dA, N, T, H, 0, Uc, gd + h->layer[l].bi, gd + h->layer[l].bc);
      } else if (has_up) HIP_TRY(hipMemsetAsync(w.dH, 0, (size_t)N * H * sizeof(float), s));
      for (int t = T - 1; t >= 0 && !bptt2; --t) {
        float* dA_t = w.dA + (int64_t)t * N * 4 * H;
        const float* a_t = act + (int64_t)t * N * 4 * H;
        const float* hp = t > 0 ? hs + (int64_t)(t - 1) * N * H : nullptr;
        {
          ProfScope ps(h, "gru_cell_bwd");
          kk::gru_bwd1(s, a_t, hp, w.dH, has_up ? w.dIn + (int64_t)t * N * H : nullptr, dA_t, w.dC /* direct dh' path */, N, H);
        }
        if (t > 0) {
          ProfScope ps(h, "gemm_o2g_bwd_dh");  // d(r*h') = d pre_n * c_h2h
          gemm::run(s, dA_t + 2 * H, 4 * H, 1, Uc, H, 1, dA_t + 3 * H, 4 * H, N, H, H, false, nullptr, 1, bf);
        }
        {
          ProfScope ps(h, "gru_cell_bwd");
          kk::gru_bwd2(s, a_t, hp, dA_t, w.dC, w.dH, N, H);
        }
        if (t > 0) {
          ProfScope ps(h, "gemm_o2g_bwd_dh");  // dh' += [d pre_r | d pre_z] * o2g
          gemm::run(s, dA_t, 4 * H, 1, Wo, H, 1, w.dH, H, N, H, 2 * H, true, nullptr, 1, bf);
        }
      }
      if (T > 1) {
        ProfScope ps(h, "gemm_o2g_bwd_dw");
        gemm::run(s, w.dA + (int64_t)N * 4 * H, 1, 4 * H, hs, H, 1, gd + h->layer[l].Wo, H, 2 * H, H, (int64_t)(T - 1) * N, true, nullptr, split, bf, /*untiled=*/true);
        // c_h2h += d pre_n[1..T-1]^T (r*h')[1..T-1]
        gemm::run(s, w.dA + (int64_t)N * 4 * H + 2 * H, 1, 4 * H, act + (int64_t)N * 4 * H + 3 * H, 4 * H, 1, gd + h->layer[l].Uc, H, H, H,
                  (int64_t)(T - 1) * N, true, nullptr, split, bf);
      }
      // The input maps of the gates and of the candidate are two arrays (i2g.weight [2H][Din], c_i2h.weight [H][Din]) but ONE operand of the record's first 3H columns:
      // one dW product into a zeroed [3H][Din] image (added to the two gradients behind it) and one dx product on a packed copy -- the [H][Din] halves alone fall below
      // the tiled kernel's 256 rows, and the second dx product was a read-modify-write pass over dIn.
      const int64_t cat = (int64_t)3 * H * Din;
      if (2 * cat > h->st_ctmp_cap) {
        HIP_TRY(hipStreamSynchronize(s));
        dfree(h->st_ctmp);
        h->st_ctmp = dalloc<float>(2 * cat);
        h->st_ctmp_cap = 2 * cat;
      }
      float* wcat = h->st_ctmp;
      float* gcat = h->st_ctmp + cat;
      {
        ProfScope ps(h, "gemm_i2g_bwd_dw");
        HIP_TRY(hipMemsetAsync(gcat, 0, (size_t)cat * sizeof(float), s));
        gemm::run(s, w.dA, 1, 4 * H, in, Din, 1, gcat, Din, 3 * H, Din, (int64_t)T * N, true, nullptr, split, bf, /*untiled=*/true);
        kk::add_into(s, gd + h->layer[l].Wi, gcat, (int64_t)2 * H * Din);
        kk::add_into(s, gd + h->layer[l].Wc, gcat + (int64_t)2 * H * Din, (int64_t)H * Din);
      }
      if (!bptt2) {   // (the persistent BPTT launch forms the sums itself)
        ProfScope ps(h, "bias_colsum");
        kk::col_sum_add(s, w.dA, (int64_t)T * N, 2 * H, gd + h->layer[l].bi, 4 * H);
        kk::col_sum_add(s, w.dA + 2 * H, (int64_t)T * N, H, gd + h->layer[l].bc, 4 * H);
      }
      {
        ProfScope ps(h, "gemm_i2g_bwd_dx");
        HIP_TRY(hipMemcpyAsync(wcat, Wi, (size_t)2 * H * Din * sizeof(float), hipMemcpyDeviceToDevice, s));
        HIP_TRY(hipMemcpyAsync(wcat + (int64_t)2 * H * Din, Wc, (size_t)H * Din * sizeof(float), hipMemcpyDeviceToDevice, s));
        gemm::run(s, w.dA, 4 * H, 1, wcat, Din, 1, w.dIn, Din, (int64_t)T * N, Din, 3 * H, false, nullptr, 1, bf);
      }
    }
  } else if (c.rnn_type == 1) {
    const int relu = c.use_relu == 1 ? 1 : 0;
    for (int l = L - 1; l >= 0; --l) {
      const int Din = h->layer[l].Din;
      const float* in = (l == 0) ? w.X : w.Hs + (int64_t)(l - 1) * T * N * H;
      const float* pre = w.ACT + (int64_t)l * T * N * H;
      const float* hs = w.Hs + (int64_t)l * T * N * H;
      const float* mask = w.mask + (int64_t)l * T * N;
      const float* Wi = h->dense + h->layer[l].Wi;
      const float* Wo = h->dense + h->layer[l].Wo;
      const bool has_up = (l < L - 1);
      const bool bptt1 = !bf && h->persist_layers && lp32::bptt_supported(1, N, H, h->persist_layers == 2);
      if (bptt1) {
        // the cell backward of all T steps + the recurrent gradient in ONE persistent launch (layer_f32_persist.hip k_bptt): dh never leaves the CU
        ProfScope ps(h, "rnn_layer_bwd");
        lp32::bptt_layer(s, 1, nullptr, nullptr, hs, mask, has_up ? w.dIn : w.dH, has_up, Wo, lp_wot_buffer(h, H, H), w.dA, N, T, H, relu, nullptr, gd + h->layer[l].bi,
                         gd + h->layer[l].bo);
      } else if (has_up) HIP_TRY(hipMemsetAsync(w.dH, 0, (size_t)N * H * sizeof(float), s));
      for (int t = T - 1; t >= 0 && !bptt1; --t) {
        float* dA_t = w.dA + (int64_t)t * N * H;
        {
          ProfScope ps(h, "rnn_cell_bwd");
          kk::rnn_cell_bwd(s, pre + (int64_t)t * N * H, hs + (int64_t)t * N * H, mask + (int64_t)t * N, has_up ? w.dIn + (int64_t)t * N * H : nullptr,
                           w.dH, dA_t, N, H, relu);
        }
        if (t > 0) {
          ProfScope ps(h, "gemm_o2g_bwd_dh");
          gemm::run(s, dA_t, H, 1, Wo, H, 1, w.dH, H, N, H, H, false, nullptr, 1, bf);
        }
      }
      if (T > 1) {
        ProfScope ps(h, "gemm_o2g_bwd_dw");
        gemm::run(s, w.dA + (int64_t)N * H, 1, H, hs, H, 1, gd + h->layer[l].Wo, H, H, H, (int64_t)(T - 1) * N, true, nullptr, split, bf);
      }
      if (!(bptt1 && lp32::bptt_sums_bias(1, H))) {   // (the persistent BPTT launch forms the sums itself)
        ProfScope ps(h, "bias_colsum");  // i2h.bias and h2h.bias see the same gradient (both are added to every pre-activation)
        kk::col_sum_add(s, w.dA, (int64_t)T * N, H, gd + h->layer[l].bi, 0, gd + h->layer[l].bo);   // (one pass over dA for both)
      }
      if (l == 0 && st_ns > 0) {
        backward_layer0_small_tables(h, b, st_ns, H, split, bf);
        return;
      }
      {
        ProfScope ps(h, "gemm_i2g_bwd_dw");
        gemm::run(s, w.dA, 1, H, in, Din, 1, gd + h->layer[l].Wi, Din, H, Din, (int64_t)T * N, true, nullptr, split, bf);
      }
      {
        ProfScope ps(h, "gemm_i2g_bwd_dx");
        gemm::run(s, w.dA, H, 1, Wi, Din, 1, w.dIn, Din, (int64_t)T * N, Din, H, false, nullptr, 1, bf);
      }
    }
  } else
  for (int l = L - 1; l >= 0; --l) {
    const int Din = h->layer[l].Din;
    const float* in = (l == 0) ? w.X : w.Hs + (int64_t)(l - 1) * T * N * H;
    const float* act = w.ACT + (int64_t)l * T * N * 4 * H;
    const float* hs = w.Hs + (int64_t)l * T * N * H;
    const float* cs = w.Cs + (int64_t)l * T * N * H;
    const float* Wi = h->dense + h->layer[l].Wi;
    const float* Wo = h->dense + h->layer[l].Wo;
    const bool has_up = (l < L - 1);
    const bool bptt0 = !bf && h->persist_layers && lp32::bptt_supported(0, N, H, h->persist_layers == 2);
    if (bptt0) {
      // the cell backward of all T steps + the recurrent gradient in ONE persistent launch (layer_f32_persist.hip k_bptt): dh / dc never leave the CU
      ProfScope ps(h, "lstm_layer_bwd");
      lp32::bptt_layer(s, 0, act, cs, nullptr, nullptr, has_up ? w.dIn : w.dH, has_up, Wo, lp_wot_buffer(h, H, 4 * H), w.dA, N, T, H, 0, nullptr, gd + h->layer[l].bi);
    } else if (has_up) {
      HIP_TRY(hipMemsetAsync(w.dH, 0, (size_t)N * H * sizeof(float), s));
      HIP_TRY(hipMemsetAsync(w.dC, 0, (size_t)N * H * sizeof(float), s));
    }
    for (int t = T - 1; t >= 0 && !bptt0; --t) {
      float* dA_t = w.dA + (int64_t)t * N * 4 * H;
      {
        ProfScope ps(h, "lstm_gates_bwd");
        kk::lstm_gates_bwd(s, act + (int64_t)t * N * 4 * H, cs + (int64_t)t * N * H, t > 0 ? cs + (int64_t)(t - 1) * N * H : nullptr,
                           has_up ? w.dIn + (int64_t)t * N * H : nullptr, w.dH, w.dC, dA_t, N, H);
      }
      if (t > 0) {
        ProfScope ps(h, "gemm_o2g_bwd_dh");
        gemm::run(s, dA_t, 4 * H, 1, Wo, H, 1, w.dH, H, N, H, 4 * H, false, nullptr, 1, bf);
      }
    }
    if (T > 1) {
      ProfScope ps(h, "gemm_o2g_bwd_dw");
      // gWo[4H,H] += dA[1..T-1]^T * h[0..T-2]
      gemm::run(s, w.dA + (int64_t)N * 4 * H, 1, 4 * H, hs, H, 1, gd + h->layer[l].Wo, H, 4 * H, H, (int64_t)(T - 1) * N, true, nullptr, split, bf);
    }
    if (!(bptt0 && lp32::bptt_sums_bias(0, H))) {   // (the persistent BPTT launch forms the sums itself)
      ProfScope ps(h, "bias_colsum");
      kk::col_sum_add(s, w.dA, (int64_t)T * N, 4 * H, gd + h->layer[l].bi);
    }
    if (l == 0 && st_ns > 0) {
      backward_layer0_small_tables(h, b, st_ns, 4 * H, split, bf);
      return;
    }
    {
      ProfScope ps(h, "gemm_i2g_bwd_dw");
      gemm::run(s, w.dA, 1, 4 * H, in, Din, 1, gd + h->layer[l].Wi, Din, 4 * H, Din, (int64_t)T * N, true, nullptr, split, bf);
    }
    {
      ProfScope ps(h, "gemm_i2g_bwd_dx");
      gemm::run(s, w.dA, 4 * H, 1, Wi, Din, 1, w.dIn, Din, (int64_t)T * N, Din, 4 * H, false, nullptr, 1, bf);
    }
  }
  {
    ProfScope ps(h, "embed_scatter");
    const bool have_index = b->key_sorted != nullptr && !b->tile_k;  // (an index built for a prefix plan lives in the reordered path space)
    kk::embed_scatter(s, b->idx, N, T, b->F, c.num_types, w.dIn, c.dt, c.de, c.dr, c.Vt, c.Vr, gd + h->off_Wt, h->g_We, gd + h->off_Wr, have_index);
  }
  if (b->key_sorted != nullptr && !b->tile_k) {
    ProfScope ps(h, "entity_grad");
    bidx::entity_grad(s, w.dIn, /*frag_order=*/0, b->key_sorted, b->pos_sorted, b->n_index, N, T, D, c.dt, c.de, c.Ve, h->g_We);
  }
}

// the loss of the last backward = fixed-order sum of the loss stage's per-workgroup partials, formed on demand
static void form_loss(kprn_handle* h) {
  if (h->loss_pending <= 0) return;
  kk::sum_partials(h->stream, h->loss_partial, h->loss_pending, h->d_loss, h->loss_accumulate);
  h->loss_pending = 0;
}

static void backward_impl(kprn_handle* h, const kprn_batch* b, int class_id, int literal, float inv_batch) {
  check_batch(h, b, class_id);
  KPRN_REQUIRE(b->labels != nullptr && b->has_index, KPRN_E_ARG, "batch has no labels (targets are required, MyOptimizer.lua:179)");
  const kprn_config& c = h->cfg;
  zero_grads(h);
  h->dense_grads_clean = false;
  forward_impl(h, b, class_id, true, /*do_pool=*/false);
  Workspace& w = h->ws;
  const int cid = class_id - 1;
  float invB = inv_batch > 0.f ? inv_batch : 1.0f / (float)b->B;
  if (inv_batch <= 0.f && c.world > 1) invB = 1.0f / ((float)b->B * (float)c.world);
  const bool fusedp = use_fused(h, b, true);
  {
    // pooling + sigmoid + select + BCE + reducer backward in one launch (the nn.Linear head's weight gradient is formed by the
    // top layer's backward kernel, fused path, or by the generic pipeline's own head backward)
    ProfScope ps(h, "loss_stage");
    float* gd = h->g_dense;
    kk::TransposeJob tj;  // fused path: the backward's W^T copies, stale after an update, are rebuilt by passenger workgroups
    const bool have_tj = fusedp && fused::transpose_job(h, &tj);
    kk::PoolJob pj{h->S2, 0, 0, 0, h->sel2, nullptr};
    if (h->pool_defer_batch) { pj.B = h->pool_defer_batch->B; pj.P = h->pool_defer_batch->P; pj.cid = h->pool_defer_cid; h->pool_defer_batch = nullptr; }
    kk::loss_stage(h->stream, h->score_buf, b->labels, /*hT=*/nullptr, b->B, b->P, c.C, c.H, cid, c.reducer, c.K, literal,
                   invB, /*pooled=*/nullptr, /*probs=*/nullptr, w.sel, w.dS, fusedp ? b->slot_of : nullptr, gd + h->off_outW + (int64_t)cid * c.H, gd + h->off_outb + cid, h->loss_partial,
                   have_tj ? &tj : nullptr, h->loss_early_armed ? h->loss_mirror : nullptr, pj.B > 0 ? &pj : nullptr);
    h->loss_pending = kk::loss_partials(b->B);
    if (h->loss_early_armed) {   // what kprn_train_step_batch waits for instead of the end of the step
      h->loss_early_n = h->loss_pending;
      HIP_TRY(hipEventRecord(h->ev_loss, h->stream));
    }
    if (h->loss_accumulate) form_loss(h);   // (one single-workgroup launch; otherwise the sum is formed when somebody asks)
  }
  view_step_rows(h, b);
  if (fusedp && h->score_rest_before_bptt && h->score_rest_batch) launch_score_rest(h);   // (kprn_internal.h: into the first BPTT launch's idle tail)
  // (no join with a scoring pass on the side stream here: the backward writes nothing that pass reads -- gradients, dx, prefix sums --
  //  and making the main stream wait for the pass's last workgroups cost 3 % of the step; apply_update joins before it writes parameters)
  if (fusedp) fused::backward(h, b, cid);
  else if (bf16p::supported(h, b)) bf16p::backward(h, b, cid);
  else backward_generic(h, b, cid);
  h->ent_grads_dirty = true;
  h->grads_serial = b->serial;
}

static void apply_update_impl(kprn_handle* h, const kprn_opt* o) {
  KPRN_REQUIRE(o != nullptr, KPRN_E_ARG, "opt is NULL");
  KPRN_REQUIRE(o->method == 0 || o->method == 1, KPRN_E_ARG, "opt.method must be 0 (adagrad) or 1 (adam)");
  const kprn_config& c = h->cfg;
  join_score(h);
  hipStream_t s = h->stream;
  const bool reg = (o->regularize == 1);
  const bool dense_ent = reg || o->entity_update == 1;
  // the exchange's union: walked in place by the row update (lazy-exact Adam, fp32 table only) or materialised first
  bool fuse_union = h->dp_union_pending && o->method == 1 && !dense_ent && !h->bf16_state && h->dp_world <= (c.de >> 2) &&
                    (c.de == 32 || c.de == 64 || c.de == 128) && (h->dp_cap & 3) == 0;
  if (h->dp_union_pending && !fuse_union) materialize_union(h);
  if (!h->view_batch) { h->rows_view = h->step_rows; h->count_view = h->step_count; }  // (the exchange may have re-allocated the list)
  const int32_t* rows = h->rows_view;
  const int32_t* rcount = h->count_view;   // (re-read below if a union is materialised late)
  const float* norm2 = nullptr;
  if (reg && o->use_grad_clip) {
    ProfScope ps(h, "grad_norm");
    HIP_TRY(hipMemsetAsync(h->d_norm2, 0, sizeof(float), s));
    kk::sumsq(s, h->g_dense, h->n_dense, h->d_norm2);
    if (h->step_rows_ub > 0) kk::sumsq_rows(s, h->g_We, rows, rcount, c.de, h->d_norm2);
    norm2 = h->d_norm2;
  }
  const float l2 = reg ? o->l2 : 0.f;
  // pad rows of the dense arena: re-zeroed by the dense kernels right after the update (zeroPadTokens, MyOptimizer.lua:219)
  const int64_t z0 = h->off_Wt + (int64_t)(c.Vt - 1) * c.dt, z1 = h->off_Wr + (int64_t)(c.Vr - 1) * c.dr;
  const int64_t pad_row = (int64_t)c.Ve - 1;
  bool pad_done = false;
  if (o->method == 1) {
    h->last_b1 = o->beta1; h->last_b2 = o->beta2; h->last_eps = o->eps;
    if (dense_ent && !h->ent_dense_mode) { flush_lazy(h); h->ent_dense_mode = true; }
    if (!dense_ent && h->ent_dense_mode) {
      kk::fill_i32(s, h->We_last, c.Ve, (int32_t)h->opt_step);
      h->ent_dense_mode = false;
    }
    h->opt_step += 1;
    const int64_t t = h->opt_step;
    step_tab_reserve(h, t + 1);
    const double bc1 = 1.0 - pow((double)o->beta1, (double)t), bc2 = 1.0 - pow((double)o->beta2, (double)t);
    const float step = (float)((double)o->lr * sqrt(bc2) / bc1);
    h->step_tab_host[t] = step;  // host mirror (table growth); the device entry is written by the dense kernel
    // the row update and the dense arena's update as ONE launch where the shapes allow (two latency-bound launches of the serial tail less; option
    // "adam_merged" = "0": separately): the same arithmetic per element either way
    bool merged = false;
    if (h->adam_merged && !dense_ent && !fuse_union && h->step_rows_ub > 0) {
      ProfScope ps(h, "adam_step");
      merged = kk::adam_step_merged(s, h->We, h->g_We, h->s1_We, h->s2_We, h->We_last, rows, rcount, h->step_rows_ub, c.de, (int32_t)t, h->step_tab, pad_row,
                                    h->dense, h->g_dense, h->s1_dense, h->s2_dense, h->n_dense, step, o->beta1, o->beta2, o->eps, norm2, o->grad_clip_norm,
                                    l2, z0, c.dt, z1, c.dr, h->step_tab + t);
    }
    if (!merged) {
      ProfScope ps(h, "adam_dense");
      kk::adam_dense(s, h->dense, h->g_dense, h->s1_dense, h->s2_dense, h->n_dense, step, o->beta1, o->beta2, o->eps, norm2, o->grad_clip_norm, l2,
                     /*consume=*/1, z0, c.dt, z1, c.dr, h->step_tab + t);
    }
    if (merged) {
      h->lazy_pending = true;
      pad_done = h->pad_clean;
    } else if (dense_ent) {
      ProfScope ps(h, "adam_entity_dense");
      kk::adam_dense(s, h->We, h->g_We, h->s1_We, h->s2_We, h->n_ent, step, o->beta1, o->beta2, o->eps, norm2, o->grad_clip_norm, l2,
                     /*consume=*/0, pad_row * c.de, c.de, 0, 0, nullptr);
      if (h->step_rows_ub > 0) kk::clear_rows(s, h->g_We, rows, rcount, h->step_rows_ub, c.de);
      pad_done = true;
    } else {
      ProfScope ps(h, "adam_entity_rows");
      if (fuse_union) {
        const int64_t stride = 4 + (int64_t)h->dp_cap * (1 + c.de) + dp_tail_words(h);
        const bool ok = kk::union_adam(s, h->dp_all, h->dp_world, h->dp_cap, stride, c.de, h->We, h->s1_We, h->s2_We, h->We_last, (int32_t)t, h->step_tab,
                                       o->beta1, o->beta2, o->eps, pad_row);
        if (ok) { h->dp_union_pending = false; h->step_rows_ub = 0; h->rows_view = h->step_rows; h->count_view = h->step_count; h->view_batch = nullptr; }
        else { fuse_union = false; materialize_union(h); rows = h->rows_view; rcount = h->count_view; }
      }
      if (!fuse_union)
        kk::adam_rows(s, h->We, h->g_We, h->s1_We, h->s2_We, h->We_last, rows, rcount, h->step_rows_ub, c.de, (int32_t)t, 1,
                      h->step_tab, o->beta1, o->beta2, o->eps, pad_row);
      h->lazy_pending = true;
      pad_done = h->pad_clean;  // the pad row is re-zeroed whenever the row update touches it; untouched it stays what it was
    }
  } else {
    const float clr = (float)((double)o->lr / (1.0 + (double)h->opt_step * (double)o->lr_decay));
    {
      ProfScope ps(h, "adagrad_dense");
      kk::adagrad_dense(s, h->dense, h->g_dense, h->s1_dense, h->n_dense, clr, norm2, o->grad_clip_norm, l2, /*consume=*/1, z0, c.dt, z1, c.dr);
    }
    if (reg) {
      ProfScope ps(h, "adagrad_entity_dense");
      kk::adagrad_dense(s, h->We, h->g_We, h->s1_We, h->n_ent, clr, norm2, o->grad_clip_norm, l2, /*consume=*/0, pad_row * c.de, c.de, 0, 0);
      if (h->step_rows_ub > 0) kk::clear_rows(s, h->g_We, rows, rcount, h->step_rows_ub, c.de);
      pad_done = true;
    } else {
      ProfScope ps(h, "adagrad_entity_rows");
      kk::adagrad_rows(s, h->We, h->g_We, h->s1_We, rows, rcount, h->step_rows_ub, c.de, clr, pad_row);
      pad_done = h->pad_clean;
    }
    h->opt_step += 1;
  }
  h->ent_grads_dirty = false;
  h->dense_grads_clean = true;
  h->opt_method = o->method;
  if (pad_done) h->pad_clean = true;
  else zero_pad_tokens(h);   // MyOptimizer.lua:219
  // the rows this step updated are current; if they were one batch's rows, a forward over that batch needs no catch-up
  h->caught_serial = h->grads_serial; h->caught_step = h->opt_step;
  fused::params_changed(h);
  // bf16 shadows: the dense arena always; the entity table row by row unless the whole table moved (dense sweep)
  const bool whole_table = (o->method == 1) ? dense_ent : reg;
  bf16p::params_changed(h, !whole_table);
  if (!whole_table && h->step_rows_ub > 0) bf16p::rows_updated(h, rows, rcount, h->step_rows_ub);
}

// ---------------------------------------------------------------------------------------
#define API_BEGIN(h)                                   \
  if (!(h)) return KPRN_E_ARG;                         \
  try {                                                \
    HIP_TRY(hipSetDevice((h)->cfg.device_id));         \
    if ((h)->ho_fault && *(volatile int*)(h)->ho_fault) throw KprnError{KPRN_E_DEVICE, "a fused kernel's tile hand-over wait timed out: results since then are invalid"};
#define API_END(h)                                                                        \
  }                                                                                       \
  catch (const KprnError& e) { (h)->err = e.msg; return e.code; }                         \
  catch (const std::exception& e) { (h)->err = e.what(); return KPRN_E_DEVICE; }          \
  catch (...) { (h)->err = "unknown error"; return KPRN_E_DEVICE; }                       \
  return KPRN_OK;

extern "C" {

const char* kprn_version(void) { return "kprn-amd 0.1 gfx950 f32-mfma"; }

const char* kprn_last_error(const kprn_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int kprn_create(const kprn_config* cfg, kprn_handle** out) {
  if (!cfg || !out) { g_create_error = "kprn_create: NULL argument"; return KPRN_E_ARG; }
  *out = nullptr;
  kprn_handle* h = nullptr;
  try {
    const kprn_config& c = *cfg;
    KPRN_REQUIRE(c.Vt > 0 && c.Ve > 0 && c.Vr > 0, KPRN_E_ARG, "vocab sizes must be positive");
    // dt == 0: no entity-type table (-includeEntityTypes 0), de == 0: no entity table (-includeEntity 0): the embedding variants of
    // OneModel.lua:210-219 / FeatureEmbedding.lua:26-34,83-110 -- x_t = [types | relations], [entities | relations] or [relations]
    KPRN_REQUIRE(c.dt >= 0 && c.de >= 0 && c.dr > 0, KPRN_E_ARG, "embedding dims: relation > 0, type / entity >= 0 (0 = that table is not part of the model)");
    KPRN_REQUIRE(c.num_types >= 1, KPRN_E_ARG, "numEntityTypes must be >= 1");
    KPRN_REQUIRE(c.num_types <= c.F, KPRN_E_ARG, "assert(numEntityTypes <= numFeatureTemplates) (OneModel.lua:107)");
    KPRN_REQUIRE(c.F >= c.num_types + 2, KPRN_E_ARG, "numFeatureTemplates must cover types + entity + relation (FeatureEmbedding.lua:51)");
    KPRN_REQUIRE(c.H > 0 && c.C > 0, KPRN_E_ARG, "rnnHidSize and labelDimension must be positive");
    KPRN_REQUIRE(c.L >= 1 && c.L <= KPRN_MAX_LAYERS, KPRN_E_ARG, "numLayers must be in 1..8");
    KPRN_REQUIRE(c.compute_dtype >= 0 && c.compute_dtype <= 3, KPRN_E_ARG,
                 "compute_dtype must be 0 (f32 MFMA), 1 (bf16 MFMA products, f32 accumulate), 2 (f32x6: fp32 products from 3 bf16 pieces) or "
                 "3 (f32x3: from 2 fp16 pieces)");
    KPRN_REQUIRE(c.rnn_type >= 0 && c.rnn_type <= 2, KPRN_E_ARG, "rnn_type must be 0 (lstm), 1 (rnn) or 2 (gru)");
    KPRN_REQUIRE(c.reducer >= 0 && c.reducer <= 2, KPRN_E_ARG, "topK must be 0 (max), 1 (topK) or 2 (LogSumExp)");
    KPRN_REQUIRE(c.reducer != 1 || c.K >= 1, KPRN_E_ARG, "K must be >= 1 for the topK reducer");
    KPRN_REQUIRE(c.L == 1 || (c.dt + c.de + c.dr) == c.H, KPRN_E_ARG,
                 "numLayers > 1 needs totalInputEmbeddingDim == rnnHidSize: every layer is FastLSTM(D,H) (OneModel.lua:236,270-273)");
    KPRN_REQUIRE(c.world >= 1 && c.rank >= 0 && c.rank < c.world, KPRN_E_ARG, "bad rank/world");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    KPRN_REQUIRE(ndev > 0, KPRN_E_DEVICE, "no HIP device visible: libkprn has no CPU path");
    KPRN_REQUIRE(c.device_id >= 0 && c.device_id < ndev, KPRN_E_ARG, "device_id out of range");
    HIP_TRY(hipSetDevice(c.device_id));
    h = new kprn_handle();
    h->cfg = c;
    if (c.stream == KPRN_STREAM_LEGACY_DEFAULT) { h->stream = nullptr; h->own_stream = false; h->stream_known = true; }   // the null stream, by request
    else if (c.stream) { h->stream = (hipStream_t)c.stream; h->own_stream = false; h->stream_known = true; }
    else { HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; h->stream_known = false; }
    build_layout(h);
    h->dense = dalloc<float>(h->n_dense); h->g_dense = dalloc<float>(h->n_dense);
    h->s1_dense = dalloc<float>(h->n_dense); h->s2_dense = dalloc<float>(h->n_dense);
    h->We = dalloc<float>(h->n_ent); h->g_We = dalloc<float>(h->n_ent);
    h->s1_We = dalloc<float>(h->n_ent); h->s2_We = dalloc<float>(h->n_ent);
    h->We_last = dalloc<int32_t>(c.Ve);
    h->d_loss = dalloc<float>(4); h->d_norm2 = dalloc<float>(4); h->d_flag = dalloc<int32_t>(4);
    h->step_count = dalloc<int32_t>(4);
    HIP_TRY(hipHostMalloc((void**)&h->h_pinned, 64 * sizeof(float)));
    hipStream_t s = h->stream;
    for (float* p : {h->g_dense, h->s1_dense, h->s2_dense}) HIP_TRY(hipMemsetAsync(p, 0, (size_t)h->n_dense * sizeof(float), s));
    for (float* p : {h->g_We, h->s1_We, h->s2_We}) HIP_TRY(hipMemsetAsync(p, 0, (size_t)h->n_ent * sizeof(float), s));
    HIP_TRY(hipMemsetAsync(h->We_last, 0, (size_t)c.Ve * sizeof(int32_t), s));
    HIP_TRY(hipMemsetAsync(h->d_loss, 0, 4 * sizeof(float), s));
    HIP_TRY(hipMemsetAsync(h->step_count, 0, 4 * sizeof(int32_t), s));
    // param:uniform(-paramInit, paramInit) over training_net:parameters() (OneModel.lua:306-309)
    for (auto& p : h->params) {
      float* dst = (p.where == 1 ? h->We : h->dense) + p.dev_off;
      kk::fill_uniform(s, dst, p.rows * p.cols, c.param_init, c.seed, (uint64_t)p.flat_off);
    }
    if (c.rnn_type == 1 && c.rnn_init == 1) {
      // -rnnInitialization 1 (OneModel.lua:310-322): i2h.weight <- torch.eye(D, H) copied in STORAGE order into the [H, D]
      // tensor, h2h.weight <- eye(H), both biases <- 0
      for (int l = 0; l < c.L; ++l) {
        const int Din = h->layer[l].Din;
        std::vector<float> wi((size_t)c.H * Din, 0.f), wh((size_t)c.H * c.H, 0.f), z((size_t)c.H, 0.f);
        for (int r = 0; r < Din; ++r) if (r < c.H) wi[(size_t)r * c.H + r] = 1.f;  // eye(Din, H)[r][r], flat index r*H + r
        for (int r = 0; r < c.H; ++r) wh[(size_t)r * c.H + r] = 1.f;
        HIP_TRY(hipMemcpyAsync(h->dense + h->layer[l].Wi, wi.data(), wi.size() * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->dense + h->layer[l].Wo, wh.data(), wh.size() * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->dense + h->layer[l].bi, z.data(), z.size() * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->dense + h->layer[l].bo, z.data(), z.size() * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
      }
    }
    step_tab_reserve(h, 1);
    HIP_TRY(hipStreamSynchronize(s));
    *out = h;
    return KPRN_OK;
  } catch (const KprnError& e) {
    g_create_error = e.msg;
    if (h) kprn_destroy(h);
    return e.code;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    if (h) kprn_destroy(h);
    return KPRN_E_DEVICE;
  }
}

void kprn_destroy(kprn_handle* h) {
  if (!h) return;
  hipSetDevice(h->cfg.device_id);
  if (h->score_stream) hipStreamSynchronize(h->score_stream);
  if (h->rest_stream) hipStreamSynchronize(h->rest_stream);
  if (h->stream) hipStreamSynchronize(h->stream);
  prof_drain(h);
  for (kprn_batch*& d : h->dropin_slot) if (d) { kprn_batch* old = d; d = nullptr; kprn_batch_destroy(h, old); }
  if (h->score_stream) { hipStreamDestroy(h->score_stream); hipEventDestroy(h->ev_fork); hipEventDestroy(h->ev_score_done); }
  if (h->feed_pool) { hostfeed::free_pool((hostfeed::Pool*)h->feed_pool); h->feed_pool = nullptr; }  // (joins the workers)
  if (h->upload_pool) { hostfeed::free_pool((hostfeed::Pool*)h->upload_pool); h->upload_pool = nullptr; }
  if (h->upload_stream) { hipStreamSynchronize(h->upload_stream); hipStreamDestroy(h->upload_stream); h->upload_stream = nullptr; }
  if (h->feed_stream) { hipStreamSynchronize(h->feed_stream); hipStreamDestroy(h->feed_stream); hipEventDestroy(h->ev_feed_fork); }
  if (h->feed_scratch) { hipFree(h->feed_scratch); h->feed_scratch = nullptr; }
  dp_release(h);
  dfree(h->S2); dfree(h->sel2); dfree(h->st_ctmp); dfree(h->lp_wot);
  fused::release(h);
  bf16p::release(h);
  if (h->bidx_scratch) { hipFree(h->bidx_scratch); h->bidx_scratch = nullptr; }
  dfree(h->loss_partial);
  if (h->rest_stream) { hipStreamSynchronize(h->rest_stream); hipStreamDestroy(h->rest_stream); h->rest_stream = nullptr; }
  if (h->ev_part1) { hipEventDestroy(h->ev_part1); h->ev_part1 = nullptr; }
  if (h->loss_mirror) { hipHostFree(h->loss_mirror); h->loss_mirror = nullptr; }
  if (h->probs_mirror) { hipHostFree(h->probs_mirror); h->probs_mirror = nullptr; }
  if (h->ho_fault) { hipHostFree(h->ho_fault); h->ho_fault = nullptr; }
  if (h->ev_loss) { hipEventDestroy(h->ev_loss); h->ev_loss = nullptr; }
  for (auto e : h->event_pool) hipEventDestroy(e);
  Workspace& w = h->ws;
  for (float** p : {&w.X, &w.Hs, &w.Cs, &w.ACT, &w.dA, &w.dIn, &w.dH, &w.dC, &w.S, &w.dS, &w.pooled, &w.probs, &w.sel, &w.dy, &w.mask}) dfree(*p);
  for (float** p : {&h->dense, &h->g_dense, &h->s1_dense, &h->s2_dense, &h->We, &h->g_We, &h->s1_We, &h->s2_We, &h->d_loss, &h->d_norm2,
                    &h->step_tab})
    dfree(*p);
  for (int32_t** p : {&h->We_last, &h->d_flag, &h->step_rows, &h->step_count, &h->pack_buf, &h->dp_mark}) dfree(*p);
  if (h->step_tab_host) hipHostFree(h->step_tab_host);
  if (h->h_pinned) hipHostFree(h->h_pinned);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  delete h;
}

int kprn_num_params(kprn_handle* h, int64_t* n) {
  API_BEGIN(h)
  KPRN_REQUIRE(n, KPRN_E_ARG, "n is NULL");
  *n = h->n_params;
  API_END(h)
}

static int copy_named(kprn_handle* h, const char* name, float* dst, const float* src, int64_t n, int which /*0 param,1 grad*/) {
  API_BEGIN(h)
  const ParamInfo* p = find_param(h, name);
  KPRN_REQUIRE(p, KPRN_E_ARG, std::string("unknown parameter name: ") + (name ? name : "(null)"));
  KPRN_REQUIRE(n == p->rows * p->cols, KPRN_E_ARG, "element count does not match the tensor");
  if (n == 0) return KPRN_OK;   // (a table that is not part of the model: -includeEntity 0 / -includeEntityTypes 0)
  KPRN_REQUIRE(dst || src, KPRN_E_ARG, "NULL buffer");
  if (p->where == 1 && which == 0) flush_lazy(h);
  if (which == 1) materialize_union(h);
  float* base;
  if (which == 0) base = (p->where == 1 ? h->We : h->dense);
  else base = (p->where == 1 ? h->g_We : h->g_dense);
  base += p->dev_off;
  if (dst) {
    HIP_TRY(hipMemcpyAsync(dst, base, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  } else {
    HIP_TRY(hipMemcpyAsync(base, src, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (which == 0) params_touched(h);
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  API_END(h)
}

int kprn_get_param(kprn_handle* h, const char* name, float* dst, int64_t n) { return copy_named(h, name, dst, nullptr, n, 0); }
int kprn_set_param(kprn_handle* h, const char* name, const float* src, int64_t n) { return copy_named(h, name, nullptr, src, n, 0); }
int kprn_get_grad(kprn_handle* h, const char* name, float* dst, int64_t n) { return copy_named(h, name, dst, nullptr, n, 1); }

// data-parallel exchange with the dense gradient arena riding in the packed row buffer (kprn_set_option "dp_dense_in_pack"): one collective
// per step instead of two.  The arena is copied behind the rows; after the all-gather every rank sums the W copies IN RANK ORDER (the same
// sequence of additions everywhere: bit-identical dense gradients on every replica by construction).
__global__ void k_dense_to_pack(const float* __restrict__ g, int64_t n, float* __restrict__ tail) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) tail[i] = g[i];
}
__global__ void k_dense_from_all(const float* __restrict__ all, int world, int64_t stride_words, int64_t tail_off, int64_t n, float* __restrict__ g) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float acc = all[tail_off + i];
    for (int r = 1; r < world; ++r) acc += all[(int64_t)r * stride_words + tail_off + i];
    g[i] = acc;
  }
}

// rows of one parameter tensor by 0-based row index (a 20 M-row entity table is 10 GB: reading the rows a batch touched must not copy it)
__global__ void k_rows_copy(float* __restrict__ W, const int64_t* __restrict__ rows, int64_t n, int64_t cols, float* __restrict__ buf, int to_table) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * cols; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = rows[i / cols], c = i % cols;
    if (to_table) W[r * cols + c] = buf[i]; else buf[i] = W[r * cols + c];
  }
}
static int copy_rows(kprn_handle* h, const char* name, const int64_t* rows, int64_t n_rows, float* dst, const float* src) {
  API_BEGIN(h)
  const ParamInfo* p = find_param(h, name);
  KPRN_REQUIRE(p, KPRN_E_ARG, std::string("unknown parameter name: ") + (name ? name : "(null)"));
  KPRN_REQUIRE(n_rows >= 0 && (rows || n_rows == 0) && (dst || src || n_rows == 0), KPRN_E_ARG, "NULL buffer");
  if (n_rows == 0 || p->rows * p->cols == 0) return KPRN_OK;
  for (int64_t i = 0; i < n_rows; ++i) KPRN_REQUIRE(rows[i] >= 0 && rows[i] < p->rows, KPRN_E_INDEX, "row index outside the tensor");
  if (p->where == 1) flush_lazy(h);
  if (src) params_touched(h);   // (BEFORE the write is queued: it makes the main stream wait for a scoring pass that may still read these rows)
  float* base = (p->where == 1 ? h->We : h->dense) + p->dev_off;
  int64_t* d_rows = nullptr;
  float* d_buf = nullptr;
  try {
    d_rows = dalloc<int64_t>(n_rows);
    d_buf = dalloc<float>(n_rows * p->cols);
    HIP_TRY(hipMemcpyAsync(d_rows, rows, (size_t)n_rows * sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
    if (src) HIP_TRY(hipMemcpyAsync(d_buf, src, (size_t)(n_rows * p->cols) * sizeof(float), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_rows_copy, dim3((unsigned)std::min<int64_t>((n_rows * p->cols + 255) / 256, 8192)), dim3(256), 0, h->stream, base, d_rows, n_rows, p->cols, d_buf,
                       src ? 1 : 0);
    if (dst) HIP_TRY(hipMemcpyAsync(dst, d_buf, (size_t)(n_rows * p->cols) * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  } catch (...) {
    hipStreamSynchronize(h->stream);
    dfree(d_rows); dfree(d_buf);
    throw;
  }
  dfree(d_rows); dfree(d_buf);
  API_END(h)
}
int kprn_get_param_rows(kprn_handle* h, const char* name, const int64_t* rows, int64_t n_rows, float* dst) { return copy_rows(h, name, rows, n_rows, dst, nullptr); }
int kprn_set_param_rows(kprn_handle* h, const char* name, const int64_t* rows, int64_t n_rows, const float* src) { return copy_rows(h, name, rows, n_rows, nullptr, src); }

static int copy_flat(kprn_handle* h, float* dst, const float* src, int64_t n, int which /*0 param,1 grad,2 s1,3 s2*/) {
  API_BEGIN(h)
  KPRN_REQUIRE(n == h->n_params, KPRN_E_ARG, "n must equal kprn_num_params");
  KPRN_REQUIRE(dst || src, KPRN_E_ARG, "NULL buffer");
  if (which != 1) flush_lazy(h);
  else materialize_union(h);
  for (auto& p : h->params) {
    float* base;
    switch (which) {
      case 0: base = (p.where == 1 ? h->We : h->dense); break;
      case 1: base = (p.where == 1 ? h->g_We : h->g_dense); break;
      case 2: base = (p.where == 1 ? h->s1_We : h->s1_dense); break;
      default: base = (p.where == 1 ? h->s2_We : h->s2_dense); break;
    }
    base += p.dev_off;
    const size_t bytes = (size_t)(p.rows * p.cols) * sizeof(float);
    if (dst) HIP_TRY(hipMemcpyAsync(dst + p.flat_off, base, bytes, hipMemcpyDeviceToHost, h->stream));
    else HIP_TRY(hipMemcpyAsync(base, src + p.flat_off, bytes, hipMemcpyHostToDevice, h->stream));
  }
  if (!dst) params_touched(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  API_END(h)
}
int kprn_get_flat_params(kprn_handle* h, float* dst, int64_t n) { return copy_flat(h, dst, nullptr, n, 0); }
int kprn_set_flat_params(kprn_handle* h, const float* src, int64_t n) { return copy_flat(h, nullptr, src, n, 0); }
int kprn_get_flat_grads(kprn_handle* h, float* dst, int64_t n) { return copy_flat(h, dst, nullptr, n, 1); }
int kprn_get_flat_opt_state(kprn_handle* h, int32_t slot, float* dst, int64_t n) {
  if (slot != 0 && slot != 1) { if (h) h->err = "slot must be 0 or 1"; return KPRN_E_ARG; }
  return copy_flat(h, dst, nullptr, n, 2 + slot);
}

int kprn_zero_pad_tokens(kprn_handle* h) {
  API_BEGIN(h)
  // a no-op while the pad rows are known to be zero (the optimiser re-zeroes them at the end of every step, MyOptimizer.lua:219): the
  // data-parallel step calls this before every backward, and zeroing again made the main stream wait for the scoring pass on the side
  // stream and invalidated the fused kernels' derived weight copies -- 0.3 ms of a 1.5 ms step
  if (h->pad_clean) return KPRN_OK;
  zero_pad_tokens(h);
  fused::params_changed(h);
  bf16p::params_changed(h, false);
  API_END(h)
}

// ---- batches ---------------------------------------------------------------------------------------------------------------
// A batch is a slot of device buffers (ids, labels, occurrence index, identical-prefix plan).  kprn_batch_create fills a new slot
// on the handle's stream and returns when it is ready; kprn_batch_feed_async refills a slot on the FEED stream and returns at
// once -- the role BatcherFileList:populateGPUTensor plays for the reference (preallocated tensors, one :copy per minibatch,
// BatcherFileList.lua:53-96), plus the per-batch device work this engine adds (validation, index, plan).
// All device arrays of a slot live in ONE allocation, laid out afresh for every fill from the batch's own sizes:
//   idx | idx_s | perm | slot_of | tile_k | pmeta | key_sorted | pos_sorted | uniq ... count | labels | flag
// (each rounded to 16 bytes).  The host-built feed prepares a page-locked image of exactly this block and uploads it with a
// single copy.
struct BatchLayout { int64_t idx, idx_s, perm, slot_of, tile_k, pmeta, key, pos, uniq, cnt, labels, flag, words; };
static BatchLayout batch_layout(int64_t B, int64_t N, int T, int F, bool plan, bool labels) {
  auto r4 = [](int64_t v) { return (v + 3) & ~(int64_t)3; };
  const int64_t nsteps = N * T, n_index = nsteps + (plan ? fused::KCAP : 0);
  BatchLayout l;
  int64_t o = 0;
  l.idx = o; o += r4(nsteps * F);
  l.idx_s = o; o += plan ? r4(nsteps * F) : 0;
  l.perm = o; o += plan ? r4(N) : 0;
  l.slot_of = o; o += plan ? r4(N) : 0;
  l.tile_k = o; o += plan ? r4((N + 63) / 64 + 1) : 0;
  l.pmeta = o; o += plan ? 24 : 0;
  l.key = o; o += r4(n_index);
  l.pos = o; o += r4(n_index);
  l.uniq = o; o += r4(n_index);
  l.cnt = o; o += 4;
  l.labels = o; o += labels ? r4(B) : 0;
  l.flag = o; o += 4;
  l.words = o;
  return l;
}

static void batch_free_buffers(kprn_batch* b) {
  dfree(b->block);
  b->block_cap = 0;
  b->idx = b->idx_s = b->perm = b->slot_of = b->tile_k = b->pmeta = b->key_sorted = b->pos_sorted = b->uniq = b->d_flag = nullptr;
  b->labels = nullptr;
}

static void batch_release(kprn_batch* b) {
  if (b->job.valid()) { try { b->job.get(); } catch (...) {} }
  if (b->hs) { hipHostFree(b->hs); b->hs = nullptr; }
  if (b->ev_fork) { hipEventDestroy(b->ev_fork); hipEventDestroy(b->ev_fork2); b->ev_fork = b->ev_fork2 = nullptr; }
  batch_free_buffers(b);
  if (b->h_meta) { hipHostFree(b->h_meta); b->h_meta = nullptr; }
  if (b->ev_ready) { hipEventDestroy(b->ev_ready); b->ev_ready = nullptr; }
  delete b;
}

static bool batch_wants_plan(kprn_handle* h, const kprn_batch* b) {
  // (small batches run on tiles of one 16-row m-tile, which have no per-tile prefix classes: lstm_fused_fwd.hip small_tiles)
  if (fused::small_tiles(h, (int64_t)b->B * b->P, false)) return false;
  return h->prefix_plan && use_fused(h, b, true) && b->F <= 16 && !(kprn_dbg_mask() & 64);
}

// buffers for a [B,P,T,F] batch; a refill that fits the slot's capacities allocates nothing.  quiesce(): called before any
// buffer of a slot in use is freed.
static void batch_reserve(kprn_handle* h, kprn_batch* b, int32_t B, int32_t P, int32_t T, int32_t F, bool labels, const std::function<void()>& quiesce,
                          int64_t min_pairs = 0, int64_t min_paths = 0) {
  b->B = B; b->P = P; b->T = T; b->F = F;
  const int64_t nsteps = (int64_t)B * P * T, N = (int64_t)B * P;
  const bool plan = batch_wants_plan(h, b);
  b->kcap = plan ? fused::KCAP : 0;
  b->n_index = nsteps + b->kcap;
  // the allocation only grows (a slot that has held the largest minibatch never allocates again; kprn_batch_slot_reserve sizes
  // it up front, for the larger of the two layouts)
  const BatchLayout l = batch_layout(B, N, T, F, plan, labels);
  int64_t want = l.words;
  if (min_pairs > 0 || min_paths > 0)
    want = std::max(want, batch_layout(std::max<int64_t>(B, min_pairs), std::max<int64_t>(N, min_paths), T, F, true, true).words);
  if (want > b->block_cap) {
    if (b->block) { quiesce(); batch_free_buffers(b); }
    b->block = dalloc<int32_t>(want);
    b->block_cap = want;
  }
  int32_t* k = b->block;
  b->idx = k + l.idx;
  b->idx_s = plan ? k + l.idx_s : nullptr; b->perm = plan ? k + l.perm : nullptr; b->slot_of = plan ? k + l.slot_of : nullptr;
  b->tile_k = plan ? k + l.tile_k : nullptr; b->pmeta = plan ? k + l.pmeta : nullptr;
  b->key_sorted = k + l.key; b->pos_sorted = k + l.pos; b->uniq = k + l.uniq;
  b->uniq_cap = l.cnt - l.uniq;  // the distinct-row count lives at uniq[uniq_cap]
  b->labels = labels ? (float*)(k + l.labels) : nullptr;
  b->d_flag = k + l.flag;
  const int64_t meta_need = 2 + 8 + 16 + (std::max<int64_t>(N, min_paths) + 63) / 64 + 1;
  if (meta_need > b->h_meta_cap) {
    if (b->h_meta) { quiesce(); hipHostFree(b->h_meta); b->h_meta = nullptr; }
    HIP_TRY(hipHostMalloc((void**)&b->h_meta, (size_t)meta_need * 2 * sizeof(int32_t)));
    b->h_meta_cap = meta_need * 2;
  }
}

static void scratch_reserve(void** scratch, size_t* bytes, size_t need) {
  if (need <= *bytes) return;
  if (*scratch) hipFree(*scratch);
  *scratch = nullptr; *bytes = 0;
  HIP_TRY(kprn_dev_malloc(scratch, need * 2));
  *bytes = need * 2;
}

// upload + validation + identical-prefix plan + occurrence index on stream s; the host-side summary (validation flag, distinct
// rows, plan header, per-tile prefix lengths) lands in the slot's pinned block behind them.  Nothing here waits for the device.
static void batch_enqueue(kprn_handle* h, kprn_batch* b, const int32_t* idx, const float* labels, hipStream_t s, void* scratch, size_t scratch_bytes) {
  const int32_t B = b->B, P = b->P, T = b->T, F = b->F;
  const int64_t nsteps = (int64_t)B * P * T, N = (int64_t)B * P;
  const bool plan = b->kcap > 0;
  HIP_TRY(hipMemcpyAsync(b->idx, idx, (size_t)nsteps * F * sizeof(int32_t), hipMemcpyHostToDevice, s));
  if (labels) HIP_TRY(hipMemcpyAsync(b->labels, labels, (size_t)B * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(b->d_flag, 0, sizeof(int32_t), s));
  kk::validate_indices(s, b->idx, nsteps, F, h->cfg.num_types, h->cfg.Vt, h->cfg.Ve, h->cfg.Vr, b->d_flag);
  // identical-prefix plan (fused path only): paths reordered by the number of leading steps they share with the batch's
  // reference step; the fused kernels start each 64-path tile behind its shared steps (lstm_fused_prefix.hip)
  if (plan)
    bidx::prefix_plan(s, b->idx, N, T, F, h->cfg.num_types, b->kcap, b->idx_s, b->perm, b->slot_of, b->tile_k, b->pmeta, scratch, scratch_bytes);
  // occurrence index: positions sorted by entity row + the sorted distinct rows (count at the tail of the list)
  bidx::build(s, plan ? b->idx_s : b->idx, N, T, F, h->cfg.Ve, plan ? b->tile_k : nullptr, plan ? b->pmeta : nullptr, b->kcap, b->key_sorted,
              b->pos_sorted, b->uniq, b->uniq + b->uniq_cap, scratch, scratch_bytes);
  int32_t* m = b->h_meta;
  HIP_TRY(hipMemcpyAsync(m, b->d_flag, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(m + 1, b->uniq + b->uniq_cap, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (plan) {
    HIP_TRY(hipMemcpyAsync(m + 2, b->pmeta, (size_t)(8 + F) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(m + 2 + 8 + 16, b->tile_k, (size_t)((N + 63) / 64) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  }
}

// the device work of batch_enqueue is complete: take the host-side summary over
static void batch_finish(kprn_handle* h, kprn_batch* b) {
  const int32_t* m = b->h_meta;
  const int64_t nsteps = (int64_t)b->B * b->P * b->T, N = (int64_t)b->B * b->P;
  b->pending = false;
  b->n_uniq = m[1];
  b->h_kmax = 0;
  b->exec_steps = nsteps;
  if (b->kcap > 0) {
    b->h_kmax = m[2];
    for (int c = 0; c < b->F && c < 16; ++c) b->h_ref[c] = m[2 + 8 + c];
    const int32_t* tk = m + 2 + 8 + 16;
    for (int64_t tl = 0; tl < (N + 63) / 64; ++tl) b->exec_steps -= (int64_t)tk[tl] * std::min<int64_t>(64, N - tl * 64);
  }
  b->serial = h->next_serial++;
  b->bad = (m[0] != 0);
  KPRN_REQUIRE(!b->bad, KPRN_E_INDEX, "an index is outside 1..vocabSize (ids are 1-based, int2torch.lua:60-63)");
}

// first use of a slot filled by kprn_batch_feed_async: wait for its feed (issued a step earlier), read the summary
static void batch_ready(kprn_handle* h, const kprn_batch* cb) {
  kprn_batch* b = const_cast<kprn_batch*>(cb);
  if (!b) return;
  if (b->pending && b->host_built) {
    // the worker thread has derived plan + index and queued the uploads: take its summary, order this stream behind the uploads
    b->pending = false;
    b->job.get();  // (rethrows what the job threw)
    const kprn_batch::HostResult& r = b->hres;
    b->bad = r.bad; b->n_uniq = r.n_uniq; b->h_kmax = r.kmax; b->exec_steps = r.exec_steps;
    for (int c = 0; c < 16; ++c) b->h_ref[c] = r.ref[c];
    b->serial = h->next_serial++;
    if (!b->bad) HIP_TRY(hipStreamWaitEvent(h->stream, b->ev_ready, 0));
  } else if (b->pending) {
    HIP_TRY(hipEventSynchronize(b->ev_ready));
    batch_finish(h, b);
  }
  KPRN_REQUIRE(!b->bad, KPRN_E_INDEX, "an index is outside 1..vocabSize (ids are 1-based, int2torch.lua:60-63)");
}

static void check_batch_args(kprn_handle* h, const int32_t* idx, int32_t B, int32_t P, int32_t T, int32_t F) {
  KPRN_REQUIRE(idx, KPRN_E_ARG, "idx is NULL");
  KPRN_REQUIRE(B > 0 && P > 0 && T > 0, KPRN_E_ARG, "B, P, T must be positive");
  KPRN_REQUIRE(F == h->cfg.F, KPRN_E_ARG, "F does not match numFeatureTemplates");
}

int kprn_batch_create(kprn_handle* h, const int32_t* idx, const float* labels, int32_t B, int32_t P, int32_t T, int32_t F, kprn_batch** out) {
  API_BEGIN(h)
  KPRN_REQUIRE(out, KPRN_E_ARG, "out is NULL");
  *out = nullptr;
  check_batch_args(h, idx, B, P, T, F);
  kprn_batch* b = new kprn_batch();
  try {
    batch_reserve(h, b, B, P, T, F, labels != nullptr, [] {});
    b->has_index = true; b->idx_valid = true;
    const int64_t N = (int64_t)B * P;
    scratch_reserve(&h->bidx_scratch, &h->bidx_scratch_bytes, std::max(bidx::scratch_bytes(b->n_index, h->cfg.Ve), bidx::prefix_scratch_bytes(N, fused::KCAP)));
    batch_enqueue(h, b, idx, labels, h->stream, h->bidx_scratch, h->bidx_scratch_bytes);
    HIP_TRY(hipStreamSynchronize(h->stream));
    batch_finish(h, b);
  } catch (...) {
    batch_release(b);
    throw;
  }
  *out = b;
  API_END(h)
}

// host-built feed: a worker thread runs hostfeed::build into a page-locked image of the slot's device block; ONE upload thread
// then moves each image with ONE copy, behind the positions the main / scoring streams had when the refill was requested, and
// keeps a single copy in flight (several streams' worth of small concurrent copies fell back from the DMA engines to copy
// kernels, which take CUs from the persistent kernels: measured, profiles/r02)
// inline_now (the host-buffer entry points kprn_train_step / kprn_forward, which return results and therefore wait for the feed anyway): derive on the
// CALLING thread and copy on the engine's own stream -- no worker hand-over, no upload thread, no cross-stream events (a 128-pair minibatch is
// microseconds of host work; the thread hand-overs were most of its feed time)
// The upload stream, made once -- by whichever feed path needs it first (the inline side upload of kprn_train_step or the worker-built feed; a second
// creation would drop the first handle with copies still queued on it, outside quiesce() and kprn_destroy's reach).  A queue of its own: HIP
// multiplexes the streams of one priority onto a few hardware queues, and this stream spends its life waiting on events of the compute streams --
// sharing a hardware queue with one of them stalls that stream's kernels behind the waits (measured: every kernel of the step 1.2-5x slower).  The
// high-priority class has its own queues.
static void ensure_upload_stream(kprn_handle* h) {
  if (h->upload_stream) return;
  int lo = 0, hi = 0;
  HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
  HIP_TRY(hipStreamCreateWithPriority(&h->upload_stream, hipStreamNonBlocking, hi));
}

static void feed_host(kprn_handle* h, kprn_batch* b, const int32_t* idx, const float* labels, const int64_t* rows, bool inline_now = false) {
  const int32_t B = b->B, P = b->P, T = b->T, F = b->F;
  const int64_t nsteps = (int64_t)B * P * T, N = (int64_t)B * P, n_index = b->n_index;
  if (!h->feed_pool && !inline_now) {
    if (h->feed_workers <= 0) {  // defaults from the machine: a GPU host has cores to spare, a small container does not
      const unsigned hc = std::thread::hardware_concurrency();
      h->feed_workers = hc >= 32 ? 4 : 2;
      if (h->feed_threads <= 0) h->feed_threads = hc >= 64 ? 8 : (hc >= 16 ? 4 : 2);
    }
    if (h->feed_threads <= 0) h->feed_threads = 4;
    h->feed_pool = hostfeed::make_pool(std::max(1, h->feed_workers));
    h->upload_pool = hostfeed::make_pool(1);
    ensure_upload_stream(h);
  }
  if (!b->ev_fork) {
    HIP_TRY(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&b->ev_fork2, hipEventDisableTiming));
  }
  const bool plan = b->kcap > 0;
  const BatchLayout l = batch_layout(B, N, T, F, plan, labels != nullptr);
  // a label-less batch is scored only: with no lazy row update pending nothing walks its entity rows -> no occurrence index is
  // built, and with a plan the ids in their original order are not uploaded either (a third of the bytes, none of the sorting)
  const bool want_index = labels != nullptr || h->lazy_pending;
  const bool want_idx = want_index || !plan;
  b->has_index = want_index; b->idx_valid = want_idx;
  if (b->block_cap > b->hs_cap) {  // the image is as large as the block: a reserved slot allocates it once
    if (b->hs) hipHostFree(b->hs);
    b->hs = nullptr; b->hs_cap = 0;
    HIP_TRY(hipHostMalloc((void**)&b->hs, (size_t)b->block_cap * sizeof(int32_t)));
    b->hs_cap = b->block_cap;
  }
  if ((int64_t)b->hw.size() < 4 * n_index) b->hw.resize((size_t)(4 * std::max(n_index, (b->block_cap / 8))));
  const hostfeed::Shape g{B, P, T, F, h->cfg.num_types, h->cfg.Vt, h->cfg.Ve, h->cfg.Vr};
  const int kcap = b->kcap, dev = h->cfg.device_id;
  int32_t* hs = b->hs;
  int32_t* hw = b->hw.data();
  if (labels && rows) { float* hl = (float*)(hs + l.labels); for (int32_t i = 0; i < B; ++i) hl[i] = labels[rows[i]]; }
  else if (labels) memcpy(hs + l.labels, labels, (size_t)B * sizeof(float));
  hs[l.flag] = 0;
  b->host_built = true;
  auto done = std::make_shared<std::promise<void>>();
  b->job = done->get_future();
  if (inline_now) {
    try {
      // where the upload runs: beside the step in flight when nothing can still read this slot (kprn_internal.h: dropin_prev_waited), else in stream order
      const bool side_copy = h->inline_upload_side && labels != nullptr && h->dropin_prev_waited_now && h->inline_side_ok;
      hipStream_t cs = h->stream;
      kprn_batch::HostResult* r = &b->hres;
      const int32_t* src = idx;
      const int nth1 = nsteps < 65536 ? 1 : std::max(1, h->feed_threads > 0 ? h->feed_threads : 4);
      if (rows) { hostfeed::gather_rows(hs + l.idx, idx, (int64_t)P * T * F, rows, B, nth1); src = hs + l.idx; }
      hostfeed::build(g, src, kcap, nth1, want_index, r, hs + l.idx_s, hs + l.perm, hs + l.slot_of, hs + l.tile_k, hs + l.pmeta, hs + l.key, hs + l.pos,
                      hs + l.uniq, hw, hw + n_index, hw + 2 * n_index, hw + 3 * n_index);
      if (!r->bad) {
        if (want_idx && !rows) memcpy(hs + l.idx, idx, (size_t)(nsteps * F) * sizeof(int32_t));
        hs[l.cnt] = r->n_uniq;
        if (side_copy) ensure_upload_stream(h);
        cs = side_copy ? h->upload_stream : h->stream;
        if (h->score_pending && h->score_stream) HIP_TRY(hipStreamWaitEvent(cs, h->ev_score_done, 0));   // (a pass on the side stream may still read the slot)
        const int64_t w0 = want_idx ? 0 : l.idx_s, w1 = want_index ? l.words : l.key;
        HIP_TRY(hipMemcpyAsync(b->block + w0, hs + w0, (size_t)(w1 - w0) * sizeof(int32_t), hipMemcpyHostToDevice, cs));
      }
      HIP_TRY(hipEventRecord(b->ev_ready, cs));   // (batch_ready orders the engine's stream behind it)
      done->set_value();
    } catch (...) { done->set_exception(std::current_exception()); }
    return;
  }
  const int nth = std::max(1, h->feed_threads);
  HIP_TRY(hipEventRecord(b->ev_fork, h->stream));
  const bool wait_score = h->score_pending && h->score_stream;
  if (wait_score) HIP_TRY(hipEventRecord(b->ev_fork2, h->score_stream));
  hostfeed::Pool* up_pool = (hostfeed::Pool*)h->upload_pool;
  hipStream_t us = h->upload_stream;
  hostfeed::submit((hostfeed::Pool*)h->feed_pool, [=]() {
    try {
      kprn_batch::HostResult* r = &b->hres;
      const int32_t* src = idx;
      if (rows) {   // pair i of the minibatch = row rows[i] of the file's array: gathered straight into the image
        hostfeed::gather_rows(hs + l.idx, idx, (int64_t)P * T * F, rows, B, nth);
        src = hs + l.idx;
      }
      hostfeed::build(g, src, kcap, nth, want_index, r, hs + l.idx_s, hs + l.perm, hs + l.slot_of, hs + l.tile_k, hs + l.pmeta, hs + l.key, hs + l.pos,
                      hs + l.uniq, hw, hw + n_index, hw + 2 * n_index, hw + 3 * n_index);
      if (!r->bad) {
        if (want_idx && !rows) memcpy(hs + l.idx, idx, (size_t)(nsteps * F) * sizeof(int32_t));
        hs[l.cnt] = r->n_uniq;
      }
    } catch (...) { done->set_exception(std::current_exception()); return; }
    hostfeed::submit(up_pool, [=]() {
      bool issued = false;
      try {
        HIP_TRY(hipSetDevice(dev));
        if (!b->hres.bad) {
          HIP_TRY(hipStreamWaitEvent(us, b->ev_fork, 0));
          if (wait_score) HIP_TRY(hipStreamWaitEvent(us, b->ev_fork2, 0));
          // the image is contiguous: idx | idx_s .. pmeta | key | pos | uniq | count | labels | flag -- a scoring-only batch moves the plan part
          const int64_t w0 = want_idx ? 0 : l.idx_s, w1 = want_index ? l.words : l.key;
          HIP_TRY(hipMemcpyAsync(b->block + w0, hs + w0, (size_t)(w1 - w0) * sizeof(int32_t), hipMemcpyHostToDevice, us));
        }
        HIP_TRY(hipEventRecord(b->ev_ready, us));
        issued = true;
        done->set_value();
        HIP_TRY(hipStreamSynchronize(us));  // one image in flight at a time
      } catch (...) { if (!issued) done->set_exception(std::current_exception()); }
    });
  });
}

static void feed_impl(kprn_handle* h, kprn_batch** slot, const int32_t* idx, const float* labels, const int64_t* rows, int32_t B, int32_t P, int32_t T, int32_t F,
                      bool inline_now = false) {
  KPRN_REQUIRE(slot, KPRN_E_ARG, "slot is NULL");
  check_batch_args(h, idx, B, P, T, F);
  if (!h->feed_build_host && !h->feed_stream) {
    // device-built feed: its kernels are small and latency-bound; they get the CUs the persistent kernels leave idle in their
    // tails.  (Created only when used: HIP multiplexes streams onto a few hardware queues, and a stream that waits on events --
    // as the feed's do -- blocks whatever shares its queue.)
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(hipStreamCreateWithPriority(&h->feed_stream, hipStreamNonBlocking, hi));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_feed_fork, hipEventDisableTiming));
  }
  kprn_batch* b = *slot;
  const bool fresh = (b == nullptr);
  if (fresh) b = new kprn_batch();
  h->inline_side_ok = !fresh && h->score_rest_batch != b && h->view_batch != b;   // (a fresh slot allocates: in stream order)
  try {
    if (!fresh) {
      if (h->score_rest_batch == b) launch_score_rest(h);   // (the deferred part of a split scoring pass still reads the slot's old contents)
      // the slot's previous contents: whatever the handle still refers to goes to owned storage; their last readers were
      // enqueued before this call, and the feed stream starts behind them (below)
      if (h->view_batch == b) {
        if (h->ent_grads_dirty) materialize_step_rows(h);  // gradients of a backward without an update still name these rows
        else { h->view_batch = nullptr; h->rows_view = h->step_rows; h->count_view = h->step_count; h->step_rows_ub = 0; }
      }
      if (h->caught_serial == b->serial) h->caught_serial = -1;
      if (b->pending && b->host_built) { b->job.get(); b->pending = false; }
      // (also when the slot was used: its uploads read the page-locked staging the next fill overwrites -- this is the feed's
      //  back-pressure on a host that runs ahead of the device)
      if (b->ev_ready && (b->pending || b->host_built)) HIP_TRY(hipEventSynchronize(b->ev_ready));
      b->pending = false;
    }
    auto quiesce = [&] {
      if (h->score_stream) HIP_TRY(hipStreamSynchronize(h->score_stream));
      if (h->rest_stream) HIP_TRY(hipStreamSynchronize(h->rest_stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (h->feed_stream) HIP_TRY(hipStreamSynchronize(h->feed_stream));
      if (h->upload_stream) HIP_TRY(hipStreamSynchronize(h->upload_stream));
    };
    batch_reserve(h, b, B, P, T, F, labels != nullptr, quiesce);
    if (!b->ev_ready) HIP_TRY(hipEventCreateWithFlags(&b->ev_ready, hipEventDisableTiming));
    if (h->feed_build_host) {
      feed_host(h, b, idx, labels, rows, inline_now);
    } else {
      b->host_built = false; b->has_index = true; b->idx_valid = true;
      const int64_t N = (int64_t)B * P;
      const size_t need = std::max(bidx::scratch_bytes(b->n_index, h->cfg.Ve), bidx::prefix_scratch_bytes(N, fused::KCAP));
      if (need > h->feed_scratch_bytes) {
        HIP_TRY(hipStreamSynchronize(h->feed_stream));
        scratch_reserve(&h->feed_scratch, &h->feed_scratch_bytes, need);
      }
      // everything enqueued on the handle so far (the last readers of this slot among it) comes first
      HIP_TRY(hipEventRecord(h->ev_feed_fork, h->stream));
      HIP_TRY(hipStreamWaitEvent(h->feed_stream, h->ev_feed_fork, 0));
      if (h->score_pending) HIP_TRY(hipStreamWaitEvent(h->feed_stream, h->ev_score_done, 0));
      const int32_t* src = idx;
      const float* lsrc = labels;
      if (rows) {   // device build: the rows are gathered by the calling thread into the slot's host scratch first
        const int64_t rw = (int64_t)P * T * F;
        if ((int64_t)b->hw.size() < (int64_t)B * rw + B) b->hw.resize((size_t)((int64_t)B * rw + B));
        hostfeed::gather_rows(b->hw.data(), idx, rw, rows, B, std::max(1, h->feed_threads));
        src = b->hw.data();
        if (labels) { float* hl = (float*)(b->hw.data() + (int64_t)B * rw); for (int32_t i = 0; i < B; ++i) hl[i] = labels[rows[i]]; lsrc = hl; }
      }
      batch_enqueue(h, b, src, lsrc, h->feed_stream, h->feed_scratch, h->feed_scratch_bytes);
      HIP_TRY(hipEventRecord(b->ev_ready, h->feed_stream));
    }
    b->pending = true;
  } catch (...) {
    if (fresh) batch_release(b);
    throw;
  }
  *slot = b;
}

int kprn_batch_feed_async(kprn_handle* h, kprn_batch** slot, const int32_t* idx, const float* labels, int32_t B, int32_t P, int32_t T, int32_t F) {
  API_BEGIN(h)
  feed_impl(h, slot, idx, labels, nullptr, B, P, T, F);
  API_END(h)
}

int kprn_batch_feed_rows_async(kprn_handle* h, kprn_batch** slot, const int32_t* data, const float* labels, int64_t n_rows, const int64_t* rows, int32_t B,
                               int32_t P, int32_t T, int32_t F) {
  API_BEGIN(h)
  KPRN_REQUIRE(rows, KPRN_E_ARG, "rows is NULL");
  for (int32_t i = 0; i < B; ++i) KPRN_REQUIRE(rows[i] >= 0 && rows[i] < n_rows, KPRN_E_ARG, "a row index is outside 0..n_rows-1");
  feed_impl(h, slot, data, labels, rows, B, P, T, F);
  API_END(h)
}

int kprn_batch_slot_reserve(kprn_handle* h, kprn_batch** slot, int32_t max_pairs, int64_t max_paths, int32_t T, int32_t F, int32_t with_labels) {
  API_BEGIN(h)
  KPRN_REQUIRE(slot && max_pairs > 0 && max_paths >= max_pairs && T > 0, KPRN_E_ARG, "bad argument");
  KPRN_REQUIRE(F == h->cfg.F, KPRN_E_ARG, "F does not match numFeatureTemplates");
  kprn_batch* b = *slot;
  const bool fresh = (b == nullptr);
  if (fresh) b = new kprn_batch();
  try {
    if (!fresh) {
      if (h->score_rest_batch == b) launch_score_rest(h);
      if (h->view_batch == b) materialize_step_rows(h);
      if (b->pending && b->host_built && b->job.valid()) { try { b->job.get(); } catch (...) {} }
      b->pending = false; b->bad = true;  // (no contents until the next feed)
    }
    auto quiesce = [&] {
      if (h->score_stream) HIP_TRY(hipStreamSynchronize(h->score_stream));
      if (h->rest_stream) HIP_TRY(hipStreamSynchronize(h->rest_stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      if (h->feed_stream) HIP_TRY(hipStreamSynchronize(h->feed_stream));
      if (h->upload_stream) HIP_TRY(hipStreamSynchronize(h->upload_stream));
    };
    const int32_t P1 = (int32_t)std::max<int64_t>(1, max_paths / max_pairs);
    batch_reserve(h, b, max_pairs, P1, T, F, with_labels != 0, quiesce, max_pairs, max_paths);
    if (fresh) b->bad = true;
  } catch (...) {
    if (fresh) batch_release(b);
    throw;
  }
  *slot = b;
  API_END(h)
}

int kprn_host_batch_index(const int32_t* idx, int32_t B, int32_t P, int32_t T, int32_t F, int32_t num_types, int32_t Vt, int32_t Ve, int32_t Vr,
                          int32_t plan, int32_t threads, int32_t* idx_s, int32_t* perm, int32_t* slot_of, int32_t* tile_k, int32_t* pmeta,
                          int32_t* key_sorted, int32_t* pos_sorted, int32_t* uniq, int64_t* summary) {
  if (!idx || B <= 0 || P <= 0 || T <= 0 || F < num_types + 2 || num_types < 1 || !key_sorted || !pos_sorted || !uniq || !summary) return KPRN_E_ARG;
  if (plan && (T < 2 || F > 16 || !idx_s || !perm || !slot_of || !tile_k || !pmeta)) return KPRN_E_ARG;
  try {
    const int kcap = plan ? fused::KCAP : 0;
    const int64_t n_index = (int64_t)B * P * T + kcap;
    std::vector<int32_t> w((size_t)(4 * n_index));
    kprn_batch::HostResult r;
    const hostfeed::Shape g{B, P, T, F, num_types, Vt, Ve, Vr};
    hostfeed::build(g, idx, kcap, std::max(1, (int)threads), true, &r, idx_s, perm, slot_of, tile_k, pmeta, key_sorted, pos_sorted, uniq, w.data(),
                    w.data() + n_index, w.data() + 2 * n_index, w.data() + 3 * n_index);
    summary[0] = r.bad ? 1 : 0; summary[1] = r.kmax; summary[2] = r.n_uniq; summary[3] = r.exec_steps;
  } catch (...) { return KPRN_E_NOMEM; }
  return KPRN_OK;
}

int kprn_host_alloc(kprn_handle* h, size_t bytes, void** out) {
  API_BEGIN(h)
  KPRN_REQUIRE(out && bytes > 0, KPRN_E_ARG, "bad argument");
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, bytes);
  if (e != hipSuccess) throw KprnError{KPRN_E_NOMEM, std::string("hipHostMalloc failed: ") + hipGetErrorString(e)};
  API_END(h)
}

int kprn_host_free(kprn_handle* h, void* p) {
  API_BEGIN(h)
  if (p) HIP_TRY(hipHostFree(p));
  API_END(h)
}

void kprn_batch_destroy(kprn_handle* h, kprn_batch* b) {
  if (!b) return;
  if (h) {
    hipSetDevice(h->cfg.device_id);
    if (h->view_batch == b) { try { materialize_step_rows(h); } catch (...) { h->view_batch = nullptr; h->rows_view = nullptr; h->step_rows_ub = 0; } }
    if (h->caught_serial == b->serial) h->caught_serial = -1;
    if (b->job.valid()) { try { b->job.get(); } catch (...) {} }
    if (h->upload_stream) hipStreamSynchronize(h->upload_stream);
    if (h->feed_stream) hipStreamSynchronize(h->feed_stream);
    if (h->score_rest_batch == b) { try { launch_score_rest(h); } catch (...) { h->score_rest_batch = nullptr; } }   // (the deferred part of a split pass reads the batch)
    if (h->pool_defer_batch == b) h->pool_defer_batch = nullptr;   // (a step that failed between its forward and its loss stage)
    if (h->score_stream) hipStreamSynchronize(h->score_stream);
  if (h->rest_stream) hipStreamSynchronize(h->rest_stream);  // a scoring pass on the second stream may still read the batch
    hipStreamSynchronize(h->stream);
  }
  batch_release(b);
}

int kprn_batch_distinct_rows(kprn_handle* h, const kprn_batch* b, int32_t* n) {
  API_BEGIN(h)
  KPRN_REQUIRE(b && n, KPRN_E_ARG, "NULL argument");
  batch_ready(h, b);
  *n = b->n_uniq;
  API_END(h)
}

int kprn_batch_executed_steps(kprn_handle* h, const kprn_batch* b, int64_t* steps) {
  API_BEGIN(h)
  KPRN_REQUIRE(b && steps, KPRN_E_ARG, "NULL argument");
  batch_ready(h, b);
  *steps = b->exec_steps;
  API_END(h)
}

int kprn_batch_handover_stats(kprn_handle* h, const kprn_batch* b, int64_t* out) {
  API_BEGIN(h)
  KPRN_REQUIRE(b && out, KPRN_E_ARG, "NULL argument");
  batch_ready(h, b);
  out[0] = out[1] = out[2] = out[3] = 0;
  if (use_fused(h, b, true) && h->cfg.compute_dtype == 0) fused::handover_stats(h, b, out);
  API_END(h)
}

// ---- a side stream that really runs beside the main stream ---------------------------------------------------------------------
// HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default; assigned round-robin as streams are made),
// and two streams that land on the same queue execute IN ORDER.  In a process that also holds torch's and RCCL's streams the scoring
// pass's stream shared the main stream's queue in about every other run: the pass then ran strictly before the training forward
// (profiles/r03: 0.41 + 0.41 ms instead of 0.74 ms for the overlapped pair).  So the stream is probed: a kernel on the main stream
// waits (bounded, 200 us) for a flag that a kernel on the candidate sets; if the flag never arrives the two share a queue and the
// next candidate is tried (every new stream advances the round-robin).
__global__ void k_probe_wait(int* flag, int* seen, long long max_ticks) {
  const long long t0 = wall_clock64();   // (100 MHz)
  int f = 0;
  while (!(f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(32);
  *seen = f;
}
__global__ void k_probe_set(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

}  // extern "C"
hipStream_t make_concurrent_stream(kprn_handle* h, int* probes) {
  const int kTries = 8;
  hipStream_t cand[kTries] = {};
  int32_t* d = dalloc<int32_t>(2);   // {flag, seen}
  hipStream_t pick = nullptr;
  int made = 0;
  try {
    for (int i = 0; i < kTries && !pick; ++i) {
      HIP_TRY(hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking));
      made = i + 1;
      HIP_TRY(hipMemsetAsync(d, 0, 2 * sizeof(int32_t), h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
      hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, h->stream, d, d + 1, (long long)20000);
      hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, cand[i], d);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(h->stream));
      HIP_TRY(hipStreamSynchronize(cand[i]));
      int32_t seen = 0;
      HIP_TRY(hipMemcpy(&seen, d + 1, sizeof(int32_t), hipMemcpyDeviceToHost));
      if (seen) pick = cand[i];
    }
  } catch (...) {
    for (int i = 0; i < made; ++i) if (cand[i]) hipStreamDestroy(cand[i]);
    dfree(d);
    throw;
  }
  if (probes) *probes = made; else h->side_stream_probes = made;   // (default: the scoring stream's count, what kprn_get_stat reports)
  if (!pick) pick = cand[0];   // (every candidate shares the main stream's queue: correct anyway, just not concurrent)
  for (int i = 0; i < made; ++i) if (cand[i] && cand[i] != pick) hipStreamDestroy(cand[i]);
  dfree(d);
  return pick;
}
extern "C" {

int kprn_forward_batch_async(kprn_handle* h, const kprn_batch* b, int32_t class_id) {
  API_BEGIN(h)
  if (h->score_rest_batch) launch_score_rest(h);   // (the second part of an earlier split pass that nobody placed: before its buffers are reused)
  h->pool_defer_batch = nullptr;
  h->last_forward_side = false;
  if (h->score_overlap && b && use_fused(h, b, false)) {
    // the pass goes to the side stream with its own output buffers; everything it reads is final on the main stream first
    check_batch(h, b, class_id);
    const int64_t N = (int64_t)b->B * b->P;
    catch_up(h, b);
    ensure_ws_common(h, N, b->B);
    if (!h->score_stream) {
      h->score_stream = make_concurrent_stream(h);
      HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ev_score_done, hipEventDisableTiming));
    }
    if (N > h->cap_N2 || b->B > h->cap_B2) {
      HIP_TRY(hipStreamSynchronize(h->score_stream));
      if (h->rest_stream) HIP_TRY(hipStreamSynchronize(h->rest_stream));
      dfree(h->S2); dfree(h->sel2);
      h->cap_N2 = std::max(N, h->cap_N2); h->cap_B2 = std::max<int64_t>(b->B, h->cap_B2);
      h->S2 = dalloc<float>(h->cap_N2 * h->cfg.C);
      h->sel2 = dalloc<float>(h->cap_B2);
    }
    fused::prefix_forward(h, b);  // (main stream; cached for the pass below and for the training forward of the same batch)
    fused::mc_prepare(h);         // (likewise: the split weights of the matrix-core forward)
    if ((h->score_dual == 1 || (h->score_dual == 2 && fused::small_tiles(h, N, b->tile_k != nullptr))) && h->cfg.compute_dtype == 0 && h->cfg.L == 2 &&
        !(h->score_split > 0.f)) {
      // "score_dual": the WHOLE pass waits for the training forward that usually comes next and rides in its launch (forward_impl, fused::forward_dual) --
      // deferred like the second part of a split pass (tile 0 on): whoever needs its result, its buffers or the parameters it reads first runs it the usual
      // way (join_score -> launch_score_rest)
      if (h->score_pending && h->ev_score_done) HIP_TRY(hipStreamWaitEvent(h->score_stream, h->ev_score_done, 0));   // (an earlier pass wrote the same S2 / sel2)
      h->score_rest_batch = b; h->score_rest_cid = class_id; h->score_rest_tile0 = 0;
      h->score_on_main = false;
      h->score_pending = true;
      h->last_forward_side = true;
      h->last_B = b->B;
      return KPRN_OK;
    }
    h->score_on_main = false;
    HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
    HIP_TRY(hipStreamWaitEvent(h->score_stream, h->ev_fork, 0));
    // an earlier pass (or its deferred part, which may have run on the rest stream: "score_rest_before_bptt") writes the same S2 / sel2: this pass
    // starts behind it whichever stream it finished on
    if (h->score_pending && h->ev_score_done) HIP_TRY(hipStreamWaitEvent(h->score_stream, h->ev_score_done, 0));
    Workspace& w = h->ws;
    hipStream_t main_stream = h->stream;
    float* S0 = w.S; float* sel0 = w.sel;
    h->stream = h->score_stream; w.S = h->S2; w.sel = h->sel2;
    // "score_split" f: the last f of the batch's tiles wait for kprn_forward_batch_async_rest (a data-parallel step places them under its
    // collective); the first part runs now, beside whatever is queued next, on the whole chip
    const int64_t n_tiles = (N + 63) / 64;
    const int64_t t_split = (h->score_split > 0.f && h->cfg.compute_dtype == 0 && n_tiles >= 64 && !fused::small_tiles(h, N, b->tile_k != nullptr))
                                ? std::max<int64_t>(1, std::min<int64_t>(n_tiles - 1, (int64_t)((1.0 - h->score_split) * n_tiles + 0.5))) : n_tiles;
    try {
      fused::forward(h, b, false, 0, t_split < n_tiles ? t_split : -1, t_split < n_tiles);
      if (t_split == n_tiles) pool_stage(h, b, class_id - 1, false);
    } catch (...) { h->stream = main_stream; w.S = S0; w.sel = sel0; throw; }
    h->stream = main_stream; w.S = S0; w.sel = sel0;
    if (t_split < n_tiles) {
      h->score_rest_batch = b; h->score_rest_cid = class_id; h->score_rest_tile0 = t_split;
      if (h->score_rest_before_bptt) {   // (the rest runs on another stream: its pooling stage reads this part's scores)
        if (!h->ev_part1) HIP_TRY(hipEventCreateWithFlags(&h->ev_part1, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(h->ev_part1, h->score_stream));
      }
    } else HIP_TRY(hipEventRecord(h->ev_score_done, h->score_stream));
    h->score_pending = true;
    h->last_forward_side = true;
    h->last_B = b->B;
    return KPRN_OK;
  }
  forward_impl(h, b, class_id, false, true, /*every_class=*/false);  // kprn_read_probs hands out the selected class
  API_END(h)
}

// the second part of a split scoring pass (kprn_set_option "score_split"): the remaining tiles + the pooling stage, behind everything queued
// on the main stream so far
static void launch_score_rest(kprn_handle* h) {
  const kprn_batch* b = h->score_rest_batch;
  h->score_rest_batch = nullptr;
  if (!b) return;
  hipStream_t rs = h->score_stream;
  if (h->score_rest_before_bptt && h->ev_part1) {
    if (!h->rest_stream) {
      int lo = 0, hi = 0;
      HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));   // (lo = the numerically largest value = the LOWEST priority)
      HIP_TRY(hipStreamCreateWithPriority(&h->rest_stream, hipStreamNonBlocking, lo));
    }
    rs = h->rest_stream;
    HIP_TRY(hipStreamWaitEvent(rs, h->ev_part1, 0));
  }
  HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
  HIP_TRY(hipStreamWaitEvent(rs, h->ev_fork, 0));
  Workspace& w = h->ws;
  hipStream_t main_stream = h->stream;
  float* S0 = w.S; float* sel0 = w.sel;
  h->stream = rs; w.S = h->S2; w.sel = h->sel2;
  try {
    fused::forward(h, b, false, h->score_rest_tile0, -1);
    pool_stage(h, b, h->score_rest_cid - 1, false);
  } catch (...) {
    // the pass cannot finish: nobody may wait for it on a stale event, or hand out its half-filled S2 / sel2 as a finished pass
    h->stream = main_stream; w.S = S0; w.sel = sel0;
    h->score_pending = false; h->last_forward_side = false; h->last_B = 0;
    (void)hipStreamSynchronize(rs);
    throw;
  }
  h->stream = main_stream; w.S = S0; w.sel = sel0;
  HIP_TRY(hipEventRecord(h->ev_score_done, rs));
}

static void score_rest_hook(kprn_handle* h) {
  if (h->score_rest_in_backward && h->score_rest_batch) launch_score_rest(h);
}

int kprn_forward_batch_async_rest(kprn_handle* h) {
  API_BEGIN(h)
  launch_score_rest(h);
  API_END(h)
}

int kprn_read_probs(kprn_handle* h, float* probs, int32_t B) {
  API_BEGIN(h)
  KPRN_REQUIRE(probs && B > 0 && B <= h->last_B, KPRN_E_ARG, "bad probs buffer / B");
  if (h->last_forward_side) {
    if (h->score_rest_batch) launch_score_rest(h);
    // (the second part of a split pass may have run on the rest stream, and a join may already have cleared score_pending: the pass's event orders this copy)
    if (h->score_on_main) {   // ("score_dual": the pass ran in a training forward's launch on the main stream)
      HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
      HIP_TRY(hipStreamWaitEvent(h->score_stream, h->ev_fork, 0));
    } else
    if (h->ev_score_done) HIP_TRY(hipStreamWaitEvent(h->score_stream, h->ev_score_done, 0));
    HIP_TRY(hipMemcpyAsync(probs, h->sel2, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, h->score_stream));
    HIP_TRY(hipStreamSynchronize(h->score_stream));
  } else {
    HIP_TRY(hipMemcpyAsync(probs, h->ws.sel, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  prof_drain(h);
  API_END(h)
}

int kprn_forward_batch(kprn_handle* h, const kprn_batch* b, int32_t class_id, float* probs, float* all_probs, float* pooled, float* path_scores) {
  API_BEGIN(h)
  KPRN_REQUIRE(b != nullptr, KPRN_E_ARG, "batch is NULL");
  // the selected class's probabilities (what model:forward hands test_from_checkpoint.lua:109) come back through a page-locked mirror the pool kernel
  // writes: no copy operation between the pass and the caller; the other outputs are copied as before and cost the all-class reduction only when asked for
  struct Disarm { kprn_handle* h; ~Disarm() { h->sel_host_armed = nullptr; } } disarm{h};
  const bool mirror = probs != nullptr && !h->prof_on;
  if (mirror) {
    if ((int64_t)b->B > h->probs_mirror_cap) {
      if (h->probs_mirror) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipHostFree(h->probs_mirror)); h->probs_mirror = nullptr; h->probs_mirror_cap = 0; }
      HIP_TRY(hipHostMalloc((void**)&h->probs_mirror, (size_t)(2 * (int64_t)b->B + 64) * sizeof(float)));
      h->probs_mirror_cap = 2 * (int64_t)b->B + 64;
    }
    h->sel_host_armed = h->probs_mirror;
  }
  forward_impl(h, b, class_id, false, /*do_pool=*/true, /*every_class=*/all_probs != nullptr || pooled != nullptr);
  const int C = h->cfg.C;
  hipStream_t s = h->stream;
  if (probs && !mirror) HIP_TRY(hipMemcpyAsync(probs, h->ws.sel, (size_t)b->B * sizeof(float), hipMemcpyDeviceToHost, s));
  if (all_probs) HIP_TRY(hipMemcpyAsync(all_probs, h->ws.probs, (size_t)b->B * C * sizeof(float), hipMemcpyDeviceToHost, s));
  if (pooled) HIP_TRY(hipMemcpyAsync(pooled, h->ws.pooled, (size_t)b->B * C * sizeof(float), hipMemcpyDeviceToHost, s));
  if (path_scores)
    HIP_TRY(hipMemcpyAsync(path_scores, h->score_buf, (size_t)b->B * b->P * C * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (mirror) memcpy(probs, h->probs_mirror, (size_t)b->B * sizeof(float));
  prof_drain(h);
  API_END(h)
}

// the host-buffer entry points' minibatch -> an engine-owned slot (0 / 1: training, 2 / 3: scoring), derived on the calling thread
static int dropin_feed(kprn_handle* h, bool score, const int32_t* idx, const float* labels, int32_t B, int32_t P, int32_t T, int32_t F) {
  API_BEGIN(h)
  int& nx = score ? h->dropin_next_score : h->dropin_next_train;
  const int si = (score ? 2 : 0) + nx;
  nx ^= 1;
  if (!h->feed_build_host) {   // device-built feed selected: the plain create path of that build
    if (h->dropin_slot[si]) { kprn_batch* old = h->dropin_slot[si]; h->dropin_slot[si] = nullptr; kprn_batch_destroy(h, old); }
    kprn_batch* b = nullptr;
    const int rc = kprn_batch_create(h, idx, labels, B, P, T, F, &b);
    if (rc != KPRN_OK) return rc;
    h->dropin_slot[si] = b;
  } else {
    feed_impl(h, &h->dropin_slot[si], idx, labels, nullptr, B, P, T, F, /*inline_now=*/true);
  }
  h->dropin_last = si;
  API_END(h)
}

int kprn_forward(kprn_handle* h, const int32_t* idx, int32_t B, int32_t P, int32_t T, int32_t F, int32_t class_id, float* probs, float* all_probs) {
  if (!h) return KPRN_E_ARG;
  int rc = dropin_feed(h, /*score=*/true, idx, nullptr, B, P, T, F);   // (engine-owned feed slots: see kprn_train_step)
  if (rc != KPRN_OK) return rc;
  return kprn_forward_batch(h, h->dropin_slot[h->dropin_last], class_id, probs, all_probs, nullptr, nullptr);
}

int kprn_embed(kprn_handle* h, const int32_t* idx, int64_t N, int32_t T, int32_t F, float* x) {
  API_BEGIN(h)
  KPRN_REQUIRE(idx && x && N > 0 && T > 0 && F == h->cfg.F, KPRN_E_ARG, "bad arguments");
  KPRN_REQUIRE(N < (1ll << 31), KPRN_E_ARG, "N too large");
  kprn_batch* b = nullptr;
  {
    int rc = kprn_batch_create(h, idx, nullptr, (int32_t)N, 1, T, F, &b);
    if (rc != KPRN_OK) return rc;
  }
  float* dx = nullptr;
  try {
    catch_up(h, b);
    const kprn_config& c = h->cfg;
    dx = dalloc<float>(N * T * h->D);
    kk::embed_gather(h->stream, b->idx, N, T, F, c.num_types, h->dense + h->off_Wt, h->We, h->dense + h->off_Wr, c.dt, c.de, c.dr, dx, false);
    HIP_TRY(hipMemcpyAsync(x, dx, (size_t)N * T * h->D * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  } catch (...) { dfree(dx); kprn_batch_destroy(h, b); throw; }
  dfree(dx);
  kprn_batch_destroy(h, b);
  API_END(h)
}

int kprn_backward_batch(kprn_handle* h, const kprn_batch* b, int32_t class_id, int32_t bce_literal, float inv_batch, float* loss) {
  API_BEGIN(h)
  backward_impl(h, b, class_id, bce_literal, inv_batch);
  if (loss) {
    form_loss(h);
    HIP_TRY(hipMemcpyAsync(loss, h->d_loss, sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    prof_drain(h);
  }
  API_END(h)
}

int kprn_apply_update(kprn_handle* h, const kprn_opt* opt) {
  API_BEGIN(h)
  apply_update_impl(h, opt);
  API_END(h)
}

int kprn_train_step_batch(kprn_handle* h, const kprn_batch* b, int32_t class_id, const kprn_opt* opt, float* loss) {
  API_BEGIN(h)
  KPRN_REQUIRE(opt, KPRN_E_ARG, "opt is NULL");
  check_batch(h, b, class_id);
  catch_up(h, b);
  if (!h->pad_clean) { zero_pad_tokens(h); fused::params_changed(h); bf16p::params_changed(h, false); }  // MyOptimizer.lua:181 (a no-op when the last step left them zero)
  // the loss goes back as soon as the loss stage has run (option "train_step_return" = "loss"); profiling and "drain" wait for the whole step as before
  struct Disarm { kprn_handle* h; ~Disarm() { h->loss_early_armed = false; } } disarm{h};
  if (loss && !h->train_step_drain && !h->prof_on) {
    const int64_t np = kk::loss_partials(b->B);
    if (np > h->loss_mirror_cap) {
      if (h->loss_mirror) { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipHostFree(h->loss_mirror)); h->loss_mirror = nullptr; h->loss_mirror_cap = 0; }
      HIP_TRY(hipHostMalloc((void**)&h->loss_mirror, (size_t)(2 * np + 64) * sizeof(float)));
      h->loss_mirror_cap = 2 * np + 64;
    }
    if (!h->ev_loss) HIP_TRY(hipEventCreateWithFlags(&h->ev_loss, hipEventDisableTiming));
    h->loss_early_armed = true;
  }
  backward_impl(h, b, class_id, opt->bce_literal, 0.f);
  apply_update_impl(h, opt);
  if (loss && h->loss_early_armed) {
    HIP_TRY(hipEventSynchronize(h->ev_loss));
    float s = 0.f;   // k_sum_partials' order: one running fp32 sum over the partials in index order
    for (int i = 0; i < h->loss_early_n; ++i) s += h->loss_mirror[i];
    *loss = s;
  } else if (loss) {
    form_loss(h);
    HIP_TRY(hipMemcpyAsync(loss, h->d_loss, sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    prof_drain(h);
  }
  API_END(h)
}

int kprn_train_step(kprn_handle* h, const int32_t* idx, int32_t B, int32_t P, int32_t T, int32_t F, const float* labels, int32_t class_id,
                    const kprn_opt* opt, float* loss) {
  if (!h) return KPRN_E_ARG;
  if (!labels) { h->err = "assert(targets) (MyOptimizer.lua:179)"; return KPRN_E_ARG; }
  // The minibatch goes through one of two engine-owned feed slots (grow-only capacity, page-locked staging: steady state allocates nothing and
  // frees nothing; alternating, so that the rows the lazy optimiser still names belong to the OTHER slot) instead of a batch created and
  // destroyed per call (two device allocations, a device-side index build with its synchronisations and three stream drains per step:
  // 1.06 ms per 128-pair step against 0.38 ms for the step itself).
  // dropin_prev_waited says "the previous call waited for its own step's loss": it holds for exactly one call.  Taken and cleared here, so that a call
  // that fails anywhere below (an id out of range, a bad argument) leaves it false and the call after it uploads in stream order again.
  const bool prev_waited = h->dropin_prev_waited;
  h->dropin_prev_waited = false;
  h->dropin_prev_waited_now = prev_waited;   // (what the inline feed of THIS call reads)
  int rc = dropin_feed(h, /*score=*/false, idx, labels, B, P, T, F);
  h->dropin_prev_waited_now = false;
  if (rc != KPRN_OK) return rc;
  // (the loss is always fetched -- the call waits for the loss stage whether or not the caller passed `loss`: that wait is what lets the NEXT call's
  // upload run beside this step's backward)
  float l = 0.f;
  rc = kprn_train_step_batch(h, h->dropin_slot[h->dropin_last], class_id, opt, &l);
  h->dropin_prev_waited = (rc == KPRN_OK);
  if (loss) *loss = l;
  return rc;
}

int kprn_read_loss(kprn_handle* h, float* loss) {
  API_BEGIN(h)
  KPRN_REQUIRE(loss, KPRN_E_ARG, "loss is NULL");
  form_loss(h);
  HIP_TRY(hipMemcpyAsync(loss, h->d_loss, sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  prof_drain(h);
  API_END(h)
}

int kprn_format_score_lines(int64_t counter0, const float* probs, const float* labels, int64_t n, char* out, int64_t cap, int64_t* written) {
  if (!probs || !labels || !written || n < 0 || (cap > 0 && !out)) return KPRN_E_ARG;
  try {
    const unsigned hc = std::thread::hardware_concurrency();
    *written = hostfeed::format_scores(counter0, probs, labels, n, out, cap, hc >= 64 ? 16 : (hc >= 8 ? 4 : 1));
  } catch (...) { return KPRN_E_NOMEM; }
  return *written < 0 ? KPRN_E_ARG : KPRN_OK;
}

int kprn_read_loss_sum(kprn_handle* h, float* sum, int32_t* steps, int32_t reset) {
  API_BEGIN(h)
  KPRN_REQUIRE(sum && steps, KPRN_E_ARG, "sum / steps is NULL");
  form_loss(h);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  HIP_TRY(hipMemcpyAsync(v, h->d_loss, 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  if (reset) HIP_TRY(hipMemsetAsync(h->d_loss + 1, 0, 2 * sizeof(float), h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *sum = v[1]; *steps = (int32_t)v[2];
  prof_drain(h);
  API_END(h)
}

int kprn_sync(kprn_handle* h) {
  API_BEGIN(h)
  if (h->score_stream) HIP_TRY(hipStreamSynchronize(h->score_stream));
      if (h->rest_stream) HIP_TRY(hipStreamSynchronize(h->rest_stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  prof_drain(h);
  API_END(h)
}

// ---- data-parallel hooks ------------------------------------------------------------------
int kprn_dense_grad_buffer(kprn_handle* h, void** dev_ptr, int64_t* n_floats) {
  API_BEGIN(h)
  KPRN_REQUIRE(dev_ptr && n_floats, KPRN_E_ARG, "NULL argument");
  KPRN_REQUIRE(h->stream_known, KPRN_E_ARG,
               "data-parallel hooks: the caller's collectives must be ordered with the engine's stream, but kprn_config.stream was NULL (the engine "
               "made its own) and kprn_stream() was never called -- pass a stream (KPRN_STREAM_LEGACY_DEFAULT for the null stream) or fetch the engine's");
  *dev_ptr = h->g_dense;
  *n_floats = h->n_dense;
  API_END(h)
}

int kprn_sparse_grad_capacity(kprn_handle* h, int32_t* max_rows) {
  API_BEGIN(h)
  KPRN_REQUIRE(max_rows, KPRN_E_ARG, "NULL argument");
  *max_rows = (int32_t)std::min<int64_t>(h->step_rows_ub, h->cfg.Ve);
  API_END(h)
}

// this step's touched entity rows (+ the dense arena behind them, dp_dense_in_pack) -> dst[words], cleared from the accumulator
static void pack_into(kprn_handle* h, int32_t capacity, int32_t* dst) {
  const int de = h->cfg.de;
  h->pack_cap = capacity;  // the capacity THIS buffer is laid out with (ids at +4, rows at +4+capacity): merge checks against it
  if (!h->view_batch) { h->rows_view = h->step_rows; h->count_view = h->step_count; }
  ProfScope ps(h, "dp_pack_rows");   // rows moved out of g_We (zeroed behind) + the dense arena behind them, one launch
  float* tail = (float*)(dst + 4 + (int64_t)capacity * (1 + de));
  kk::pack_rows(h->stream, h->g_We, h->rows_view, h->count_view, h->step_rows_ub, de, dst + 4, (float*)(dst + 4 + capacity), dst, h->g_dense,
                h->dp_dense_in_pack ? h->n_dense : 0, tail);
}

int kprn_sparse_grad_pack(kprn_handle* h, int32_t capacity, void** dev_buf, int64_t* n_words) {
  API_BEGIN(h)
  KPRN_REQUIRE(dev_buf && n_words, KPRN_E_ARG, "NULL argument");
  KPRN_REQUIRE(h->stream_known, KPRN_E_ARG,
               "data-parallel hooks: the caller's collectives must be ordered with the engine's stream, but kprn_config.stream was NULL (the engine "
               "made its own) and kprn_stream() was never called -- pass a stream (KPRN_STREAM_LEGACY_DEFAULT for the null stream) or fetch the engine's");
  KPRN_REQUIRE(capacity >= h->step_rows_ub && capacity > 0, KPRN_E_ARG, "capacity smaller than this step's touched-row count");
  materialize_union(h);   // (a gathered gradient still pending: make it what the unfused merge would have left before packing again)
  const int64_t words = 4 + (int64_t)capacity * (1 + h->cfg.de) + dp_tail_words(h);
  if (words > h->pack_words) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    dfree(h->pack_buf);
    h->pack_buf = dalloc<int32_t>(words);
    h->pack_words = words;
  }
  pack_into(h, capacity, h->pack_buf);
  *dev_buf = h->pack_buf;
  *n_words = words;
  API_END(h)
}

static void merge_impl(kprn_handle* h, const void* dev_all, int32_t world, int32_t capacity) {
  KPRN_REQUIRE(capacity == h->pack_cap, KPRN_E_ARG, "capacity differs from the packed buffer's (every rank packs with the same capacity)");
  const int64_t n = (int64_t)world * capacity;
  if (n + 4 > h->step_rows_cap) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    dfree(h->step_rows);
    h->step_rows_cap = (n + 4) * 2;
    h->step_rows = dalloc<int32_t>(h->step_rows_cap);
  }
  // the dense arena: sum of the W copies in rank order, now (the dense update and the norm read g_dense)
  if (h->dp_dense_in_pack) {
    ProfScope ps(h, "dp_dense_sum");
    const int64_t row_words = 4 + (int64_t)capacity * (1 + h->cfg.de);
    hipLaunchKernelGGL(k_dense_from_all, dim3((unsigned)std::min<int64_t>((h->n_dense + 255) / 256, 512)), dim3(256), 0, h->stream, (const float*)dev_all, (int)world,
                       row_words + dp_tail_words(h), row_words, h->n_dense, h->g_dense);
    HIP_TRY(hipGetLastError());
  }
  // the rows: recorded; the optimiser walks the gathered buffer itself (dp_fused_update: the caller keeps dev_all alive and unchanged until
  // kprn_apply_update returns) or the union is built here -- the optimiser then walks the union of all ranks' rows (sorted); exact count on the
  // device, upper bound on the host
  h->dp_all = dev_all; h->dp_world = world; h->dp_cap = capacity; h->dp_union_pending = true;
  h->view_batch = nullptr; h->rows_view = h->step_rows; h->count_view = h->step_count;
  h->step_rows_ub = std::min<int64_t>(n, h->cfg.Ve);
  h->ent_grads_dirty = true;
  if (!h->dp_fused_update) materialize_union(h);
}

int kprn_sparse_grad_merge(kprn_handle* h, const void* dev_all, int32_t world, int32_t capacity) {
  API_BEGIN(h)
  KPRN_REQUIRE(dev_all && world > 0 && capacity > 0, KPRN_E_ARG, "bad argument");
  KPRN_REQUIRE(h->stream_known, KPRN_E_ARG,
               "data-parallel hooks: the caller's collectives must be ordered with the engine's stream, but kprn_config.stream was NULL (the engine "
               "made its own) and kprn_stream() was never called -- pass a stream (KPRN_STREAM_LEGACY_DEFAULT for the null stream) or fetch the engine's");
  merge_impl(h, dev_all, world, capacity);
  API_END(h)
}

// ---- the exchange issued by the engine itself: RCCL on the engine's stream(s) ------------------------------------------------------
// The hooks above leave the collective to the caller (torch.distributed: its own stream, two cross-stream hand-overs of ~20 us each and
// ~0.3 ms of Python per step between pack and update).  Here the engine holds a communicator of its own and queues
// pack -> ncclAllGather (IN PLACE: every rank packs straight into its slot of the gathered buffer) -> dense sum -> optimiser step (the
// union of the rows inside the row kernel) back to back from one C call.  librccl is dlopen'ed (the copy the host process already
// uses -- torch ships one -- so the library has no link-time dependency on it); the communicator is bootstrapped by the caller's own
// control plane: rank 0 draws the 128-byte id (kprn_dp_unique_id), the caller broadcasts it, every rank calls kprn_dp_init.
}  // extern "C"   (the helpers below are C++: they return std::string)
namespace {
struct Id128 { char b[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES), passed BY VALUE to ncclCommInitRank
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::string g_rccl_err;
bool rccl_load(const char* path) {
  if (g_rccl.lib) return true;
  const char* names[] = {path && path[0] ? path : nullptr, "librccl.so", "librccl.so.1"};
  void* lib = nullptr;
  for (const char* n : names) {
    if (!n) continue;
    lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
    const char* e = dlerror();
    g_rccl_err = e ? e : "dlopen failed";
  }
  if (!lib) return false;
  Rccl r;
  r.lib = lib;
  *(void**)&r.GetUniqueId = dlsym(lib, "ncclGetUniqueId");
  *(void**)&r.CommInitRank = dlsym(lib, "ncclCommInitRank");
  *(void**)&r.AllGather = dlsym(lib, "ncclAllGather");
  *(void**)&r.CommDestroy = dlsym(lib, "ncclCommDestroy");
  *(void**)&r.CommCount = dlsym(lib, "ncclCommCount");
  *(void**)&r.GetErrorString = dlsym(lib, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy) { g_rccl_err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy"; return false; }
  g_rccl = r;
  return true;
}
std::string rccl_msg(const char* what, int rc) {
  return std::string(what) + " failed: " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?") + " (" + std::to_string(rc) + ")";
}
const int kNcclInt32 = 2;   // ncclDataType_t (rccl.h)
}  // namespace
extern "C" {

static void dp_release(kprn_handle* h) {
  if (h->dp_comm && g_rccl.CommDestroy) {
    g_rccl.CommDestroy(h->dp_comm);
    h->dp_dense_in_pack = h->dp_saved_dense_in_pack;   // (kprn_dp_init forced both on)
    h->dp_fused_update = h->dp_saved_fused_update;
  }
  h->dp_comm = nullptr;
  // an exchange in flight dies with the communicator: nothing may point into the gathered buffer freed below (a later materialize_union /
  // kprn_get_grad / kprn_sparse_grad_pack would read it).  The rows a begun exchange had moved out of g_We are lost with it -- the caller
  // shut the exchange down between begin and finish.
  h->dp_begun = false; h->dp_union_pending = false; h->dp_all = nullptr; h->dp_world = 0; h->dp_cap = 0;
  if (h->dp_comm_stream) { hipStreamSynchronize(h->dp_comm_stream); hipStreamDestroy(h->dp_comm_stream); h->dp_comm_stream = nullptr; }
  if (h->ev_dp_packed) { hipEventDestroy(h->ev_dp_packed); h->ev_dp_packed = nullptr; }
  if (h->ev_dp_gathered) { hipEventDestroy(h->ev_dp_gathered); h->ev_dp_gathered = nullptr; }
  dfree(h->dp_gather); h->dp_gather_words = 0;
}

int kprn_dp_available(const char* rccl_path) {
  if (!rccl_load(rccl_path)) { g_create_error = "cannot load librccl: " + g_rccl_err; return KPRN_E_DEVICE; }
  return KPRN_OK;
}

int kprn_dp_unique_id(const char* rccl_path, void* id128) {
  if (!id128) return KPRN_E_ARG;
  if (!rccl_load(rccl_path)) { g_create_error = "kprn_dp_unique_id: cannot load librccl: " + g_rccl_err; return KPRN_E_DEVICE; }
  const int rc = g_rccl.GetUniqueId(id128);
  if (rc != 0) { g_create_error = rccl_msg("ncclGetUniqueId", rc); return KPRN_E_DEVICE; }
  return KPRN_OK;
}

int kprn_dp_init(kprn_handle* h, const char* rccl_path, const void* id128, int32_t rank, int32_t world) {
  API_BEGIN(h)
  KPRN_REQUIRE(id128 && world >= 1 && rank >= 0 && rank < world, KPRN_E_ARG, "bad id / rank / world");
  KPRN_REQUIRE(!h->dp_comm, KPRN_E_ARG, "kprn_dp_init: this handle already holds a communicator");
  KPRN_REQUIRE(rccl_load(rccl_path), KPRN_E_DEVICE, "cannot load librccl: " + g_rccl_err);
  Id128 id;
  memcpy(id.b, id128, 128);
  void* comm = nullptr;
  const int rc = g_rccl.CommInitRank(&comm, world, id, rank);   // (collective: every rank of the job is in this call)
  KPRN_REQUIRE(rc == 0 && comm, KPRN_E_DEVICE, rccl_msg("ncclCommInitRank", rc));
  h->dp_comm = comm; h->dp_rank = rank; h->dp_nranks = world;
  h->dp_saved_dense_in_pack = h->dp_dense_in_pack; h->dp_saved_fused_update = h->dp_fused_update;
  h->dp_dense_in_pack = true;    // one collective per step
  h->dp_fused_update = true;     // union of the rows inside the row update
  h->stream_known = true;        // (the collective is queued by the engine: nobody else has to know the stream)
  API_END(h)
}

int kprn_dp_comm_size(kprn_handle* h, int32_t* nranks) {
  API_BEGIN(h)
  KPRN_REQUIRE(nranks, KPRN_E_ARG, "NULL argument");
  KPRN_REQUIRE(h->dp_comm, KPRN_E_ARG, "kprn_dp_comm_size before kprn_dp_init");
  int n = h->dp_nranks;
  if (g_rccl.CommCount) {   // RCCL's own count of the communicator's ranks (a self-check for scaling runs: must equal the launcher's world size)
    const int rc = g_rccl.CommCount(h->dp_comm, &n);
    KPRN_REQUIRE(rc == 0, KPRN_E_DEVICE, rccl_msg("ncclCommCount", rc));
  }
  *nranks = n;
  API_END(h)
}

int kprn_dp_shutdown(kprn_handle* h) {
  API_BEGIN(h)
  HIP_TRY(hipStreamSynchronize(h->stream));
  dp_release(h);
  API_END(h)
}

int kprn_dp_exchange_begin(kprn_handle* h, int32_t capacity) {
  API_BEGIN(h)
  KPRN_REQUIRE(h->dp_comm, KPRN_E_ARG, "kprn_dp_exchange_begin before kprn_dp_init");
  KPRN_REQUIRE(capacity >= h->step_rows_ub && capacity > 0, KPRN_E_ARG, "capacity smaller than this step's touched-row count");
  KPRN_REQUIRE((capacity & 3) == 0, KPRN_E_ARG, "capacity must be a multiple of 4 rows (every rank's slot of the gathered buffer stays 16-byte aligned)");
  materialize_union(h);
  const int W = h->dp_nranks;
  const int64_t words = 4 + (int64_t)capacity * (1 + h->cfg.de) + dp_tail_words(h);
  if (words * W > h->dp_gather_words) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->dp_comm_stream) HIP_TRY(hipStreamSynchronize(h->dp_comm_stream));
    dfree(h->dp_gather);
    h->dp_gather = dalloc<int32_t>(words * W);
    h->dp_gather_words = words * W;
  }
  int32_t* mine = h->dp_gather + (int64_t)h->dp_rank * words;
  pack_into(h, capacity, mine);
  hipStream_t cs = h->stream;
  if (h->dp_comm_stream_on) {   // (also at world 1, where the collective is empty: the hand-over is the same code a world-8 run takes)
    if (!h->dp_comm_stream) {
      h->dp_comm_stream = make_concurrent_stream(h);
      HIP_TRY(hipEventCreateWithFlags(&h->ev_dp_packed, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&h->ev_dp_gathered, hipEventDisableTiming));
    }
    cs = h->dp_comm_stream;
    HIP_TRY(hipEventRecord(h->ev_dp_packed, h->stream));
    HIP_TRY(hipStreamWaitEvent(cs, h->ev_dp_packed, 0));
  }
  {
    ProfScope ps(h, "dp_allgather");
    // in place: sendbuff = recvbuff + rank * count (no staging copy; with one rank RCCL returns at once)
    const int rc = g_rccl.AllGather(mine, h->dp_gather, (size_t)words, kNcclInt32, h->dp_comm, cs);
    KPRN_REQUIRE(rc == 0, KPRN_E_DEVICE, rccl_msg("ncclAllGather", rc));
  }
  if (cs != h->stream) HIP_TRY(hipEventRecord(h->ev_dp_gathered, cs));
  h->dp_begun = true;
  API_END(h)
}

int kprn_dp_exchange_finish(kprn_handle* h, const kprn_opt* opt) {
  API_BEGIN(h)
  KPRN_REQUIRE(h->dp_comm && h->dp_begun, KPRN_E_ARG, "kprn_dp_exchange_finish without kprn_dp_exchange_begin");
  h->dp_begun = false;
  if (h->dp_comm_stream_on && h->dp_comm_stream) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_dp_gathered, 0));
  merge_impl(h, h->dp_gather, h->dp_nranks, (int32_t)h->pack_cap);
  apply_update_impl(h, opt);
  API_END(h)
}

/* the stream everything is queued on */
int kprn_stream(kprn_handle* h, void** stream) {
  API_BEGIN(h)
  KPRN_REQUIRE(stream, KPRN_E_ARG, "NULL argument");
  *stream = (void*)h->stream;
  h->stream_known = true;
  API_END(h)
}

// ---- checkpoints ------------------------------------------------------------------------
static const char kMagic[8] = {'K', 'P', 'R', 'N', 'A', 'M', 'D', '1'};

int kprn_save(kprn_handle* h, const char* path) {
  API_BEGIN(h)
  KPRN_REQUIRE(path, KPRN_E_ARG, "path is NULL");
  std::vector<float> flat((size_t)h->n_params);
  {
    int rc = kprn_get_flat_params(h, flat.data(), h->n_params);
    if (rc != KPRN_OK) return rc;
  }
  FILE* f = fopen(path, "wb");
  KPRN_REQUIRE(f, KPRN_E_IO, std::string("cannot open for writing: ") + path);
  const kprn_config& c = h->cfg;
  int32_t hdr[16] = {c.Vt, c.Ve, c.Vr, c.dt, c.de, c.dr, c.F, c.num_types, c.H, c.L, c.C, c.rnn_type, c.reducer, c.K, 0, 0};
  bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(hdr, sizeof(int32_t), 16, f) == 16 && fwrite(&h->n_params, sizeof(int64_t), 1, f) == 1 &&
            fwrite(flat.data(), sizeof(float), flat.size(), f) == flat.size();
  ok = (fclose(f) == 0) && ok;
  KPRN_REQUIRE(ok, KPRN_E_IO, "short write");
  API_END(h)
}

int kprn_load(kprn_handle* h, const char* path) {
  API_BEGIN(h)
  KPRN_REQUIRE(path, KPRN_E_ARG, "path is NULL");
  FILE* f = fopen(path, "rb");
  KPRN_REQUIRE(f, KPRN_E_IO, std::string("cannot open: ") + path);
  char magic[8];
  int32_t hdr[16];
  int64_t n = 0;
  bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, kMagic, 8) == 0 && fread(hdr, sizeof(int32_t), 16, f) == 16 &&
            fread(&n, sizeof(int64_t), 1, f) == 1;
  const kprn_config& c = h->cfg;
  const int32_t want[12] = {c.Vt, c.Ve, c.Vr, c.dt, c.de, c.dr, c.F, c.num_types, c.H, c.L, c.C, c.rnn_type};
  if (ok) ok = memcmp(hdr, want, sizeof(want)) == 0 && n == h->n_params;
  std::vector<float> flat;
  if (ok) { flat.resize((size_t)n); ok = fread(flat.data(), sizeof(float), flat.size(), f) == flat.size(); }
  fclose(f);
  KPRN_REQUIRE(ok, KPRN_E_IO, "not a kprn checkpoint for this configuration");
  return kprn_set_flat_params(h, flat.data(), n);
  API_END(h)
}

// ---- measurement ------------------------------------------------------------------------
int kprn_profile_enable(kprn_handle* h, int32_t on) {
  API_BEGIN(h)
  prof_drain(h);
  h->prof_on = on != 0;
  API_END(h)
}
int kprn_profile_reset(kprn_handle* h) {
  API_BEGIN(h)
  prof_drain(h);
  h->prof.clear();
  API_END(h)
}
int kprn_profile_get(kprn_handle* h, kprn_prof_entry* out, int32_t cap, int32_t* n) {
  API_BEGIN(h)
  KPRN_REQUIRE(n, KPRN_E_ARG, "n is NULL");
  prof_drain(h);
  int k = 0;
  for (auto& kv : h->prof) {
    if (out && k < cap) {
      memset(&out[k], 0, sizeof(kprn_prof_entry));
      strncpy(out[k].name, kv.first.c_str(), sizeof(out[k].name) - 1);
      out[k].total_ms = kv.second.total_ms;
      out[k].launches = kv.second.launches;
    }
    ++k;
  }
  *n = k;
  API_END(h)
}

/* measurement hook (scripts/gpu_gemm_bench.py): times one GEMM shape of the generic pipeline on random data.
 * what: 0 = C = A B^T (forward i2g), 1 = C = A B (dx / dh), 2 = C += A^T B with split-K (dW), 3 = FastLSTM step kernel (M paths, N = H, K = Din),
 *       4 = Recurrence step kernel.  Returns the mean milliseconds per launch in *ms. */
int kprn_debug_gemm(kprn_handle* h, int32_t what, int64_t M, int32_t N, int64_t K, int32_t iters, float* ms) {
  API_BEGIN(h)
  KPRN_REQUIRE(ms && M > 0 && N > 0 && K > 0 && iters > 0, KPRN_E_ARG, "bad argument");
  hipStream_t s = h->stream;
  if (what == 5 || what == 6) {   // the bf16 pipeline's product (lstm_bf16.hip): 5 = C = A B^T, 6 = split-K accumulate
    *ms = bf16p::debug_gemm16(s, M, N, K, what == 6 ? 1024 : 1, iters);
    return KPRN_OK;
  }
  float *A = nullptr, *B = nullptr, *C = nullptr, *X = nullptr, *Z = nullptr;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
  try {
    const int64_t H = N, Din = K;
    if (what <= 2) {
      const int64_t na = M * K, nb = (int64_t)N * K, nc = (what == 2) ? (int64_t)N * K : M * (int64_t)N;
      A = dalloc<float>(what == 2 ? M * (int64_t)N : na); B = dalloc<float>(what == 2 ? M * K : nb); C = dalloc<float>(nc);
      kk::fill_uniform(s, A, what == 2 ? M * (int64_t)N : na, 0.1f, 1, 0); kk::fill_uniform(s, B, what == 2 ? M * K : nb, 0.1f, 2, 0);
      HIP_TRY(hipMemsetAsync(C, 0, (size_t)nc * sizeof(float), s));
    } else {
      X = dalloc<float>(M * Din); A = dalloc<float>(2 * M * H); B = dalloc<float>(4 * H * (Din + H) + 8 * H); C = dalloc<float>(2 * M * H); Z = dalloc<float>(M * 4 * H);
      kk::fill_uniform(s, X, M * Din, 0.1f, 1, 0); kk::fill_uniform(s, A, 2 * M * H, 0.1f, 2, 0); kk::fill_uniform(s, B, 4 * H * (Din + H) + 8 * H, 0.1f, 3, 0);
    }
    for (int it = -2; it < iters; ++it) {
      if (it == 0) HIP_TRY(hipEventRecord(e0, s));
      if (what == 0) gemm::run(s, A, K, 1, B, 1, K, C, N, M, N, K, false, nullptr, 1);
      else if (what == 1) gemm::run(s, A, K, 1, B, N, 1, C, N, M, N, K, false, nullptr, 1);            // B [K][N] n-contiguous
      else if (what == 2) gemm::run(s, A, 1, N, B, K, 1, C, K, N, (int)K, M, true, nullptr, 1024);     // C[N][K] += A[M][N]^T B[M][K]
      else if (what == 3) gemm::lstm_step(s, X, Din, (int)Din, B, B + 4 * H * (Din + H), A, B + 4 * H * Din, A + M * H, C, C + M * H, H, Z, M, (int)H);
      else gemm::rnn_step(s, X, Din, (int)Din, B, B + 4 * H * (Din + H), A, B + H * Din, B + 4 * H * (Din + H) + H, A + M * H, Z, C, H, M, (int)H, 1);
    }
    HIP_TRY(hipEventRecord(e1, s));
    HIP_TRY(hipEventSynchronize(e1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, e0, e1));
    *ms = t / (float)iters;
  } catch (...) { dfree(A); dfree(B); dfree(C); dfree(X); dfree(Z); hipEventDestroy(e0); hipEventDestroy(e1); throw; }
  dfree(A); dfree(B); dfree(C); dfree(X); dfree(Z); hipEventDestroy(e0); hipEventDestroy(e1);
  API_END(h)
}

int kprn_set_option(kprn_handle* h, const char* key, const char* value) {
  API_BEGIN(h)
  KPRN_REQUIRE(key && value, KPRN_E_ARG, "NULL argument");
  if (strcmp(key, "impl") == 0) {
    if (strcmp(value, "auto") == 0) h->impl = 0;
    else if (strcmp(value, "generic") == 0) h->impl = 1;
    else throw KprnError{KPRN_E_ARG, "impl must be auto or generic"};
  } else if (strcmp(key, "prefix_plan") == 0) {
    h->prefix_plan = atoi(value) ? 1 : 0;  // batches created from now on (an existing batch keeps what it was built with)
  } else if (strcmp(key, "small_tiles") == 0) {
    // batches created / fed from now on: <= 8 192 paths on tiles of one 16-row m-tile and no identical-prefix plan ("1", default), or the
    // 64-path tiles at every size ("0").  (A split scoring pass whose second part is still unplaced is finished first: the rest launch decides
    // its tiling from this option.)
    join_score(h);
    h->small_tiles_on = atoi(value) != 0;
  } else if (strcmp(key, "tile_handover") == 0) {
    // fused D = H = 64 backward launches: a tile may change workgroups once between two of its steps -- "2" (default): workgroup b paired with b + G / 2,
    // "1": with G - 1 - b -- or workgroups run whole tiles only ("0")
    const int v = atoi(value);
    KPRN_REQUIRE(v >= 0 && v <= 2, KPRN_E_ARG, "tile_handover must be 0, 1 or 2");
    h->tile_handover = v;
  } else if (strcmp(key, "adam_merged") == 0) {
    // lazy-exact Adam: the entity rows' update and the dense arena's update in ONE launch ("1", default) or in two ("0": the A/B reference; bit-identical)
    h->adam_merged = atoi(value) != 0;
  } else if (strcmp(key, "bwd_pipe") == 0) {
    // fused path, small batches (16-row tiles), two layers: both layers' BPTT as ONE launch with the bottom layer one step behind the top layer ("1", default)
    // or as two launches ("0": the A/B reference)
    h->bwd_pipe = atoi(value) != 0;
  } else if (strcmp(key, "score_dual") == 0) {
    // with "score_overlap": a scoring pass queued by kprn_forward_batch_async is held back and runs in the launch of the training forward that follows
    // (one kernel, two branches: no second stream, no fork / join events) -- "1": always, "2" (default): for batches below the 16-row-tile threshold
    // (8 192 paths: measured break-even; above it the side stream's pass lets the loss stage and the backward start under its tail), "0": never
    const int v = atoi(value);
    KPRN_REQUIRE(v >= 0 && v <= 2, KPRN_E_ARG, "score_dual must be 0, 1 or 2");
    h->score_dual = v;
  } else if (strcmp(key, "catchup_prefix") == 0) {
    // fused D = H = 64 path: a batch's lazy-exact row catch-up and its identical-prefix table in one launch ("1", default) or in two ("0": the A/B reference;
    // bit-identical)
    h->catchup_prefix = atoi(value) != 0;
  } else if (strcmp(key, "fused_small_tables") == 0) {
    // fused D = H = 64 path: the type / relation table gradients are formed inside the bottom layer's BPTT launch (one-hot MFMAs on the dx registers) ("1", default) or by a
    // passenger job of the entity-gradient launch that re-reads dx ("0": the A/B reference)
    h->fused_small_tables = atoi(value) != 0;
  } else if (strcmp(key, "persist_layers") == 0) {
    // generic fp32 pipelines: a recurrent layer as ONE persistent launch ("1", default: where layer_f32_persist.hip takes the shape and the batch gives
    // every CU a tile; "2": at any batch size -- tests) or one launch per step ("0")
    const int v = atoi(value);
    KPRN_REQUIRE(v >= 0 && v <= 2, KPRN_E_ARG, "persist_layers must be 0, 1 or 2");
    h->persist_layers = v;
  } else if (strcmp(key, "small_tables") == 0) {
    // generic fp32 pipelines: layer 0's type / relation gradients from G = dA^T [S_r | S_t] ("1", default) or from the full dx product + the
    // table-gradient launch ("0": the A/B reference)
    h->small_tables = atoi(value) != 0;
  } else if (strcmp(key, "bf16_t_pad") == 0) {
    bf16p::set_t_pad(atoi(value));   // small-table route: pad (elements) of the row pitch of dA^T / Z^T ("0": rows 2^18-aligned at the bench's size)
  } else if (strcmp(key, "bf16_gemm_regstage") == 0) {
    // bf16 split-K products on gx::k_gemm16r ("1": opt-in, operands global -> registers -> LDS, four chunks in flight per thread) or gx::k_gemm16x ("0", default: LDS-DMA, two)
    bf16p::set_gemm_regstage(atoi(value) != 0);
  } else if (strcmp(key, "bf16_gemm_touch") == 0) {
    // bf16 products on gx::k_gemm16x: every wave touches (one dword per tile row = one cache line) the chunk this many chunks ahead of its DMA: an L2 prefetch
    // for a launch that is bound by its operand staging ("0": off).  Process-wide.
    bf16p::set_gemm_touch(atoi(value));
  } else if (strcmp(key, "bf16_gemm_pingpong") == 0) {
    // bf16 pipeline: split-K products with >= 256 x 256 outputs on the two-group kernel gx::k_gemm16p ("1": opt-in, measured slower) or on gx::k_gemm16x ("0", default).  Process-wide.
    bf16p::set_gemm_pingpong(atoi(value) != 0);
  } else if (strcmp(key, "bf16_bptt_dxe") == 0) {
    // bf16 pipeline with "bf16_small_tables": the entity slice of dx is formed INSIDE the persistent BPTT launch (a fourth result tile per wave; value =
    // depth of its weight ring, 8 (default) or 16) or by its own product launch reading dA^T once more ("0")
    const int v = atoi(value);
    KPRN_REQUIRE(v == 0 || v == 8 || v == 16, KPRN_E_ARG, "bf16_bptt_dxe must be 0, 8 or 16");
    h->bf16_bptt_dxe = v;
  } else if (strcmp(key, "bf16_small_tables") == 0) {
    // bf16 pipeline, persistent BPTT: gradients of the type / relation tables (<= 128 rows together) and of their column blocks of W_i2g from one
    // extra column block of the merged dW product ("1", default) or from the full dx product + the table-gradient launch ("0": the A/B reference)
    h->bf16_small_tables = atoi(value) != 0;
  } else if (strcmp(key, "score_rest_before_bptt") == 0) {
    join_score(h);
    h->score_rest_before_bptt = atoi(value) ? 1 : 0;
  } else if (strcmp(key, "score_rest_in_backward") == 0) {
    join_score(h);
    h->score_rest_in_backward = atoi(value) ? 1 : 0;
    h->after_bptt_hook = h->score_rest_in_backward ? score_rest_hook : nullptr;
  } else if (strcmp(key, "score_split") == 0) {
    // kprn_forward_batch_async queues only the first (1 - f) of the batch's tiles; kprn_forward_batch_async_rest the others + the pooling
    join_score(h);
    const double f = atof(value);
    KPRN_REQUIRE(f >= 0.0 && f < 1.0, KPRN_E_ARG, "score_split: a fraction in [0, 1)");
    h->score_split = (float)f;
  } else if (strcmp(key, "score_overlap") == 0) {
    // kprn_forward_batch_async on a second stream (fused path): the scoring pass shares the chip with the work enqueued after it
    join_score(h);
    h->score_overlap = atoi(value) ? 1 : 0;
  } else if (strcmp(key, "loss_accumulate") == 0) {
    h->loss_accumulate = atoi(value) ? 1 : 0;
  } else if (strcmp(key, "inline_upload") == 0) {
    if (strcmp(value, "side") == 0) h->inline_upload_side = 1;
    else if (strcmp(value, "main") == 0) h->inline_upload_side = 0;
    else throw KprnError{KPRN_E_ARG, "inline_upload must be side or main"};
  } else if (strcmp(key, "train_step_return") == 0) {
    if (strcmp(value, "loss") == 0) h->train_step_drain = 0;
    else if (strcmp(value, "drain") == 0) h->train_step_drain = 1;
    else throw KprnError{KPRN_E_ARG, "train_step_return must be loss or drain"};
  } else if (strcmp(key, "feed_build") == 0) {
    if (strcmp(value, "host") == 0) h->feed_build_host = 1;
    else if (strcmp(value, "device") == 0) h->feed_build_host = 0;
    else throw KprnError{KPRN_E_ARG, "feed_build must be host or device"};
  } else if (strcmp(key, "feed_threads") == 0) {
    const int v = atoi(value);
    KPRN_REQUIRE(v >= 1 && v <= 64, KPRN_E_ARG, "feed_threads must be in 1..64");
    h->feed_threads = v;
  } else if (strcmp(key, "feed_workers") == 0) {
    const int v = atoi(value);
    KPRN_REQUIRE(v >= 1 && v <= 64 && !h->feed_pool, KPRN_E_ARG, "feed_workers must be in 1..64 and set before the first feed");
    h->feed_workers = v;
  } else if (strcmp(key, "profile_filter") == 0) {
    h->prof_filter = value;  // "" = every kernel family; else only families whose name starts with this
  } else if (strcmp(key, "dp_comm_stream") == 0) {
    // kprn_dp_exchange_begin queues the all-gather on a stream of its own (1) or on the main stream (0, default): with 1, work queued on the
    // main stream between _begin and _finish (a scoring pass on the side stream forks from there) runs while the collective is in flight
    h->dp_comm_stream_on = atoi(value) != 0;
  } else if (strcmp(key, "dp_fused_update") == 0) {
    h->dp_fused_update = atoi(value) != 0;
  } else if (strcmp(key, "dp_dense_in_pack") == 0) {
    // data-parallel exchange: the dense gradient arena travels behind the packed entity rows (one all-gather, no all-reduce); the merge sums
    // the ranks' copies in rank order
    h->dp_dense_in_pack = atoi(value) != 0;
  } else if (strcmp(key, "reserve_cus") == 0) {
    // the fused SCORING forward is a persistent one-workgroup-per-CU kernel that fills the register file of every CU it runs
    // on; leaving a few CUs free lets the copy kernels of a concurrently running collective (RCCL) make progress beside it
    const int v = atoi(value);
    KPRN_REQUIRE(v >= 0 && v <= 128, KPRN_E_ARG, "reserve_cus must be in 0..128");
    h->reserve_cus = v;
  } else {
    throw KprnError{KPRN_E_ARG, std::string("unknown option: ") + key};
  }
  API_END(h)
}

}  // extern "C"
