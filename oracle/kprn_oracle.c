/*
 * kprn_oracle.c -- CPU restatement of the eBay/KPRN songPathRnn hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under kprn_amd/ may import, link or call this
 * file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / reported CPU baseline -- never as the product path.
 *
 * PARITY UNPINNED.  The reference (Lua/Torch7) cannot run in this image (no th /
 * luajit / libTH) and ships no golden activations, losses or gradients for this path
 * (SURVEY.md section 8c).  The in-tree logic is restated line by line; the out-of-tree
 * arithmetic (Torch7 nn / Element-Research rnn / optim, no version pinned by the
 * reference) is restated from its published behaviour.  Each such assumption is
 * marked [A#] below so a reviewer with a Torch7 install can falsify it:
 *   [A1] nn.LookupTable = plain row gather; backward = scatter-add (duplicates add).
 *   [A2] nn.FastLSTM: a = i2g.W x + i2g.b + o2g.W h (o2g has NO bias); a viewed
 *        [4,H]; chunks in memory order = input gate (sigmoid), candidate (tanh),
 *        forget gate (sigmoid), output gate (sigmoid); c = f*c' + i*g; h = o*tanh(c);
 *        h0 = c0 = 0 at every forward (Sequencer remember 'neither').
 *   [A3] nn.BCECriterion: eps = 1e-12, sizeAverage = true:
 *        loss = -(1/B) sum t log(p+eps) + (1-t) log(1-p+eps)
 *        dp   = -(t-p) / ((1-p+eps)(p+eps)) / B
 *   [A4] optim.adam: t+=1; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g*g;
 *        denom = sqrt(v) + eps; step = lr*sqrt(1-b2^t)/(1-b1^t); x -= step*m/denom.
 *   [A5] optim.adagrad: clr = lr/(1+nevals*lrd); G += g*g; x -= clr*g/(sqrt(G)+1e-10).
 *   [A7] rnnType "rnn" (OneModel.lua:240-266): h_t = act(i2h.W x_t + i2h.b + h2h.W h_{t-1} + h2h.b), h_0 = 0,
 *        act = ReLU when -useReLU 1 else Tanh (OneModel.lua:225-229), wrapped in nn.MaskZero(rm, 1): rows of the
 *        output whose INPUT row x_t (layer > 1: h^{l-1}_t) is all zeros are zeroed, and so are their gradients
 *        (pad steps have zero embeddings after zeroPadTokens: the state stays 0 until the first real step).
 *        parameters() order per layer: i2h.weight[H,D_l], i2h.bias[H], h2h.weight[H,H], h2h.bias[H].
 *   [A8] rnnType "gru" (OneModel.lua:237-238, nn.GRU of Element-Research/rnn, the 2016 graph with nn.SAdd(-1,true)):
 *        [r; z] = sigmoid(i2g.W x + i2g.b + o2g.W h')   (i2g = nn.Linear(D,2H), o2g = nn.LinearNoBias(H,2H); first H rows = reset r)
 *        n = tanh(c_i2h.W x + c_i2h.b + c_h2h.W (r * h'))   (the reset gate is applied BEFORE the candidate's recurrent product)
 *        h = (1 - z) * n + z * h';  h_0 = 0;  no MaskZero.
 *        parameters() order per layer: i2g.weight[2H,D_l], i2g.bias[2H], o2g.weight[2H,H], c_i2h.weight[H,D_l], c_i2h.bias[H], c_h2h.weight[H,H].
 *   [A6] getParameters() flat order = module traversal order:
 *        Wt | We | Wr | (i2g.W[4H,D_l], i2g.b[4H], o2g.W[4H,H]) x L | out.W[C,H] | out.b[C]
 * The restatement is cross-checked against an independent implementation (PyTorch CPU
 * autograd with permuted gates) and finite differences in tests/test_oracle.py.
 *
 * Reference lines followed (all under /root/reference/release/songPathRnn/):
 *   model/module/MapReduce.lua:20-50,74-85      view [B,P,..]->[B*P,..], map, reduce dim 2
 *   model/net/FeatureEmbedding.lua:26-56,83-89,112-121   type/entity/relation lookup + JoinTable(3)
 *   model/OneModel.lua:223-275                  SplitTable(3)->embedding->SplitTable(2)->Sequencer(FastLSTM)xL
 *                                               ->SelectTable(-1)->Linear(H,46)
 *   model/OneModel.lua:284-294                  reducer: Max / TopK+Mean / LogSumExp ; Sigmoid
 *   model/module/LogSumExp.lua:13-36            stable LSE forward, softmax*g backward
 *   model/module/TopK.lua:17-38                 topk forward, scatter backward
 *   model/optimizer/MyOptimizer.lua:74-93       zeroPadTokens (row vocabSize, 1-based)
 *   model/optimizer/MyOptimizer.lua:177-221     trainBatch: zeroPad; zeroGrad; fwd; BCE; bwd;
 *                                               iff regularize==1 {clip to gradClipNorm; g += l2*theta};
 *                                               optim step; zeroPad
 *   model/optimizer/MyOptimizer.lua:126         Select(2, classId)
 *   eval/test_from_checkpoint.lua:68-118        scoring: forward, Select(2,1), "%.5f"
 *
 * Built twice (REAL=double -> *_f64, REAL=float -> *_f32) into oracle/libkprn_oracle.so.
 * The CPU reference computes in float64; the f32 build exists to separate
 * rounding drift from logic errors when checking the fp32 HIP path.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL double
#endif
#ifndef SUFFIX
#define SUFFIX _f64
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

typedef struct {
  int32_t Vt, Ve, Vr;     /* vocab sizes (rows of the three tables)                 */
  int32_t dt, de, dr;     /* embedding dims                                         */
  int32_t F;              /* numFeatureTemplates (cols per step)                    */
  int32_t numTypes;       /* numEntityTypes: type cols are F-numTypes-1 .. F-2 (1-based), FeatureEmbedding.lua:51 */
  int32_t H, L, C;        /* rnnHidSize, numLayers, labelDimension (46)             */
  int32_t reducer;        /* 0 = Max, 1 = TopK+Mean, 2 = LogSumExp (OneModel.lua:284-293) */
  int32_t K;              /* topK K                                                  */
  int32_t rnn_type;       /* 0 = lstm (nn.FastLSTM), 1 = rnn (Recurrence + MaskZero) [A7], 2 = gru (nn.GRU) [A8] */
  int32_t use_relu;       /* rnn: -useReLU 1 -> ReLU, else Tanh                      */
} okprn_cfg;

typedef struct {
  int32_t method;         /* 0 = adagrad, 1 = adam (OneModel.lua:347-361)            */
  double lr, beta1, beta2, eps, lr_decay;
  int32_t regularize, use_grad_clip;
  double grad_clip_norm, l2;
  int32_t bce_literal;    /* 1 = literal Torch BCE backward through sigmoid; 0 = fused (p-t)/B */
} okprn_opt;

static inline REAL sigm(REAL x) { return (REAL)1 / ((REAL)1 + (REAL)exp(-(double)x)); }

/* ---- flat parameter layout [A6] ------------------------------------------------ */
typedef struct {
  size_t Wt, We, Wr;
  size_t i2gW[8], i2gb[8], o2gW[8], h2hb[8];  /* rnn: i2gW = i2h.W, i2gb = i2h.b, o2gW = h2h.W, h2hb = h2h.b */
  size_t ciW[8], cib[8], chW[8];              /* gru: candidate c_i2h.W, c_i2h.b, c_h2h.W */
  int G;  /* rows of the recurrent weights per hidden unit: 4 (lstm gates) or 1 (rnn) */
  size_t outW, outb, total;
  int D;
} layout_t;

static layout_t make_layout(const okprn_cfg* c) {
  layout_t l;
  size_t o = 0;
  l.D = c->dt + c->de + c->dr;
  l.Wt = o; o += (size_t)c->Vt * c->dt;
  l.We = o; o += (size_t)c->Ve * c->de;
  l.Wr = o; o += (size_t)c->Vr * c->dr;
  l.G = (c->rnn_type == 1) ? 1 : ((c->rnn_type == 2) ? 2 : 4);
  for (int i = 0; i < c->L; ++i) {
    int Din = (i == 0) ? l.D : c->H;
    l.i2gW[i] = o; o += (size_t)l.G * c->H * Din;
    l.i2gb[i] = o; o += (size_t)l.G * c->H;
    l.o2gW[i] = o; o += (size_t)l.G * c->H * c->H;
    l.h2hb[i] = o; if (c->rnn_type == 1) o += (size_t)c->H;
    l.ciW[i] = o; if (c->rnn_type == 2) o += (size_t)c->H * Din;
    l.cib[i] = o; if (c->rnn_type == 2) o += (size_t)c->H;
    l.chW[i] = o; if (c->rnn_type == 2) o += (size_t)c->H * c->H;
  }
  l.outW = o; o += (size_t)c->C * c->H;
  l.outb = o; o += (size_t)c->C;
  l.total = o;
  return l;
}

size_t FN(okprn_num_params)(const okprn_cfg* c) { return make_layout(c).total; }

/* offsets of every named parameter, for the tests: out[0..2]=Wt,We,Wr; then 3 per
 * layer (rnn: 4 per layer); then outW,outb,total */
void FN(okprn_layout)(const okprn_cfg* c, int64_t* out) {
  layout_t l = make_layout(c);
  int k = 0;
  out[k++] = (int64_t)l.Wt; out[k++] = (int64_t)l.We; out[k++] = (int64_t)l.Wr;
  for (int i = 0; i < c->L; ++i) {
    out[k++] = (int64_t)l.i2gW[i]; out[k++] = (int64_t)l.i2gb[i]; out[k++] = (int64_t)l.o2gW[i];
    if (c->rnn_type == 1) out[k++] = (int64_t)l.h2hb[i];
    if (c->rnn_type == 2) { out[k++] = (int64_t)l.ciW[i]; out[k++] = (int64_t)l.cib[i]; out[k++] = (int64_t)l.chW[i]; }
  }
  out[k++] = (int64_t)l.outW; out[k++] = (int64_t)l.outb; out[k++] = (int64_t)l.total;
}

/* ---- embedding: x[D] = [ sum_k Wt[type_k] | We[ent] | Wr[rel] ] ------------------
 * FeatureEmbedding.lua:36-56 (NarrowTable(F-numTypes-1, numTypes) + shared LookupTables +
 * CAddTable), :83-89 (SelectTable(F-1)), :26-34 (SelectTable(-1)), :118-119 (cat order
 * type, entity, relation; JoinTable(3)).  ids are 1-based (int2torch.lua:60-63). */
static void embed_step(const okprn_cfg* c, const layout_t* l, const REAL* th,
                       const int32_t* feat /*[F]*/, REAL* x /*[D]*/) {
  const int F = c->F, nT = c->numTypes;
  for (int k = 0; k < nT; ++k) { /* CAddTable: first slot copied, the rest added in slot order */
    int32_t id = feat[F - nT - 2 + k]; /* 1-based col F-nT-1+k -> 0-based F-nT-2+k */
    const REAL* row = th + l->Wt + (size_t)(id - 1) * c->dt;
    if (k == 0) for (int j = 0; j < c->dt; ++j) x[j] = row[j];
    else for (int j = 0; j < c->dt; ++j) x[j] += row[j];
  }
  {
    const REAL* row = th + l->We + (size_t)(feat[F - 2] - 1) * c->de;
    for (int j = 0; j < c->de; ++j) x[c->dt + j] = row[j];
  }
  {
    const REAL* row = th + l->Wr + (size_t)(feat[F - 1] - 1) * c->dr;
    for (int j = 0; j < c->dr; ++j) x[c->dt + c->de + j] = row[j];
  }
}

/* per-path activation record: for each (t,l): i,g,f,o,c,h  (6H), plus x per t (D) */
typedef struct {
  REAL* x;    /* [T][D]        */
  REAL* act;  /* [T][L][6][H]  */
} pathrec_t;

/* FastLSTM stack over T steps, h0=c0=0 [A2]; returns pointer to h_T of the top layer */
static const REAL* path_forward(const okprn_cfg* c, const layout_t* l, const REAL* th,
                                const int32_t* steps /*[T][F]*/, int T, pathrec_t* r, REAL* a /*[4H]*/) {
  const int H = c->H, L = c->L, D = l->D;
  for (int t = 0; t < T; ++t) {
    REAL* x = r->x + (size_t)t * D;
    embed_step(c, l, th, steps + (size_t)t * c->F, x);
    const REAL* in = x;
    int Din = D;
    for (int ly = 0; ly < L; ++ly) {
      const REAL* Wi = th + l->i2gW[ly];
      const REAL* bi = th + l->i2gb[ly];
      const REAL* Wo = th + l->o2gW[ly];
      REAL* cur = r->act + ((size_t)t * L + ly) * 6 * H;
      const REAL* prev = (t > 0) ? r->act + ((size_t)(t - 1) * L + ly) * 6 * H : NULL;
      if (c->rnn_type == 2) { /* [A8] nn.GRU: record slots 0 = r, 1 = z, 2 = n (candidate), 3 = r*h', 5 = h */
        const REAL* Wc = th + l->ciW[ly];
        const REAL* bc = th + l->cib[ly];
        const REAL* Uc = th + l->chW[ly];
        const REAL* hp = prev ? prev + 5 * H : NULL;
        for (int n = 0; n < 2 * H; ++n) {
          REAL s = bi[n];
          const REAL* w = Wi + (size_t)n * Din;
          for (int k = 0; k < Din; ++k) s += w[k] * in[k];
          if (hp) { const REAL* w2 = Wo + (size_t)n * H; for (int k = 0; k < H; ++k) s += w2[k] * hp[k]; }
          cur[n] = sigm(s);
        }
        for (int j = 0; j < H; ++j) cur[3 * H + j] = hp ? cur[j] * hp[j] : (REAL)0;
        for (int n = 0; n < H; ++n) {
          REAL s = bc[n];
          const REAL* w = Wc + (size_t)n * Din;
          for (int k = 0; k < Din; ++k) s += w[k] * in[k];
          if (hp) { const REAL* w2 = Uc + (size_t)n * H; for (int k = 0; k < H; ++k) s += w2[k] * cur[3 * H + k]; }
          REAL nn_ = (REAL)tanh((double)s);
          REAL z = cur[H + n];
          cur[2 * H + n] = nn_;
          cur[5 * H + n] = ((REAL)1 - z) * nn_ + z * (hp ? hp[n] : (REAL)0);
        }
        in = cur + 5 * H;
        Din = H;
        continue;
      }
      if (c->rnn_type == 1) { /* [A7] Recurrence(MaskZero(act(i2h x + h2h h'))) */
        const REAL* bh = th + l->h2hb[ly];
        int nonzero = 0;
        for (int k = 0; k < Din; ++k) if (in[k] != (REAL)0) { nonzero = 1; break; }
        for (int n = 0; n < H; ++n) {
          REAL s = bi[n] + bh[n];  /* nn.Linear(h2h) of the zero initial state still adds its bias */
          const REAL* w = Wi + (size_t)n * Din;
          for (int k = 0; k < Din; ++k) s += w[k] * in[k];
          if (prev) {
            const REAL* hp = prev + 5 * H;
            const REAL* w2 = Wo + (size_t)n * H;
            for (int k = 0; k < H; ++k) s += w2[k] * hp[k];
          }
          REAL hh = c->use_relu ? (s > (REAL)0 ? s : (REAL)0) : (REAL)tanh((double)s);
          if (!nonzero) hh = (REAL)0;
          cur[n] = (REAL)nonzero;  /* slot 0: the MaskZero flag of this step */
          cur[H + n] = s;          /* slot 1: pre-activation */
          cur[5 * H + n] = hh;
        }
        in = cur + 5 * H;
        Din = H;
        continue;
      }
      for (int n = 0; n < 4 * H; ++n) {
        REAL s = bi[n];
        const REAL* w = Wi + (size_t)n * Din;
        for (int k = 0; k < Din; ++k) s += w[k] * in[k];
        if (prev) {
          const REAL* hp = prev + 5 * H;
          const REAL* w2 = Wo + (size_t)n * H;
          for (int k = 0; k < H; ++k) s += w2[k] * hp[k];
        }
        a[n] = s;
      }
      for (int j = 0; j < H; ++j) {
        REAL ig = sigm(a[j]);
        REAL gg = (REAL)tanh((double)a[H + j]);
        REAL fg = sigm(a[2 * H + j]);
        REAL og = sigm(a[3 * H + j]);
        REAL cp = prev ? prev[4 * H + j] : (REAL)0;
        REAL cc = fg * cp + ig * gg;
        REAL hh = og * (REAL)tanh((double)cc);
        cur[j] = ig; cur[H + j] = gg; cur[2 * H + j] = fg; cur[3 * H + j] = og;
        cur[4 * H + j] = cc; cur[5 * H + j] = hh;
      }
      in = cur + 5 * H;
      Din = H;
    }
  }
  return r->act + ((size_t)(T - 1) * L + (L - 1)) * 6 * H + 5 * H;
}

/* BPTT for one path given ds[C] (grad wrt the head outputs). Accumulates into g (flat).
 * dh/dc scratch: [L][H] each, da [4H], dxin [max(D,H)] */
static void path_backward(const okprn_cfg* c, const layout_t* l, const REAL* th, REAL* g /* embedding tables: flat base */,
                          REAL* gd /* dense params: gd[off - l->i2gW[0]] */,
                          const int32_t* steps, int T, const pathrec_t* r, const REAL* ds,
                          REAL* dh /*[L][H]*/, REAL* dc /*[L][H]*/, REAL* da /*[4H]*/, REAL* dxin, int atomic_emb) {
  const int H = c->H, L = c->L, D = l->D, C = c->C, F = c->F, nT = c->numTypes;
  const size_t doff = l->i2gW[0];
  memset(dh, 0, sizeof(REAL) * L * H);
  memset(dc, 0, sizeof(REAL) * L * H);
  /* Linear head (OneModel.lua:275): s = W hT + b */
  {
    const REAL* hT = r->act + ((size_t)(T - 1) * L + (L - 1)) * 6 * H + 5 * H;
    const REAL* W = th + l->outW;
    REAL* gW = gd + (l->outW - doff);
    REAL* gb = gd + (l->outb - doff);
    REAL* dht = dh + (size_t)(L - 1) * H;
    for (int k = 0; k < C; ++k) {
      REAL d = ds[k];
      if (d == 0) continue;
      gb[k] += d;
      for (int j = 0; j < H; ++j) { gW[(size_t)k * H + j] += d * hT[j]; dht[j] += d * W[(size_t)k * H + j]; }
    }
  }
  for (int t = T - 1; t >= 0; --t) {
    for (int ly = L - 1; ly >= 0; --ly) {
      const REAL* cur = r->act + ((size_t)t * L + ly) * 6 * H;
      const REAL* prev = (t > 0) ? r->act + ((size_t)(t - 1) * L + ly) * 6 * H : NULL;
      const int Din = (ly == 0) ? D : H;
      const REAL* in = (ly == 0) ? r->x + (size_t)t * D : r->act + ((size_t)t * L + ly - 1) * 6 * H + 5 * H;
      REAL* dhl = dh + (size_t)ly * H;
      REAL* dcl = dc + (size_t)ly * H;
      if (c->rnn_type == 2) { /* [A8] */
        const REAL* Wi = th + l->i2gW[ly];
        const REAL* Wo = th + l->o2gW[ly];
        const REAL* Wc = th + l->ciW[ly];
        const REAL* Uc = th + l->chW[ly];
        REAL* gWi = gd + (l->i2gW[ly] - doff);
        REAL* gbi = gd + (l->i2gb[ly] - doff);
        REAL* gWo = gd + (l->o2gW[ly] - doff);
        REAL* gWc = gd + (l->ciW[ly] - doff);
        REAL* gbc = gd + (l->cib[ly] - doff);
        REAL* gUc = gd + (l->chW[ly] - doff);
        const REAL* hp = prev ? prev + 5 * H : NULL;
        REAL* dac = da;           /* [H]  candidate pre-activation grad */
        REAL* dag = da + H;       /* [2H] gate pre-activation grads (r then z) */
        REAL* drh = dcl;          /* [H]  scratch: grad wrt (r * h') -- the lstm cell-state slot is free for gru */
        for (int k = 0; k < Din; ++k) dxin[k] = 0;
        for (int j = 0; j < H; ++j) {
          REAL z = cur[H + j], nn_ = cur[2 * H + j], dhh = dhl[j];
          REAL hpj = hp ? hp[j] : (REAL)0;
          dac[j] = dhh * ((REAL)1 - z) * ((REAL)1 - nn_ * nn_);
          dag[H + j] = dhh * (hpj - nn_) * z * ((REAL)1 - z);
          dhl[j] = dhh * z;   /* -> dh' (direct path) */
          drh[j] = 0;
        }
        for (int n = 0; n < H; ++n) {
          REAL d = dac[n];
          gbc[n] += d;
          const REAL* w = Wc + (size_t)n * Din;
          REAL* gw = gWc + (size_t)n * Din;
          for (int k = 0; k < Din; ++k) { gw[k] += d * in[k]; dxin[k] += d * w[k]; }
          if (hp) {
            const REAL* w2 = Uc + (size_t)n * H;
            REAL* gw2 = gUc + (size_t)n * H;
            for (int k = 0; k < H; ++k) { gw2[k] += d * cur[3 * H + k]; drh[k] += d * w2[k]; }
          }
        }
        for (int j = 0; j < H; ++j) {
          REAL r = cur[j], hpj = hp ? hp[j] : (REAL)0;
          dag[j] = drh[j] * hpj * r * ((REAL)1 - r);
          dhl[j] += drh[j] * r;
        }
        for (int n = 0; n < 2 * H; ++n) {
          REAL d = dag[n];
          gbi[n] += d;
          const REAL* w = Wi + (size_t)n * Din;
          REAL* gw = gWi + (size_t)n * Din;
          for (int k = 0; k < Din; ++k) { gw[k] += d * in[k]; dxin[k] += d * w[k]; }
          if (hp) {
            const REAL* w2 = Wo + (size_t)n * H;
            REAL* gw2 = gWo + (size_t)n * H;
            for (int k = 0; k < H; ++k) { gw2[k] += d * hp[k]; dhl[k] += d * w2[k]; }
          }
        }
        for (int j = 0; j < H; ++j) dcl[j] = 0;
        goto below;
      }
      if (c->rnn_type == 1) { /* [A7] */
        const REAL* Wi = th + l->i2gW[ly];
        const REAL* Wo = th + l->o2gW[ly];
        REAL* gWi = gd + (l->i2gW[ly] - doff);
        REAL* gbi = gd + (l->i2gb[ly] - doff);
        REAL* gWo = gd + (l->o2gW[ly] - doff);
        REAL* gbh = gd + (l->h2hb[ly] - doff);
        const int nonzero = cur[0] != (REAL)0;
        for (int j = 0; j < H; ++j) {
          REAL hh = cur[5 * H + j], pre = cur[H + j];
          REAL der = c->use_relu ? (pre > (REAL)0 ? (REAL)1 : (REAL)0) : ((REAL)1 - hh * hh);
          da[j] = nonzero ? dhl[j] * der : (REAL)0;  /* MaskZero: masked rows pass no gradient */
          dhl[j] = 0;
        }
        for (int k = 0; k < Din; ++k) dxin[k] = 0;
        for (int n = 0; n < H; ++n) {
          REAL d = da[n];
          gbi[n] += d;
          gbh[n] += d;
          const REAL* w = Wi + (size_t)n * Din;
          REAL* gw = gWi + (size_t)n * Din;
          for (int k = 0; k < Din; ++k) { gw[k] += d * in[k]; dxin[k] += d * w[k]; }
          if (prev) {
            const REAL* hp = prev + 5 * H;
            const REAL* w2 = Wo + (size_t)n * H;
            REAL* gw2 = gWo + (size_t)n * H;
            for (int k = 0; k < H; ++k) { gw2[k] += d * hp[k]; dhl[k] += d * w2[k]; }
          }
        }
        goto below;
      }
      for (int j = 0; j < H; ++j) {
        REAL ig = cur[j], gg = cur[H + j], fg = cur[2 * H + j], og = cur[3 * H + j], cc = cur[4 * H + j];
        REAL tc = (REAL)tanh((double)cc);
        REAL dhh = dhl[j];
        REAL dO = dhh * tc;
        REAL dC = dcl[j] + dhh * og * ((REAL)1 - tc * tc);
        REAL cp = prev ? prev[4 * H + j] : (REAL)0;
        da[j] = dC * gg * ig * ((REAL)1 - ig);
        da[H + j] = dC * ig * ((REAL)1 - gg * gg);
        da[2 * H + j] = dC * cp * fg * ((REAL)1 - fg);
        da[3 * H + j] = dO * og * ((REAL)1 - og);
        dcl[j] = dC * fg; /* -> dc_{t-1} */
        dhl[j] = 0;       /* will be refilled with dh_{t-1} below */
      }
      {
      const REAL* Wi = th + l->i2gW[ly];
      const REAL* Wo = th + l->o2gW[ly];
      REAL* gWi = gd + (l->i2gW[ly] - doff);
      REAL* gbi = gd + (l->i2gb[ly] - doff);
      REAL* gWo = gd + (l->o2gW[ly] - doff);
      for (int k = 0; k < Din; ++k) dxin[k] = 0;
      for (int n = 0; n < 4 * H; ++n) {
        REAL d = da[n];
        gbi[n] += d;
        const REAL* w = Wi + (size_t)n * Din;
        REAL* gw = gWi + (size_t)n * Din;
        for (int k = 0; k < Din; ++k) { gw[k] += d * in[k]; dxin[k] += d * w[k]; }
        if (prev) {
          const REAL* hp = prev + 5 * H;
          const REAL* w2 = Wo + (size_t)n * H;
          REAL* gw2 = gWo + (size_t)n * H;
          for (int k = 0; k < H; ++k) { gw2[k] += d * hp[k]; dhl[k] += d * w2[k]; }
        }
      }
      }
    below:
      if (ly > 0) {
        REAL* dbelow = dh + (size_t)(ly - 1) * H;
        for (int k = 0; k < H; ++k) dbelow[k] += dxin[k];
      } else {
        /* scatter-add into the three tables [A1]; type slots all receive the type slice */
        const int32_t* feat = steps + (size_t)t * F;
        for (int k = 0; k < nT; ++k) {
          REAL* row = g + l->Wt + (size_t)(feat[F - nT - 2 + k] - 1) * c->dt;
          for (int j = 0; j < c->dt; ++j) {
            if (atomic_emb) {
#pragma omp atomic
              row[j] += dxin[j];
            } else row[j] += dxin[j];
          }
        }
        {
          REAL* row = g + l->We + (size_t)(feat[F - 2] - 1) * c->de;
          for (int j = 0; j < c->de; ++j) {
            if (atomic_emb) {
#pragma omp atomic
              row[j] += dxin[c->dt + j];
            } else row[j] += dxin[c->dt + j];
          }
        }
        {
          REAL* row = g + l->Wr + (size_t)(feat[F - 1] - 1) * c->dr;
          for (int j = 0; j < c->dr; ++j) {
            if (atomic_emb) {
#pragma omp atomic
              row[j] += dxin[c->dt + c->de + j];
            } else row[j] += dxin[c->dt + c->de + j];
          }
        }
      }
    }
  }
}

/* reducer over the P paths of one pair, for every class column.
 * s: [P][C] -> y[C]; also fills w[P][C] = d y[c] / d s[p][c]  (LogSumExp.lua:13-36,
 * TopK.lua:17-38 + nn.Mean(2), nn.Max(2)). */
static void reduce_pair(const okprn_cfg* c, const REAL* s, int P, REAL* y, REAL* w) {
  const int C = c->C;
  for (int k = 0; k < C; ++k) {
    if (c->reducer == 2) {
      REAL m = s[k];
      for (int p = 1; p < P; ++p) if (s[(size_t)p * C + k] > m) m = s[(size_t)p * C + k];
      REAL sum = 0;
      for (int p = 0; p < P; ++p) { REAL e = (REAL)exp((double)(s[(size_t)p * C + k] - m)); if (w) w[(size_t)p * C + k] = e; sum += e; }
      y[k] = (REAL)log((double)sum) + m;
      if (w) for (int p = 0; p < P; ++p) w[(size_t)p * C + k] /= sum;
    } else if (c->reducer == 0) {
      int arg = 0;
      for (int p = 1; p < P; ++p) if (s[(size_t)p * C + k] > s[(size_t)arg * C + k]) arg = p;
      y[k] = s[(size_t)arg * C + k];
      if (w) for (int p = 0; p < P; ++p) w[(size_t)p * C + k] = (p == arg) ? (REAL)1 : (REAL)0;
    } else {
      /* TopK(k,2) then Mean(2): k = min(K,P) largest, first index wins ties */
      int kk = c->K < P ? c->K : P;
      REAL acc = 0;
      if (w) for (int p = 0; p < P; ++p) w[(size_t)p * C + k] = 0;
      /* selection by repeated argmax over not-yet-taken entries */
      unsigned char taken[4096];
      int PP = P < 4096 ? P : 4096;
      memset(taken, 0, (size_t)PP);
      for (int q = 0; q < kk; ++q) {
        int arg = -1;
        for (int p = 0; p < PP; ++p) if (!taken[p] && (arg < 0 || s[(size_t)p * C + k] > s[(size_t)arg * C + k])) arg = p;
        taken[arg] = 1;
        acc += s[(size_t)arg * C + k];
        if (w) w[(size_t)arg * C + k] = (REAL)1 / (REAL)kk;
      }
      y[k] = acc / (REAL)kk;
    }
  }
}

static pathrec_t rec_alloc(const okprn_cfg* c, int D, int T) {
  pathrec_t r;
  r.x = (REAL*)malloc(sizeof(REAL) * (size_t)T * D);
  r.act = (REAL*)malloc(sizeof(REAL) * (size_t)T * c->L * 6 * c->H);
  return r;
}
static void rec_free(pathrec_t* r) { free(r->x); free(r->act); }

/* ---- public: embedding output only (bit-exact gather check) ---------------------- */
void FN(okprn_embed)(const okprn_cfg* c, const REAL* th, const int32_t* idx, int64_t N, int T, REAL* x /*[N][T][D]*/) {
  layout_t l = make_layout(c);
  for (int64_t n = 0; n < N; ++n)
    for (int t = 0; t < T; ++t)
      embed_step(c, &l, th, idx + ((size_t)n * T + t) * c->F, x + ((size_t)n * T + t) * l.D);
}

/* ---- public: forward (scoring) ---------------------------------------------------
 * idx [B][P][T][F] 1-based int32.  Outputs (any may be NULL):
 *   path_scores [B*P][C]  (mapper output, OneModel.lua:275)
 *   pooled      [B][C]    (reducer output, before Sigmoid)
 *   probs       [B][C]    (after Sigmoid, OneModel.lua:294)                          */
void FN(okprn_forward)(const okprn_cfg* c, const REAL* th, const int32_t* idx, int B, int P, int T,
                       REAL* path_scores, REAL* pooled, REAL* probs) {
  layout_t l = make_layout(c);
  const int C = c->C, H = c->H;
#pragma omp parallel
  {
    pathrec_t r = rec_alloc(c, l.D, T);
    REAL* a = (REAL*)malloc(sizeof(REAL) * 4 * H);
    REAL* s = (REAL*)malloc(sizeof(REAL) * (size_t)P * C);
    REAL* y = (REAL*)malloc(sizeof(REAL) * C);
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      for (int p = 0; p < P; ++p) {
        const int32_t* steps = idx + (((size_t)b * P + p) * T) * c->F;
        const REAL* hT = path_forward(c, &l, th, steps, T, &r, a);
        for (int k = 0; k < C; ++k) {
          REAL acc = th[l.outb + k];
          const REAL* w = th + l.outW + (size_t)k * H;
          for (int j = 0; j < H; ++j) acc += w[j] * hT[j];
          s[(size_t)p * C + k] = acc;
        }
      }
      if (path_scores) memcpy(path_scores + (size_t)b * P * C, s, sizeof(REAL) * (size_t)P * C);
      reduce_pair(c, s, P, y, NULL);
      for (int k = 0; k < C; ++k) {
        if (pooled) pooled[(size_t)b * C + k] = y[k];
        if (probs) probs[(size_t)b * C + k] = sigm(y[k]);
      }
    }
    rec_free(&r); free(a); free(s); free(y);
  }
}

/* ---- public: forward + BCE + backward (MyOptimizer.lua:186-195) -------------------
 * grad is ACCUMULATED into (caller zeroes it = zeroGradParameters).  classId 1-based.
 * Returns loss.  probs_out [B] = p[b] = sigmoid(pooled[b][classId]).               */
double FN(okprn_forward_backward)(const okprn_cfg* c, const REAL* th, const int32_t* idx, int B, int P, int T,
                                  const REAL* labels, int classId, int bce_literal, double inv_batch,
                                  REAL* grad, REAL* probs_out) {
  layout_t l = make_layout(c);
  const int C = c->C, H = c->H, L = c->L;
  const int cid = classId - 1;
  const REAL eps = (REAL)1e-12;
  const REAL invB = (REAL)((inv_batch > 0) ? inv_batch : 1.0 / (double)B);
  double loss = 0;
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  /* dense (non-embedding) grads are accumulated per thread, then summed in thread order */
  const size_t dense_off = l.i2gW[0];
  const size_t dense_n = l.total - dense_off;
  REAL* tg = NULL;
  if (nthreads > 1) tg = (REAL*)calloc((size_t)nthreads * dense_n, sizeof(REAL));
#pragma omp parallel reduction(+ : loss)
  {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    REAL* gd = tg ? tg + (size_t)tid * dense_n : grad + dense_off;
    pathrec_t* recs = (pathrec_t*)malloc(sizeof(pathrec_t) * (size_t)P);
    for (int p = 0; p < P; ++p) recs[p] = rec_alloc(c, l.D, T);
    REAL* a = (REAL*)malloc(sizeof(REAL) * 4 * H);
    REAL* s = (REAL*)malloc(sizeof(REAL) * (size_t)P * C);
    REAL* w = (REAL*)malloc(sizeof(REAL) * (size_t)P * C);
    REAL* y = (REAL*)malloc(sizeof(REAL) * C);
    REAL* ds = (REAL*)calloc((size_t)C, sizeof(REAL));
    REAL* dh = (REAL*)malloc(sizeof(REAL) * (size_t)L * H);
    REAL* dc = (REAL*)malloc(sizeof(REAL) * (size_t)L * H);
    REAL* da = (REAL*)malloc(sizeof(REAL) * 4 * H);
    int mx = l.D > H ? l.D : H;
    REAL* dxin = (REAL*)malloc(sizeof(REAL) * (size_t)mx);
#pragma omp for schedule(static)
    for (int b = 0; b < B; ++b) {
      for (int p = 0; p < P; ++p) {
        const int32_t* steps = idx + (((size_t)b * P + p) * T) * c->F;
        const REAL* hT = path_forward(c, &l, th, steps, T, &recs[p], a);
        for (int k = 0; k < C; ++k) {
          REAL acc = th[l.outb + k];
          const REAL* ww = th + l.outW + (size_t)k * H;
          for (int j = 0; j < H; ++j) acc += ww[j] * hT[j];
          s[(size_t)p * C + k] = acc;
        }
      }
      reduce_pair(c, s, P, y, w);
      REAL pr = sigm(y[cid]);
      REAL t = labels[b];
      if (probs_out) probs_out[b] = pr;
      /* BCE [A3] */
      loss += -(double)(t * (REAL)log((double)(pr + eps)) + ((REAL)1 - t) * (REAL)log((double)((REAL)1 - pr + eps))) * (double)invB;
      REAL dy;
      if (bce_literal) {
        REAL dp = -(t - pr) / (((REAL)1 - pr + eps) * (pr + eps)) * invB;
        dy = dp * pr * ((REAL)1 - pr); /* nn.Sigmoid backward */
      } else {
        dy = (pr - t) * invB;
      }
      for (int p = 0; p < P; ++p) {
        const int32_t* steps = idx + (((size_t)b * P + p) * T) * c->F;
        ds[cid] = w[(size_t)p * C + cid] * dy; /* only column classId gets gradient (Select) */
        path_backward(c, &l, th, grad, gd, steps, T, &recs[p], ds, dh, dc, da, dxin, nthreads > 1);
      }
    }
    for (int p = 0; p < P; ++p) rec_free(&recs[p]);
    free(recs); free(a); free(s); free(w); free(y); free(ds); free(dh); free(dc); free(da); free(dxin);
  }
  if (tg) { /* sum the per-thread dense grads in thread order (deterministic for a fixed thread count) */
    for (int k = 0; k < nthreads; ++k)
      for (size_t i = 0; i < dense_n; ++i) grad[dense_off + i] += tg[(size_t)k * dense_n + i];
    free(tg);
  }
  return loss;
}

/* ---- optimisers over the flat vector ------------------------------------------- */
void FN(okprn_adam)(int64_t n, REAL* x, const REAL* g, REAL* m, REAL* v, int64_t t /* already incremented */,
                    double lr, double b1, double b2, double eps) {
  double bc1 = 1.0 - pow(b1, (double)t), bc2 = 1.0 - pow(b2, (double)t);
  REAL step = (REAL)(lr * sqrt(bc2) / bc1);
  REAL B1 = (REAL)b1, B2 = (REAL)b2, E = (REAL)eps;
  REAL omb1 = (REAL)(1.0 - b1), omb2 = (REAL)(1.0 - b2);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    REAL gi = g[i];
    REAL mi = m[i] * B1 + omb1 * gi;
    REAL vi = v[i] * B2 + omb2 * gi * gi;
    m[i] = mi; v[i] = vi;
    REAL denom = (REAL)sqrt((double)vi) + E;
    x[i] = x[i] - step * (mi / denom);
  }
}

void FN(okprn_adagrad)(int64_t n, REAL* x, const REAL* g, REAL* G, int64_t nevals /* before increment */,
                       double lr, double lrd) {
  REAL clr = (REAL)(lr / (1.0 + (double)nevals * lrd));
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    REAL gi = g[i];
    REAL Gi = G[i] + gi * gi;
    G[i] = Gi;
    x[i] = x[i] - clr * gi / ((REAL)sqrt((double)Gi) + (REAL)1e-10);
  }
}

/* zeroPadTokens (MyOptimizer.lua:74-93): rows Vt, Vr, Ve (1-based = last rows) */
void FN(okprn_zero_pad)(const okprn_cfg* c, REAL* th) {
  layout_t l = make_layout(c);
  memset(th + l.Wt + (size_t)(c->Vt - 1) * c->dt, 0, sizeof(REAL) * c->dt);
  memset(th + l.Wr + (size_t)(c->Vr - 1) * c->dr, 0, sizeof(REAL) * c->dr);
  memset(th + l.We + (size_t)(c->Ve - 1) * c->de, 0, sizeof(REAL) * c->de);
}

/* one MyOptimizer:trainBatch (MyOptimizer.lua:177-221).  state1/state2: adam m,v or
 * adagrad G (state2 unused).  *step is optState.t / evalCounter, updated in place.
 * grad_scratch [nParams] is overwritten.  Returns the loss. */
double FN(okprn_train_step)(const okprn_cfg* c, REAL* th, REAL* grad_scratch, REAL* state1, REAL* state2,
                            int64_t* step, const okprn_opt* o, const int32_t* idx, int B, int P, int T,
                            const REAL* labels, int classId, REAL* probs_out) {
  layout_t l = make_layout(c);
  const int64_t n = (int64_t)l.total;
  FN(okprn_zero_pad)(c, th);                                   /* :181 */
  memset(grad_scratch, 0, sizeof(REAL) * (size_t)n);           /* :186 */
  double loss = FN(okprn_forward_backward)(c, th, idx, B, P, T, labels, classId, o->bce_literal, 0.0,
                                           grad_scratch, probs_out); /* :189-195 */
  if (o->regularize == 1) {                                    /* :196-214 */
    if (o->use_grad_clip) {
      double nn = 0;
      for (int64_t i = 0; i < n; ++i) nn += (double)grad_scratch[i] * (double)grad_scratch[i];
      nn = sqrt(nn);
      if (nn > o->grad_clip_norm) {
        REAL sc = (REAL)(o->grad_clip_norm / nn);
        for (int64_t i = 0; i < n; ++i) grad_scratch[i] *= sc;
      }
    }
    REAL l2 = (REAL)o->l2;
    for (int64_t i = 0; i < n; ++i) grad_scratch[i] += l2 * th[i];
  }
  if (o->method == 1) {                                        /* :218 optim.adam */
    *step += 1;
    FN(okprn_adam)(n, th, grad_scratch, state1, state2, *step, o->lr, o->beta1, o->beta2, o->eps);
  } else {
    FN(okprn_adagrad)(n, th, grad_scratch, state1, *step, o->lr, o->lr_decay);
    *step += 1;
  }
  FN(okprn_zero_pad)(c, th);                                   /* :219 */
  return loss;
}
