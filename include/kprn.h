/*
 * kprn.h -- C ABI of libkprn.so: the MI355X (gfx950) path-scoring engine behind the
 * songPathRnn training / scoring surface of eBay/KPRN.
 *
 * The reference has NO native boundary; the operator API this library replaces is the
 * Torch7 nn.Module protocol as consumed by exactly two callers:
 *   release/songPathRnn/model/optimizer/MyOptimizer.lua:177-221  (trainBatch: zeroGrad,
 *       forward, BCE, backward, clip/L2, optim step, zeroPadTokens)
 *   release/songPathRnn/eval/test_from_checkpoint.lua:98-118     (model:forward -> preds[i])
 * Each entry point below cites the reference lines it stands in for.  The binding a
 * maintainer adds on the reference side (LuaJIT ffi.cdef) is in INTEGRATION.md and
 * bindings/kprn.lua.
 *
 * Conventions
 *  - plain C, no torch types; every function returns 0 on success or a negative
 *    kprn_status; the message is available from kprn_last_error(h) (h may be NULL for
 *    a failed kprn_create).  Nothing throws or aborts across the ABI.
 *  - the caller owns every host buffer it passes; the library owns all device memory.
 *  - indices are int32, 1-BASED, row-major [B,P,T,F] exactly like the reference's
 *    data tensor (release/songPathRnn/model/batcher/Batcher.lua:51): F columns per step =
 *    numEntityTypes type ids, then entity id, then relation id
 *    (model/net/FeatureEmbedding.lua:51,88,31).  Out-of-range ids are an error
 *    (KPRN_E_INDEX), never undefined behaviour.
 *  - one handle <-> one GPU <-> one host thread at a time.  Work is queued on one HIP
 *    stream; functions that return results to host memory synchronise that stream,
 *    functions documented "async" do not.
 *  - parameters are named and ordered as nn.Module:getParameters() flattens them
 *    (MyOptimizer.lua:42):  type_emb[Vt,dt] | entity_emb[Ve,de] | relation_emb[Vr,dr] |
 *    lstm{l}.i2g.weight[4H,D_l] lstm{l}.i2g.bias[4H] lstm{l}.o2g.weight[4H,H] (l=1..L) |
 *    out.weight[C,H] | out.bias[C]          (host side: row-major fp32)
 *    FastLSTM gate order inside the 4H rows: input, candidate(tanh), forget, output.
 *    rnnType rnn: per layer rnn{l}.i2h.weight[H,D_l] | rnn{l}.i2h.bias[H] | rnn{l}.h2h.weight[H,H] | rnn{l}.h2h.bias[H]
 *    rnnType gru: per layer gru{l}.i2g.weight[2H,D_l] | gru{l}.i2g.bias[2H] | gru{l}.o2g.weight[2H,H] (rows: reset, update) |
 *                 gru{l}.c_i2h.weight[H,D_l] | gru{l}.c_i2h.bias[H] | gru{l}.c_h2h.weight[H,H]
 */
#ifndef KPRN_H
#define KPRN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kprn_handle kprn_handle;
typedef struct kprn_batch kprn_batch;

typedef enum {
  KPRN_OK = 0,
  KPRN_E_ARG = -1,          /* bad argument / shape / name                        */
  KPRN_E_INDEX = -2,        /* an id outside 1..V                                 */
  KPRN_E_DEVICE = -3,       /* HIP error (message has the HIP string)             */
  KPRN_E_UNSUPPORTED = -4,  /* valid reference option not built yet               */
  KPRN_E_IO = -5,
  KPRN_E_NOMEM = -6
} kprn_status;

/* mirrors the torch.CmdLine flags of model/OneModel.lua:27-87 that shape the graph */
typedef struct {
  int32_t Vt, Ve, Vr;        /* -entityTypeVocabSize -entityVocabSize -relationVocabSize        */
  int32_t dt, de, dr;        /* -entityTypeEmbeddingDim -entityEmbeddingDim -relationEmbeddingDim; dt = 0: -includeEntityTypes 0 (no type
                                table in x_t), de = 0: -includeEntity 0 (no entity table): OneModel.lua:207-219; generic pipeline */
  int32_t F;                 /* -numFeatureTemplates                                             */
  int32_t num_types;         /* -numEntityTypes                                                  */
  int32_t H;                 /* -rnnHidSize                                                      */
  int32_t L;                 /* -numLayers  (L>1 requires dt+de+dr == H, OneModel.lua:236,270)   */
  int32_t C;                 /* labelDimension, 46 in the reference (OneModel.lua:119)           */
  int32_t rnn_type;          /* -rnnType: 0 = lstm (nn.FastLSTM), 1 = rnn (nn.Recurrence + nn.MaskZero, OneModel.lua:240-266;
                                the shipped config.sh default), 2 = gru (nn.GRU, OneModel.lua:237-238); rnn and gru run on the
                                generic pipeline */
  int32_t use_relu;          /* -useReLU (rnn): 1 = nn.ReLU, else nn.Tanh (OneModel.lua:225-229)  */
  int32_t rnn_init;          /* -rnnInitialization (rnn): i2h / h2h weights <- eye, biases <- 0 (OneModel.lua:310-322) */
  int32_t compute_dtype;     /* 0 = f32 (exact fp32 MFMA; the reference's arithmetic type on GPU).
                                1 = bf16: the recurrent / head GEMMs multiply in bf16 (operands rounded to nearest even) and
                                    accumulate in f32; parameters, activations and the optimiser stay f32 (BASELINE configs[3]).
                                    Scoring at D = H = 64 takes the fused matrix-core forward, everything else the generic pipeline.
                                2 = f32x6: fp32 results from the bf16 matrix cores -- every fp32 operand is split exactly into
                                    three bf16 pieces and the six partial products of weight >= 2^-16 are accumulated in f32; the
                                    dropped products are below 2^-24, i.e. the error is that of an fp32 FMA chain or smaller.
                                    Fused path only (forward on the matrix cores, backward as for 0); held to the same parity
                                    tolerances as 0.
                                3 = f32x3: as 2 with two fp16 pieces (2 x 11 mantissa bits) of every power-of-two pre-scaled
                                    operand and three partial products: half the matrix instructions of 2; the operands keep 22
                                    of their 24 mantissa bits.  Same tolerances.  New options, not in the reference.            */
  int32_t reducer;           /* -topK: 0 = Max, 1 = TopK+Mean, 2 = LogSumExp (OneModel.lua:284-293) */
  int32_t K;                 /* -K                                                               */
  int32_t device_id;         /* HIP device ordinal                                               */
  int32_t rank, world;       /* data-parallel position; loss is scaled by the GLOBAL batch       */
  float param_init;          /* -paramInit: uniform(-a, a) over every parameter (OneModel.lua:306-309) */
  uint64_t seed;             /* init RNG seed (the reference leaves it unseeded, OneModel.lua:123) */
  void* stream;              /* hipStream_t the engine queues on.  NULL = the engine creates its own non-blocking stream (fetch it
                                with kprn_stream).  NOTE: the legacy default stream's handle is ALSO 0 -- a caller that means "my
                                default stream" (torch.cuda.current_stream() before any stream is set) must say
                                KPRN_STREAM_LEGACY_DEFAULT, or its own work is not ordered with the engine's.                    */
} kprn_config;
/* kprn_config.stream value for the legacy default (null) stream: the engine then queues on stream 0 instead of creating one */
#define KPRN_STREAM_LEGACY_DEFAULT ((void*)(intptr_t)-1)

/* mirrors optInfo / optConfig (OneModel.lua:340-384) */
typedef struct {
  int32_t method;            /* -useAdam: 1 = optim.adam, 0 = optim.adagrad                      */
  float lr;                  /* -learningRate                                                    */
  float beta1, beta2, eps;   /* 0.9, 0.999, -epsilon                                             */
  float lr_decay;            /* -learningRateDecay (adagrad)                                     */
  int32_t regularize;        /* -regularize: gates BOTH clip and L2 (MyOptimizer.lua:196)        */
  int32_t use_grad_clip;     /* -useGradClip                                                     */
  float grad_clip_norm;      /* -gradClipNorm                                                    */
  float l2;                  /* -l2                                                              */
  int32_t bce_literal;       /* 1 = nn.BCECriterion backward then nn.Sigmoid backward, eps 1e-12;
                                0 = fused (p - t)/B (identical away from fp32 saturation)         */
  int32_t entity_update;     /* 0 = lazy-exact: rows of entity_emb are brought up to date when
                                    touched; bit-identical to the dense update of optim.adam
                                1 = dense: every row every step, as the reference does          */
} kprn_opt;

/* ---- lifetime -------------------------------------------------------------------- */
/* OneModel.lua:204-309: build predictor_net + reducer, init uniform(-paramInit,paramInit) */
int kprn_create(const kprn_config* cfg, kprn_handle** out);
void kprn_destroy(kprn_handle* h);
const char* kprn_last_error(const kprn_handle* h);
/* library / kernel build id, e.g. "kprn-amd 0.1 gfx950" */
const char* kprn_version(void);

/* ---- parameters (nn.Module:parameters()/getParameters(), MyOptimizer.lua:42) -------- */
int kprn_num_params(kprn_handle* h, int64_t* n);
/* n = element count of dst/src, must match the tensor */
int kprn_get_param(kprn_handle* h, const char* name, float* dst, int64_t n);
int kprn_set_param(kprn_handle* h, const char* name, const float* src, int64_t n);
int kprn_get_grad(kprn_handle* h, const char* name, float* dst, int64_t n);
/* n_rows rows of one tensor by 0-based row index (Lua: lookup.weight[id], id = row + 1; net/FeatureEmbedding.lua:29,41-49,86):
   what one reads / writes of a 20 M-row nn.LookupTable without moving the table.  dst / src: [n_rows][cols] */
int kprn_get_param_rows(kprn_handle* h, const char* name, const int64_t* rows, int64_t n_rows, float* dst);
int kprn_set_param_rows(kprn_handle* h, const char* name, const int64_t* rows, int64_t n_rows, const float* src);
/* whole flat vector in getParameters() order */
int kprn_get_flat_params(kprn_handle* h, float* dst, int64_t n);
int kprn_set_flat_params(kprn_handle* h, const float* src, int64_t n);
int kprn_get_flat_grads(kprn_handle* h, float* dst, int64_t n);
/* optimiser state (optState: adam m,v / adagrad paramVariance) -- slot 0 or 1 */
int kprn_get_flat_opt_state(kprn_handle* h, int32_t slot, float* dst, int64_t n);
/* MyOptimizer:zeroPadTokens (MyOptimizer.lua:74-93): zero row V (1-based) of each table */
int kprn_zero_pad_tokens(kprn_handle* h);

/* ---- batches resident in HBM (BatcherFileList:populateGPUTensor, BatcherFileList.lua:78-96) */
/* validates every id against its vocabulary; labels may be NULL for scoring            */
int kprn_batch_create(kprn_handle* h, const int32_t* idx, const float* labels,
                      int32_t B, int32_t P, int32_t T, int32_t F, kprn_batch** out);
void kprn_batch_destroy(kprn_handle* h, kprn_batch* b);
/* Streaming feed = BatcherFileList's GPU double buffer (BatcherFileList.lua:53-96: tensors preallocated once, every minibatch
 * :copy()'d into them): (re)fills *slot (NULL: a new slot is allocated; a refill that fits the slot's buffers allocates nothing)
 * with the upload, the id validation, the occurrence index and the identical-prefix plan queued on a dedicated FEED stream, and
 * returns at once.  The feed starts behind everything queued on the handle so far (the last readers of the slot's previous
 * contents among it) and runs under whatever is queued next -- call it for batch i+1 right before the step on batch i.  The first
 * call that uses the slot waits for its feed; an out-of-range id surfaces there as KPRN_E_INDEX.  idx / labels must stay
 * unchanged until then; only page-locked host memory (kprn_host_alloc) is copied without holding the calling thread.      */
int kprn_batch_feed_async(kprn_handle* h, kprn_batch** slot, const int32_t* idx, const float* labels,
                          int32_t B, int32_t P, int32_t T, int32_t F);
/* The same for a SHUFFLED epoch (Batcher:shuffle, Batcher.lua:35-41, permutes the file's tensors; OneModel.lua:326 turns it on):
 * pair i of the minibatch is row rows[i] (0-based, < n_rows) of the file's arrays data [n_rows,P,T,F] / labels [n_rows], gathered
 * by the feed's worker threads straight into the upload image -- the caller permutes nothing and copies nothing.  data and rows
 * must stay unchanged until the slot's first use; labels are read before the call returns.                              */
int kprn_batch_feed_rows_async(kprn_handle* h, kprn_batch** slot, const int32_t* data, const float* labels, int64_t n_rows,
                               const int64_t* rows, int32_t B, int32_t P, int32_t T, int32_t F);
/* BatcherFileList.lua:53-60: size a slot once for the largest minibatch it will hold (max_pairs pairs, max_paths paths of T steps), so that
 * no later feed allocates (an allocation waits for the device).  *slot NULL: a new, empty slot.  Capacities only ever grow.           */
int kprn_batch_slot_reserve(kprn_handle* h, kprn_batch** slot, int32_t max_pairs, int64_t max_paths, int32_t T, int32_t F, int32_t with_labels);
/* Where the feed derives a batch's plan and index: kprn_set_option(h, "feed_build", "host") (default) -- worker threads on the host
 * cores write them into page-locked staging and the GPU sees DMA traffic only ("feed_workers" batches at once, "feed_threads" helper
 * threads each) -- or "device": the kernels of kprn_batch_create on a side stream.  Same arrays either way.
 * kprn_host_batch_index is the host derivation on its own (no handle, no GPU): ids [B,P,T,F] -> validation, identical-prefix plan
 * (plan != 0: idx_s [B*P*T*F], perm / slot_of [B*P], tile_k [ceil(B*P/64)], pmeta [24]) and the entity-occurrence index
 * (key_sorted / pos_sorted / uniq: B*P*T + 8 entries each); summary[4] = {an id out of range, longest shared prefix, distinct
 * entity rows, (path, step) positions the kernels execute}.                                                             */
int kprn_host_batch_index(const int32_t* idx, int32_t B, int32_t P, int32_t T, int32_t F, int32_t num_types, int32_t Vt, int32_t Ve, int32_t Vr,
                          int32_t plan, int32_t threads, int32_t* idx_s, int32_t* perm, int32_t* slot_of, int32_t* tile_k, int32_t* pmeta,
                          int32_t* key_sorted, int32_t* pos_sorted, int32_t* uniq, int64_t* summary);
/* page-locked host buffers for the feed (the reference preallocates its staging tensors likewise, BatcherFileList.lua:53-60) */
int kprn_host_alloc(kprn_handle* h, size_t bytes, void** out);
int kprn_host_free(kprn_handle* h, void* p);
/* number of distinct entity rows the batch references (= the rows one training step on it touches) */
int kprn_batch_distinct_rows(kprn_handle* h, const kprn_batch* b, int32_t* n);
/* (path, step) positions a pass over the batch executes: B*P*T, less the leading steps that whole 64-path tiles share with
 * the batch's reference step (left padding, movie_data_format.py:250-254) and that the fused kernels therefore run once
 * for the batch instead of once per path -- same results; for work / roofline accounting                                  */
int kprn_batch_executed_steps(kprn_handle* h, const kprn_batch* b, int64_t* steps);
/* what the fused BPTT launches' time-split tile hand-over (option "tile_handover") does with this batch: out[0] = pairs of workgroups
 * between which a tile changes hands, out[1] = steps moved in all, out[2] / out[3] = the longest workgroup's work in HALF steps with whole
 * tiles only / with the hand-over (a tile's first executed step counts one half: it has no recurrent product).  All zero / equal when the
 * batch does not run on the fused kernels or the option is off.  Diagnostics: waits for the engine's stream.                          */
int kprn_batch_handover_stats(kprn_handle* h, const kprn_batch* b, int64_t* out /* [4] */);

/* ---- scoring: model:forward(inputs) (test_from_checkpoint.lua:81-82,109) ------------
 * probs[B]      = Sigmoid(reduce_p(mapper))[:, classId]       (Select(2,classId))
 * all_probs     = optional [B,C] (before Select)
 * pooled        = optional [B,C] reducer output before Sigmoid
 * path_scores   = optional [B*P,C] mapper output (nn.Linear(H,46), OneModel.lua:275)      */
int kprn_forward(kprn_handle* h, const int32_t* idx, int32_t B, int32_t P, int32_t T, int32_t F,
                 int32_t class_id, float* probs, float* all_probs);
int kprn_forward_batch(kprn_handle* h, const kprn_batch* b, int32_t class_id,
                       float* probs, float* all_probs, float* pooled, float* path_scores);
/* async variant for throughput loops: results stay on the device until kprn_read_probs  */
int kprn_forward_batch_async(kprn_handle* h, const kprn_batch* b, int32_t class_id);
/* kprn_set_option(h, "score_split", "f"): kprn_forward_batch_async queues the first (1 - f) of the batch's 64-path tiles only; this call queues
 * the rest + the pooling stage behind everything queued so far (a data-parallel step: between kprn_dp_exchange_begin and _finish, so that the
 * collective has compute to hide under while most of the pass shared the chip with the training forward).  Whatever waits for the pass
 * (the optimiser step, kprn_read_probs) places a forgotten second part itself. */
int kprn_forward_batch_async_rest(kprn_handle* h);
int kprn_read_probs(kprn_handle* h, float* probs, int32_t B);
/* embedding sub-net output x[N,T,D] (FeatureEmbedding.lua:112-121), for bit-exact checks */
int kprn_embed(kprn_handle* h, const int32_t* idx, int64_t N, int32_t T, int32_t F, float* x);

/* ---- training ---------------------------------------------------------------------- */
/* fEval of MyOptimizer.lua:184-195: zeroGradParameters; forward; BCE; backward.
 * inv_batch = 0 -> 1/B; data-parallel callers pass 1/B_global.  loss may be NULL (async). */
int kprn_backward_batch(kprn_handle* h, const kprn_batch* b, int32_t class_id, int32_t bce_literal,
                        float inv_batch, float* loss);
/* MyOptimizer.lua:196-219 on the gradients now in the handle: clip/L2 iff regularize==1,
 * optim step, zeroPadTokens.  Async.                                                     */
int kprn_apply_update(kprn_handle* h, const kprn_opt* opt);
/* MyOptimizer:trainBatch = zeroPadTokens + kprn_backward_batch + kprn_apply_update.
 * With loss != NULL the call returns as soon as the loss is on the host, i.e. after the forward and the loss stage of this step: the backward and
 * the optimiser step are queued and complete in stream order BEFORE anything a later call on this handle can observe (parameters, gradients,
 * scores, the next step) -- the caller's idx / labels have been consumed, and the caller prepares its next minibatch while the device finishes
 * this one (MyOptimizer.lua:184-221 returns the same number; it just cannot overlap).  A device error raised by the rest of the step surfaces at
 * the next call.  kprn_set_option(h, "train_step_return", "drain") restores the wait for the whole step; kprn_sync always waits for everything.
 * kprn_train_step (host buffers) waits for the loss stage whether or not `loss` is NULL -- that wait is what lets the next call upload its
 * minibatch beside this step's backward; kprn_train_step_batch with loss == NULL is fully asynchronous.                                      */
int kprn_train_step(kprn_handle* h, const int32_t* idx, int32_t B, int32_t P, int32_t T, int32_t F,
                    const float* labels, int32_t class_id, const kprn_opt* opt, float* loss);
int kprn_train_step_batch(kprn_handle* h, const kprn_batch* b, int32_t class_id, const kprn_opt* opt,
                          float* loss /* NULL = async */);
/* loss of the most recent backward, once the stream has drained */
int kprn_read_loss(kprn_handle* h, float* loss);
/* kprn_set_option(h, "loss_accumulate", "1"): every backward adds its loss to a running sum on the device; this reads the sum and
 * the number of steps in it (MyOptimizer.lua:148-156 prints totalError / steps every gradientStepCounter steps and per epoch), so
 * the training loop needs no host sync per step.  reset != 0 starts a new sum.                                          */
int kprn_read_loss_sum(kprn_handle* h, float* sum, int32_t* steps, int32_t reset);
int kprn_sync(kprn_handle* h);

/* The scoring writer's lines (eval/test_from_checkpoint.lua:110-118: counter \t string.format("%.5f", score) \t label, one per
 * pair, the label as Lua 5.1 prints a number = "%.14g"), n of them from counter0 on, formatted by the host cores into out.
 * Host-only (no handle, no GPU).  *written = bytes written; KPRN_E_ARG with *written = -(bytes needed) when cap is too small. */
int kprn_format_score_lines(int64_t counter0, const float* probs, const float* labels, int64_t n, char* out, int64_t cap,
                            int64_t* written);

/* ---- data-parallel hooks (new design; the reference is single-device) -----
 * The dense gradients (type_emb, relation_emb, LSTM, head) live in ONE contiguous device
 * buffer that the caller all-reduces (RCCL).  entity_emb gradients are row-sparse: pack
 * -> ONE all-gather -> merge on every rank.  All pointers are DEVICE pointers.           */
int kprn_dense_grad_buffer(kprn_handle* h, void** dev_ptr, int64_t* n_floats);
/* upper bound of the rows one step of this rank touches (agree on the MAX over ranks as the packing capacity) */
int kprn_sparse_grad_capacity(kprn_handle* h, int32_t* max_rows_per_step);
/* moves this step's touched entity rows into ONE packed device buffer of 32-bit words
 *   { int32 count, 3 x pad, int32 ids[capacity] (sorted, 0-based), float rows[capacity][d_entity] }
 * and clears them from the local accumulator; *n_words = 4 + capacity (1 + d_entity): the unit of the all-gather. */
int kprn_sparse_grad_pack(kprn_handle* h, int32_t capacity, void** dev_buf, int64_t* n_words);
/* dev_all = the `world` packed buffers back to back (all-gather output, THIS rank's included).  The union of the rows with
 * their sums in rank order (identical bits on every rank) is what the next kprn_apply_update walks: built here in the
 * accumulator (a marking pass per rank + one compaction), or -- kprn_set_option(h, "dp_fused_update", "1") -- only recorded
 * and formed inside the optimiser's row kernel (lazy-exact Adam without clip / L2; anything else builds it first): the
 * caller then keeps dev_all alive and unchanged until kprn_apply_update has been queued.  With "dp_dense_in_pack" = 1 the
 * dense gradient buffer rides behind the rows (n_words grows by its length rounded up to 4) and is summed here too.       */
int kprn_sparse_grad_merge(kprn_handle* h, const void* dev_all, int32_t world, int32_t capacity);
/* the stream everything is queued on (hipStream_t), so the caller can order collectives  */
int kprn_stream(kprn_handle* h, void** stream);

/* ---- the exchange issued by the engine (new design, SURVEY 8e: RCCL over xGMI) --------
 * Instead of handing buffers to the caller's collectives, the engine holds an RCCL communicator and queues
 *   pack (straight into its slot of the gathered buffer) -> ncclAllGather IN PLACE -> dense sum -> optimiser step on the union
 * on its own stream from two C calls; nothing of the host program sits between the kernels.  librccl is dlopen'ed
 * (rccl_path, or NULL = "librccl.so" as the process already has it); the caller's control plane carries the bootstrap:
 *   rank 0: kprn_dp_unique_id(path, id) -> broadcast the 128 bytes -> every rank: kprn_dp_init(h, path, id, rank, world)
 * (collective).  Per step, after kprn_backward_batch with inv_batch = 1 / (pairs of the GLOBAL minibatch):
 *   kprn_dp_exchange_begin(h, capacity)   capacity = the same multiple of 4 on every rank, >= every rank's touched rows
 *   ... work that does not depend on the update (a scoring pass) may be queued here ...
 *   kprn_dp_exchange_finish(h, opt)       = kprn_sparse_grad_merge + kprn_apply_update on the gathered buffer
 * kprn_set_option(h, "dp_comm_stream", "1") puts the collective on a stream of its own so that the work queued in between
 * overlaps it (world > 1).  Replicas stay bit-identical (rank-ordered sums).                                             */
int kprn_dp_available(const char* rccl_path);   /* KPRN_OK when librccl loads and has the entry points (no handle, no GPU work) */
int kprn_dp_unique_id(const char* rccl_path, void* id128 /* out: 128 bytes */);
int kprn_dp_init(kprn_handle* h, const char* rccl_path, const void* id128, int32_t rank, int32_t world);
int kprn_dp_exchange_begin(kprn_handle* h, int32_t capacity);
int kprn_dp_exchange_finish(kprn_handle* h, const kprn_opt* opt);
int kprn_dp_comm_size(kprn_handle* h, int32_t* nranks);   /* the communicator's rank count as RCCL reports it (ncclCommCount) */
int kprn_dp_shutdown(kprn_handle* h);   /* destroys the communicator (kprn_destroy does it too); restores the two options kprn_dp_init forced on */

/* ---- checkpoints (OneModel.lua:392-408 torch.save{embeddingLayer,predictor_net}) ------
 * native format: header + flat fp32 vector in getParameters() order (optimizer state is
 * NOT saved, like the reference).                                                        */
int kprn_save(kprn_handle* h, const char* path);
int kprn_load(kprn_handle* h, const char* path);

/* ---- measurement ------------------------------------------------------------------- */
/* when enabled, every kernel family is bracketed by HIP events on the handle's stream   */
int kprn_profile_enable(kprn_handle* h, int32_t on);
int kprn_profile_reset(kprn_handle* h);
/* fills up to cap entries; returns the number of kernel families seen in *n             */
typedef struct { char name[48]; double total_ms; int64_t launches; } kprn_prof_entry;
int kprn_profile_get(kprn_handle* h, kprn_prof_entry* out, int32_t cap, int32_t* n);
/* options (all strings):
 *   "impl"            "auto" (fused kernels where the shape allows) | "generic"
 *   "prefix_plan"     "1" (default): batches created from now on get an identical-prefix plan (leading steps shared by whole
 *                     64-path tiles are run once per batch, fused path); "0": every step of every path is executed
 *   "score_overlap"   "1": kprn_forward_batch_async runs the (fused) scoring pass on a second stream with its own output
 *                     buffers, so that it shares the chip with the work enqueued after it -- typically the training forward of the
 *                     same step, which does not depend on it.  Whatever would change what the pass reads (an optimiser step, a row
 *                     catch-up, kprn_set_*) waits for it; kprn_read_probs / kprn_sync wait for it on the host.  "0" (default): in
 *                     order on the handle's stream.
 *   "reserve_cus"     CUs the persistent scoring kernel leaves free (a collective's copy kernels run beside it), 0..128
 *   "profile_filter"  kernel-family name prefix: only those families get HIP events while profiling is on ("" = all); an event
 *                     pair costs ~4 us of stream time
 *   round 5 (each is the A/B switch of one design choice; the defaults are the fast paths, the tests run both sides in one process):
 *   "bf16_small_tables" "1" (default): bf16 pipeline (compute_dtype 1, persistent launches): the gradients of the type / relation tables (<= 128
 *                     rows together) and of their column blocks of W_i2g come from 128 one-hot columns of ONE merged dW product, dx is formed
 *                     for the entity slice only; "0": full dx product + table-gradient launch
 *   "bf16_bptt_dxe"   "8" (default) | "16": that slice of dx is formed inside the persistent BPTT launch (value = depth of its weight ring);
 *                     "0": by its own product launch
 *   "small_tables"    "1" (default): the same identity on layer 0 of the generic fp32 LSTM / rnn backward; "0": dx product + table-gradient launch
 *   "persist_layers"  "1" (default): a recurrent layer of the generic fp32 pipeline (FastLSTM / rnn, Din % 4 == 0, Din, H <= 256) runs as ONE
 *                     persistent launch, forward and BPTT, once the batch gives every CU a 64-path tile; "2": at any batch size; "0": one launch
 *                     per step
 *   "bf16_gemm_pingpong" "0" (default) | "1", "bf16_gemm_regstage" "0" | "1", "bf16_gemm_touch" "0" | chunks ahead, "bf16_t_pad" "64" | 0..512
 *                     (elements, a multiple of 8): measured alternatives of the bf16 split-K dW product (two wave groups one barrier apart; operands
 *                     staged through registers; L2 prefetch by touch; row pitch of its transposed operands) -- none faster than the default,
 *                     kept as the record of DESIGN.md section 7-3 and run against the default by tests/test_gpu_persist.py
 *   "score_rest_before_bptt" "0" (default) | "1": with "score_split" f > 0, the deferred part is queued right behind the loss stage on a stream of the lowest
 *                     priority: its single-tile workgroups take the CUs the first BPTT launch leaves idle in its tail
 *   "score_rest_in_backward" "0" (default) | "1": with "score_split" f > 0, the fused backward places the deferred part of the scoring pass itself, right behind its
 *                     last BPTT launch (beside the step's serial tail); measured slower than the whole pass first at world 1
 *   "tile_handover"   "2" (default) | "1" | "0": the fused D = H = 64 BPTT launches let a 64-path tile change workgroups once, between two of its steps, so
 *                     that the workgroups' step sums differ by less than a step on a left-padded path set (workgroup b paired with b + G / 2; "1": with
 *                     G - 1 - b; "0": whole tiles only).  Same gradients up to fp32 re-association of the weight-gradient partial sums; see
 *                     kprn_batch_handover_stats, DESIGN.md section 3.3b
 *   "adam_merged"     "1" (default) | "0": lazy-exact Adam updates the touched entity rows and the dense arena in one launch (bit-identical to two)
 *   "bwd_pipe"        "1" (default) | "0": fused path, batches of 16-row tiles, two layers: both layers' BPTT in one launch, the bottom layer a step behind
 *                     the top layer | one launch per layer
 *   "score_dual"      "2" (default) | "1" | "0": with "score_overlap": a pass queued by kprn_forward_batch_async is held back and runs in the launch of the
 *                     training forward that follows (one kernel, no second stream) -- for batches below 8 192 paths | always | never; anything that needs
 *                     the pass earlier runs it the usual way
 *   "catchup_prefix"  "1" (default) | "0": fused D = H = 64 path: a batch's lazy-exact row catch-up and its identical-prefix table in one launch | two
 *                     (bit-identical)
 *   "fused_small_tables" "1" (default) | "0": fused D = H = 64 path: type / relation table gradients formed inside the bottom BPTT launch | by a
 *                     passenger job of the entity-gradient launch (equal to fp32 re-association)
 *   "train_step_return" "loss" (default) | "drain": see kprn_train_step
 *   "inline_upload"   "side" (default): kprn_train_step uploads its minibatch on the upload stream, beside the previous step's backward, whenever the previous
 *                     call waited for its loss (every reader of the slot being refilled is then known to be done); "main": on the engine's stream
 *   (also: "small_tiles", "score_split", "loss_accumulate", "feed_build" / "feed_threads" / "feed_workers", "dp_comm_stream",
 *    "dp_fused_update", "dp_dense_in_pack" -- described at the calls they modify)                                                 */
int kprn_set_option(kprn_handle* h, const char* key, const char* value);
/* measurement hook: mean milliseconds per launch of one GEMM shape of the generic pipeline on random data (scripts/gpu_gemm_bench.py).
 * what: 0 C = A B^T, 1 C = A B, 2 C += A^T B (split-K), 3 FastLSTM step kernel (M paths, N = H, K = Din), 4 Recurrence step kernel */
int kprn_debug_gemm(kprn_handle* h, int32_t what, int64_t M, int32_t N, int64_t K, int32_t iters, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* KPRN_H */
