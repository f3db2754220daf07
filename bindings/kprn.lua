--[[ kprn.lua -- LuaJIT FFI binding of libkprn.so (include/kprn.h).

This is the stub a maintainer of release/songPathRnn adds so that MyOptimizer.lua and
eval/test_from_checkpoint.lua call the MI355X engine instead of the Torch7 nn graph.  LuaJIT is not
in the build image, so this file is exercised only through its Python twin (kprn_amd/_ffi.py binds the
same symbols with the same struct layouts; tests/test_abi.py checks both against the header).

  local kprn = require 'kprn'
  local net  = kprn.create{Vt=6, Ve=2851220, Vr=9, dt=16, de=32, dr=16, F=3, num_types=1, H=64, L=2, reducer=2}
  -- MyOptimizer:trainBatch(inputs, targets)          (model/optimizer/MyOptimizer.lua:177-221)
  local err = net:trainBatch(inputs, targets, classId, self.optConfig, self.optInfo)
  -- test_from_checkpoint.lua:109   local preds = model:forward(inputs)
  local preds = net:forward(inputs, 1)
]]
local ffi = require 'ffi'

ffi.cdef[[
typedef struct kprn_handle kprn_handle;
typedef struct kprn_batch kprn_batch;
typedef struct {
  int32_t Vt, Ve, Vr, dt, de, dr, F, num_types, H, L, C, rnn_type, use_relu, rnn_init, compute_dtype, reducer, K, device_id, rank, world;
  float param_init; uint64_t seed; void* stream;
} kprn_config;
typedef struct {
  int32_t method; float lr, beta1, beta2, eps, lr_decay; int32_t regularize, use_grad_clip;
  float grad_clip_norm, l2; int32_t bce_literal, entity_update;
} kprn_opt;
int kprn_create(const kprn_config*, kprn_handle**);
void kprn_destroy(kprn_handle*);
const char* kprn_last_error(const kprn_handle*);
int kprn_num_params(kprn_handle*, int64_t*);
int kprn_get_param(kprn_handle*, const char*, float*, int64_t);
int kprn_get_param_rows(kprn_handle*, const char*, const int64_t*, int64_t, float*);
int kprn_set_param_rows(kprn_handle*, const char*, const int64_t*, int64_t, const float*);
int kprn_set_param(kprn_handle*, const char*, const float*, int64_t);
int kprn_get_flat_params(kprn_handle*, float*, int64_t);
int kprn_set_flat_params(kprn_handle*, const float*, int64_t);
int kprn_zero_pad_tokens(kprn_handle*);
int kprn_forward(kprn_handle*, const int32_t*, int32_t, int32_t, int32_t, int32_t, int32_t, float*, float*);
int kprn_train_step(kprn_handle*, const int32_t*, int32_t, int32_t, int32_t, int32_t, const float*, int32_t, const kprn_opt*, float*);
int kprn_save(kprn_handle*, const char*);
int kprn_load(kprn_handle*, const char*);
/* batches resident in HBM and the streaming feed (BatcherFileList's GPU double buffer, BatcherFileList.lua:53-96) */
int kprn_batch_create(kprn_handle*, const int32_t*, const float*, int32_t, int32_t, int32_t, int32_t, kprn_batch**);
void kprn_batch_destroy(kprn_handle*, kprn_batch*);
int kprn_batch_slot_reserve(kprn_handle*, kprn_batch**, int32_t, int64_t, int32_t, int32_t, int32_t);
int kprn_batch_feed_async(kprn_handle*, kprn_batch**, const int32_t*, const float*, int32_t, int32_t, int32_t, int32_t);
int kprn_batch_feed_rows_async(kprn_handle*, kprn_batch**, const int32_t*, const float*, int64_t, const int64_t*, int32_t, int32_t, int32_t, int32_t);
int kprn_host_alloc(kprn_handle*, size_t, void**);
int kprn_host_free(kprn_handle*, void*);
int kprn_forward_batch(kprn_handle*, const kprn_batch*, int32_t, float*, float*, float*, float*);
int kprn_forward_batch_async(kprn_handle*, const kprn_batch*, int32_t);
int kprn_read_probs(kprn_handle*, float*, int32_t);
int kprn_train_step_batch(kprn_handle*, const kprn_batch*, int32_t, const kprn_opt*, float*);
int kprn_read_loss(kprn_handle*, float*);
int kprn_read_loss_sum(kprn_handle*, float*, int32_t*, int32_t);
int kprn_set_option(kprn_handle*, const char*, const char*);
int kprn_sync(kprn_handle*);
int kprn_backward_batch(kprn_handle*, const kprn_batch*, int32_t, int32_t, float, float*);
int kprn_sparse_grad_capacity(kprn_handle*, int32_t*);
int kprn_dp_unique_id(const char*, void*);
int kprn_dp_init(kprn_handle*, const char*, const void*, int32_t, int32_t);
int kprn_dp_exchange_begin(kprn_handle*, int32_t);
int kprn_dp_exchange_finish(kprn_handle*, const kprn_opt*);
int kprn_forward_batch_async_rest(kprn_handle*);
int kprn_dp_comm_size(kprn_handle*, int32_t*);
int kprn_dp_shutdown(kprn_handle*);
]]

local C = ffi.load('kprn')
local M = {}
local Net = {}
Net.__index = Net

local function check(h, rc)
  if rc ~= 0 then error(('kprn error %d: %s'):format(rc, ffi.string(C.kprn_last_error(h)))) end
end

-- kprn_config.stream: nil = the engine creates its own stream; a hipStream_t; or M.STREAM_LEGACY_DEFAULT for the null stream
-- (cutorch's default stream: its handle is 0, which would otherwise read as "create your own" -- include/kprn.h)
M.STREAM_LEGACY_DEFAULT = ffi.cast('void*', ffi.cast('intptr_t', -1))

function M.create(o)
  local cfg = ffi.new('kprn_config')
  cfg.stream = o.stream
  cfg.Vt, cfg.Ve, cfg.Vr = o.Vt, o.Ve, o.Vr
  cfg.dt, cfg.de, cfg.dr = o.dt, o.de, o.dr
  cfg.F, cfg.num_types = o.F or 3, o.num_types or 1
  cfg.H, cfg.L, cfg.C = o.H, o.L or 1, o.C or 46
  cfg.rnn_type, cfg.reducer, cfg.K = o.rnn_type or 0, o.reducer or 2, o.K or 5
  cfg.use_relu, cfg.rnn_init, cfg.compute_dtype = o.use_relu or 1, o.rnn_init or 0, o.compute_dtype or 0
  cfg.device_id, cfg.rank, cfg.world = o.device_id or 0, 0, 1
  cfg.param_init, cfg.seed = o.paramInit or 0.1, o.seed or 12345
  local ph = ffi.new('kprn_handle*[1]')
  local rc = C.kprn_create(cfg, ph)
  if rc ~= 0 then error(('kprn_create failed (%d): %s'):format(rc, ffi.string(C.kprn_last_error(nil)))) end
  return setmetatable({h = ffi.gc(ph[0], C.kprn_destroy), C = cfg.C}, Net)
end

-- DoubleTensor[B,P,T,F] of 1-based ids (model/batcher/Batcher.lua:51) -> int32 host buffer, once per batch
local function to_int32(t)
  local n = t:nElement()
  local src = t:contiguous():data()
  local dst = ffi.new('int32_t[?]', n)
  for i = 0, n - 1 do dst[i] = src[i] end
  return dst
end

function Net:forward(inputs, classId)  -- == nn.Sequential():add(training_net):add(nn.Select(2,classId)):forward(inputs)
  local B, P, T, F = inputs:size(1), inputs:size(2), inputs:size(3), inputs:size(4)
  local probs = ffi.new('float[?]', B)
  check(self.h, C.kprn_forward(self.h, to_int32(inputs), B, P, T, F, classId or 1, probs, nil))
  local out = torch.DoubleTensor(B)
  for i = 1, B do out[i] = probs[i - 1] end
  return out
end

-- optConfig / optInfo: the tables model/OneModel.lua:340-383 builds and MyOptimizer keeps (MyOptimizer.lua:19-27).  Only fields
-- that exist there are read: optInfo.optimMethod (optim.adam | optim.adagrad -- there is no useAdam field), optInfo.regularize,
-- optInfo.useGradClip (a boolean, OneModel.lua:114), optInfo.gradClipNorm, optInfo.l2; optConfig.learningRate, .beta1, .beta2,
-- .epsilon (adam) or .learningRateDecay (adagrad).  tests/test_host.py checks this list against the reference.
function Net:trainBatch(inputs, targets, classId, optConfig, optInfo)
  local B, P, T, F = inputs:size(1), inputs:size(2), inputs:size(3), inputs:size(4)
  local opt = ffi.new('kprn_opt')
  local optim = rawget(_G, 'optim') or require 'optim'   -- OneModel.lua:16 has it loaded
  assert(optInfo.optimMethod == optim.adam or optInfo.optimMethod == optim.adagrad, 'optimMethod must be optim.adam or optim.adagrad')
  opt.method = (optInfo.optimMethod == optim.adam) and 1 or 0
  opt.lr, opt.beta1, opt.beta2, opt.eps = optConfig.learningRate, optConfig.beta1 or 0.9, optConfig.beta2 or 0.999, optConfig.epsilon or 1e-8
  opt.lr_decay = optConfig.learningRateDecay or 0
  opt.regularize, opt.use_grad_clip = optInfo.regularize, optInfo.useGradClip and 1 or 0
  opt.grad_clip_norm, opt.l2 = optInfo.gradClipNorm, optInfo.l2
  local lab = ffi.new('float[?]', B)
  local td = targets:contiguous():data()
  for i = 0, B - 1 do lab[i] = td[i] end
  local loss = ffi.new('float[1]')
  -- returns once the LOSS is on the host (the int32 copy of `inputs` and `lab` have been consumed by then); the backward and the optimiser step finish in
  -- stream order while Lua fetches the next minibatch -- Net:sync() waits for everything, e.g. before os.clock() or torch.save
  check(self.h, C.kprn_train_step(self.h, to_int32(inputs), B, P, T, F, lab, classId or 1, opt, loss))
  return loss[0]
end

-- ---- the fast path: what Batcher / BatcherFileList / MyOptimizer:train call when the data stays in the engine's hands ----------
-- A file's tensors converted ONCE (Batcher.lua:12-16 loads labels / data): int32 ids [n,P,T,F] and float labels [n] in host memory.
function M.file_arrays(labels, data)
  local n = data:size(1)
  local lab = ffi.new('float[?]', n)
  local ld = labels:contiguous():data()
  for i = 0, n - 1 do lab[i] = ld[i] end
  return {n = n, P = data:size(2), T = data:size(3), F = data:size(4), idx = to_int32(data), labels = lab}
end

local function fill_opt(optConfig, optInfo)
  local opt = ffi.new('kprn_opt')
  local optim = rawget(_G, 'optim') or require 'optim'
  opt.method = (optInfo.optimMethod == optim.adam) and 1 or 0
  opt.lr, opt.beta1, opt.beta2, opt.eps = optConfig.learningRate, optConfig.beta1 or 0.9, optConfig.beta2 or 0.999, optConfig.epsilon or 1e-8
  opt.lr_decay = optConfig.learningRateDecay or 0
  opt.regularize, opt.use_grad_clip = optInfo.regularize, optInfo.useGradClip and 1 or 0
  opt.grad_clip_norm, opt.l2 = optInfo.gradClipNorm, optInfo.l2
  return opt
end

-- BatcherFileList:populateGPUTensor for a shuffled epoch: minibatch = rows `rows` (0-based int64 cdata, count B) of `file`
-- (M.file_arrays); slot: ffi.new('kprn_batch*[1]') kept by the caller, refilled in place.  Returns at once; the gather, the
-- index build and the upload run under the step queued next.
function Net:feedRows(slot, file, rows, B)
  check(self.h, C.kprn_batch_feed_rows_async(self.h, slot, file.idx, file.labels, file.n, rows, B, file.P, file.T, file.F))
end

-- MyOptimizer:trainBatch on a fed slot; no host synchronisation (the epoch's error comes from Net:lossSum)
function Net:trainBatchSlot(slot, classId, optConfig, optInfo)
  check(self.h, C.kprn_train_step_batch(self.h, slot[0], classId or 1, fill_opt(optConfig, optInfo), nil))
end

function Net:accumulateLoss(on) check(self.h, C.kprn_set_option(self.h, 'loss_accumulate', on and '1' or '0')) end
function Net:lossSum(reset)  -- -> totalError, steps since the last reset (MyOptimizer.lua:148-156)
  local s, n = ffi.new('float[1]'), ffi.new('int32_t[1]')
  check(self.h, C.kprn_read_loss_sum(self.h, s, n, reset and 1 or 0))
  return s[0], n[0]
end

-- ---- one process per GPU (the reference is single-device; include/kprn.h "the exchange issued by the engine") ---------------
-- The Lua program only carries 128 bytes: rank 0 draws the id, whatever control plane the job has (a file, a socket, MPI) hands
-- it to every rank, and every rank joins.  After that a data-parallel trainBatch is three calls; the engine queues pack ->
-- in-place RCCL all-gather -> optimiser step on the union of all ranks' rows on its own stream.
function M.dpUniqueId(rcclPath)
  local id = ffi.new('char[128]')
  local rc = C.kprn_dp_unique_id(rcclPath, id)
  if rc ~= 0 then error(('kprn error %d: %s'):format(rc, ffi.string(C.kprn_last_error(nil)))) end
  return ffi.string(id, 128)
end
function Net:dpInit(id, rank, world, rcclPath) check(self.h, C.kprn_dp_init(self.h, rcclPath, id, rank, world)) end
-- slot: THIS rank's pairs of the global minibatch; globalPairs: pairs of all ranks (the BCE mean is over the global minibatch);
-- capacity: the same multiple of 4 on every rank, >= the entity rows any rank touches in a step (minibatch * P * T is a bound)
function Net:trainBatchSlotDP(slot, classId, optConfig, optInfo, globalPairs, capacity)
  check(self.h, C.kprn_zero_pad_tokens(self.h))                                                     -- MyOptimizer.lua:181
  check(self.h, C.kprn_backward_batch(self.h, slot[0], classId or 1, 0, 1.0 / globalPairs, nil))   -- MyOptimizer.lua:186-195
  check(self.h, C.kprn_dp_exchange_begin(self.h, capacity))
  check(self.h, C.kprn_dp_exchange_finish(self.h, fill_opt(optConfig, optInfo)))                    -- MyOptimizer.lua:196-219
end

function Net:zeroPadTokens() check(self.h, C.kprn_zero_pad_tokens(self.h)) end
function Net:sync() check(self.h, C.kprn_sync(self.h)) end   -- everything queued on the engine has finished (timing, hand-over of the GPU)
function Net:save(path) check(self.h, C.kprn_save(self.h, path)) end
function Net:load(path) check(self.h, C.kprn_load(self.h, path)) end

return M
