"""Data-parallel training step over torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

New design -- the reference is single-device.  One process per GPU; each
rank owns a full parameter replica and a disjoint slice of the (user,item) pairs (pairs are
independent units: model/module/MapReduce.lua:24-47 never mixes them).  Per step:

  1. every rank runs zeroGrad + forward + BCE + backward on ITS pairs, with the loss scaled by
     1/B_global (the mean over the global minibatch, nn.BCECriterion sizeAverage);
  2. dense gradients (type/relation tables, LSTM, head: ONE contiguous device buffer, ~0.3 MB
     for D=H=64,L=2) ride behind the rows of 3. (adapter.dense_in_pack, the default) and are summed
     in rank order by the merge; or (dense_in_pack=False) one all-reduce(sum);
  3. entity-table gradients are row-sparse -> each rank packs {count, row ids (ascending), grad rows} of the
     rows it touched into ONE fixed-capacity buffer, ONE all-gather, then every rank forms the union of all
     ranks' rows with sums in rank order (identical addition order everywhere => replicas stay
     bit-identical): inside the optimiser's row kernel (adapter.fused_update, the default: the entry of the
     lowest rank that touched a row owns it and looks the other ranks' lists up by binary search) or as a
     separate merge pass;
  4. the optimiser step runs locally on the summed gradient (MyOptimizer.lua:196-219).

The collective calls only see an "adapter" that exposes the engine's buffers as torch tensors,
so the same code is exercised on CPU (gloo, world_size 2) with a numpy-backed adapter.
"""
import contextlib
import os

import numpy as np
import torch
import torch.distributed as dist


def call_with_watchdog(fn, timeout_s, what):
    """fn() on a daemon thread, at most timeout_s seconds -> (ok, reason).  A collective bootstrap that never returns (ncclCommInitRank with a rank
    missing, a wedged transport) must not hang the job: the caller falls back and records `reason`; the stuck thread is left behind (it cannot be
    cancelled) and dies with the process.  An exception raised by fn is a failure with its message as the reason."""
    import threading
    box = {}

    def run():
        try:
            fn()
            box["ok"] = True
        except BaseException as ex:   # noqa: BLE001
            box["err"] = f"{type(ex).__name__}: {ex}"

    th = threading.Thread(target=run, name=f"kprn-watchdog-{what}", daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return False, f"{what} did not return within {timeout_s:g} s"
    if "err" in box:
        return False, f"{what} failed: {box['err']}"
    return True, None


def first_reason(mine, group=None):
    """the first rank's non-empty reason (every rank learns why the job fell back, not only the rank it happened on).  Collective."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return mine
    allr = [None] * dist.get_world_size(group)
    dist.all_gather_object(allr, mine, group=group)
    for r, why in enumerate(allr):
        if why:
            return why if why == mine else f"rank {r}: {why}"
    return None


class DpHang(RuntimeError):
    """the engine's own exchange did not complete on the device within the watchdog's bound"""


class _DevArray:
    """minimal __cuda_array_interface__ carrier for a raw device pointer owned by libkprn."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 3, "strides": None}


def _round4(n):
    return (int(n) + 3) // 4 * 4


def wrap_device(ptr, n, kind, device):
    typestr = {"f32": "<f4", "i32": "<i4"}[kind]
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


class GpuAdapter:
    """Exposes one kprn Engine's gradient buffers as CUDA tensors (zero-copy)."""

    def __init__(self, engine, device, dense_in_pack=True, fused_update=True):
        self.e = engine
        self.device = torch.device(device)
        sh = engine.stream()   # (first: the engine hands out its exchange buffers only to a caller that knows which stream it queues on)
        ptr, n = engine.dense_grad_buffer()
        self.dense = wrap_device(ptr, n, "f32", self.device)
        self.de = engine.cfg.de
        self._cap = 0
        # Every torch operation of the exchange (the collectives' stream hand-over, copies, timing events) must be ordered with the
        # engine's kernels: they are issued with the ENGINE's stream as torch's current stream.  (Handing the engine torch's current
        # stream is not enough: the default stream's handle is 0, which kprn_create reads as "make your own" -- the collectives then
        # raced the pack / merge kernels: right at step 0 by luck, entity rows off by an optimiser step from step 1 on.)
        # (an engine created with KPRN_STREAM_LEGACY_DEFAULT queues on the null stream = torch's default stream)
        self.stream = torch.cuda.ExternalStream(sh, device=self.device) if sh else torch.cuda.default_stream(self.device)
        # ONE collective per step: the dense gradient arena (0.27 MB at D = H = 64) rides behind the packed entity rows in the all-gather and
        # every rank sums the W copies in rank order inside the merge -- no all-reduce, two stream hand-overs less, and dense gradients that
        # are bit-identical on every replica by construction
        self.dense_in_pack = bool(dense_in_pack)
        engine.set_option("dp_dense_in_pack", "1" if self.dense_in_pack else "0")
        # union + optimiser step in one launch: the merge only records the gathered buffer, the lazy-exact Adam row update walks it in place
        # (kprn_api.hip dp_fused_update; DataParallel keeps that buffer alive and untouched until the update has been queued)
        self.fused_update = bool(fused_update)
        engine.set_option("dp_fused_update", "1" if self.fused_update else "0")

    # ---- the exchange issued by the engine itself (kprn_dp_*, include/kprn.h): RCCL on the engine's stream ----
    def native_setup(self, group, rank, world, timeout_s=None):
        """Bootstraps the engine's own RCCL communicator over the torch process group (which only carries the 128-byte id) -> True, or
        False when every rank agrees it cannot be had (then the collectives stay with torch.distributed; self.fallback_reason says why).
        Collective.  ncclCommInitRank runs under a watchdog (KPRN_DP_INIT_TIMEOUT seconds, default 60): a bootstrap that never returns on
        some rank fails everywhere instead of hanging the job."""
        from . import _ffi
        self.fallback_reason = None
        if timeout_s is None:
            timeout_s = float(os.environ.get("KPRN_DP_INIT_TIMEOUT", "60"))
        if dist.get_backend(group) != "nccl":
            self.fallback_reason = "process group backend is not nccl"
            return False
        if not (self.dense_in_pack and self.fused_update):
            self.fallback_reason = "caller asked for the dense all-reduce / the separate merge"
            return False   # kprn_dp_init forces both options on: a caller who asked for the all-reduce / the separate merge keeps the hook path
        path = _ffi.torch_rccl_path()
        idb, ok = bytes(128), 1
        try:
            if rank == 0:
                idb = _ffi.dp_unique_id(path)
            elif not _ffi.dp_available(path):
                ok = 0
        except Exception as ex:   # noqa: BLE001
            ok = 0
            self.fallback_reason = f"librccl not usable from the engine: {ex}"
        with torch.cuda.stream(self.stream):
            t = torch.tensor([ok], dtype=torch.int32, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            if int(t.item()) == 0:
                self.fallback_reason = self.fallback_reason or "librccl not usable from the engine on some rank"
                return False
            idt = torch.tensor(list(idb), dtype=torch.uint8, device=self.device)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast(idt, src=src, group=group)
            idb = bytes(idt.cpu().tolist())

            def init():
                if os.environ.get("KPRN_DP_TEST_INIT_HANG") == "1":   # (tests: a bootstrap that never returns)
                    import time
                    time.sleep(3600)
                self.e.dp_init(idb, rank, world, path)   # ncclCommInitRank: every rank is in here
            good, why = call_with_watchdog(init, timeout_s, "kprn_dp_init (ncclCommInitRank)")
            if not good:
                ok = 0
                self.fallback_reason = why
            t = torch.tensor([ok], dtype=torch.int32, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            if int(t.item()) == 0:
                self.fallback_reason = first_reason(self.fallback_reason, group) or "kprn_dp_init failed on another rank"
                if ok:
                    self.e.dp_shutdown()
                self.e.set_option("dp_dense_in_pack", "1" if self.dense_in_pack else "0")
                self.e.set_option("dp_fused_update", "1" if self.fused_update else "0")
                return False
        self.dense_in_pack = True
        self.fused_update = True
        return True

    def exchange_begin(self, capacity):
        self.e.dp_exchange_begin(capacity)

    def exchange_finish(self, opt):
        self.e.dp_exchange_finish(opt)

    def backward(self, batch, class_id, bce_literal, inv_batch):
        self.e.backward(batch, class_id, bce_literal, inv_batch, want_loss=False)

    def dense_grads(self):
        return self.dense

    def local_rows(self):
        return self.e.sparse_grad_capacity()

    def batch_rows(self, batch):
        """distinct entity rows a step on `batch` touches (known from the batch index before the backward runs)"""
        return batch.n_uniq

    def pack(self, capacity):
        ptr, n_words = self.e.sparse_grad_pack(capacity)
        return wrap_device(ptr, n_words, "i32", self.device)

    def merge(self, all_buf, world, capacity):
        self.e.sparse_grad_merge(all_buf.data_ptr(), world, capacity)

    def apply_update(self, opt):
        self.e.apply_update(opt)

    def zero_pad(self):
        self.e.zero_pad_tokens()

    def new(self, n, dtype):
        return torch.empty(n, dtype=dtype, device=self.device)


class DataParallel:
    """adapter: GpuAdapter (or the numpy stand-in of the CPU tests).

    Packing capacity (rows per rank in the all-gathered buffer, identical on every rank):
      * set_capacity(bound) with a TRUE upper bound of the distinct rows any rank touches in any step (bench.py: the
        largest distinct-row count of its fixed batches; MyOptimizer: min(largest minibatch * P * T + 8, Ve) from the
        batcher) -> no per-step agreement, nothing on the host waits for the device;
      * otherwise (no bound promised) every step agrees on it: ONE small all-reduce carrying every rank's row count and
        pair count, read on the host; the capacity grows when a step needs more.  Correct for ragged shards, one host
        round trip per step slower.
    Loss scale: 1 / (pairs of the GLOBAL minibatch).  equal_shards=True takes B * world; else the agreed sum (or the
    caller's global_pairs)."""

    def __init__(self, adapter, group=None, equal_shards=True):
        self.a = adapter
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # with a process group (even of size 1: bench.py --force-dp) every collective below is really issued, so the
        # single-GPU dev box exercises the exact call sequence of the N-GPU run
        self.collectives = dist.is_initialized()
        self.capacity = 0
        self.bounded = False
        self.equal_shards = equal_shards
        self._all = None
        self.timing = False   # bench.py: events around the parts of the exchange (GPU adapter only)
        self._ev = []
        # The engine's own exchange (one RCCL all-gather in place on the engine's stream, queued from C between pack and update) when the
        # adapter can set it up -- GPU adapter over the "nccl" backend; KPRN_DP_NATIVE=0 keeps the collectives with torch.distributed
        self.native = False
        self.fallback_reason = None   # why the engine's own exchange is not in use (None: it is, or it was never asked for)
        if self.collectives and hasattr(adapter, "native_setup") and os.environ.get("KPRN_DP_NATIVE", "1") != "0":
            self.native = bool(adapter.native_setup(group, self.rank, self.world))
            if not self.native:
                self.fallback_reason = getattr(adapter, "fallback_reason", None)
        elif self.collectives and hasattr(adapter, "native_setup"):
            self.fallback_reason = "KPRN_DP_NATIVE=0"
        # the FIRST exchange through the engine's communicator is waited for on the host, bounded (KPRN_DP_FIRST_STEP_TIMEOUT seconds, default 60): a
        # collective that never completes on the device raises DpHang instead of hanging the first synchronisation of the job
        self._first_native_checked = False

    def _ctx(self):
        """torch's current stream := the adapter's stream (GPU adapter), for the duration of the exchange's torch calls"""
        st = getattr(self.a, "stream", None)
        return torch.cuda.stream(st) if st is not None else contextlib.nullcontext()

    def _dev(self, t):
        if self.collectives and dist.get_backend(self.group) == "nccl":
            return t.to(self.a.device)
        return t

    def set_capacity(self, local_max_rows, bound=True):
        """fixed per-rank packing capacity = max over ranks of local_max_rows; bound=True promises that no step of any rank
        touches more distinct rows than that."""
        with self._ctx():
            t = self._dev(torch.tensor([int(local_max_rows)], dtype=torch.int64))
            if self.collectives:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            self.capacity = _round4(max(1, int(t.item())))   # (multiple of 4 rows: every rank's slice of the gathered buffer stays 16-byte aligned)
        self.bounded = bool(bound)
        self._all = None
        return self.capacity

    def _agree(self, rows, pairs):
        """-> (largest row count of any rank this step, pairs of the global minibatch)"""
        t = torch.zeros(self.world + 1, dtype=torch.int64)
        t[self.rank] = int(rows)
        t[self.world] = int(pairs)
        with self._ctx():
            t = self._dev(t)
            if self.collectives:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            t = t.cpu()
        return int(t[:self.world].max().item()), int(t[self.world].item())

    def _mark(self):
        if self.timing:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._ev[-1].append(e)

    def timing_summary(self):
        """mean ms per step of the parts: dense all-reduce | pack | all-gather (+ whatever the overlap callable queued) | merge | update"""
        torch.cuda.synchronize()
        names = ["dense_allreduce", "pack", "allgather_and_overlapped_work", "merge", "optimizer_step"]
        acc = {n: 0.0 for n in names}
        k = 0
        for ev in self._ev:
            if len(ev) != len(names) + 1:
                continue
            for i, n in enumerate(names):
                acc[n] += ev[i].elapsed_time(ev[i + 1])
            k += 1
        self._ev = []
        return {n: round(v / max(k, 1), 4) for n, v in acc.items()}

    def _bounded_wait(self, timeout_s):
        """host waits, at most timeout_s, for everything queued on the adapter's stream so far"""
        import time
        if not hasattr(self.a, "stream") or not torch.cuda.is_available():
            return
        ev = torch.cuda.Event()
        ev.record()
        t0 = time.perf_counter()
        while not ev.query():
            if time.perf_counter() - t0 > timeout_s:
                raise DpHang(f"rank {self.rank}: the first exchange through the engine's RCCL communicator did not complete within {timeout_s:g} s")
            time.sleep(0.002)

    def train_step(self, batch, opt, class_id=1, global_pairs=None, overlap=None):
        """one data-parallel MyOptimizer:trainBatch; `batch` holds THIS rank's pairs.
        overlap: optional callable that ENQUEUES work which does not depend on this step's update (e.g. a scoring pass
        with the pre-update parameters); it runs while the entity-row all-gather is in flight."""
        with self._ctx():
            self._train_step(batch, opt, class_id, global_pairs, overlap)

    def _train_step(self, batch, opt, class_id, global_pairs, overlap):
        a = self.a
        need = None
        if global_pairs is None and not self.equal_shards:
            need, global_pairs = self._agree(a.batch_rows(batch), batch.B)
        gp = global_pairs if global_pairs is not None else batch.B * self.world
        a.zero_pad()  # MyOptimizer.lua:181
        a.backward(batch, class_id, bool(opt.bce_literal), 1.0 / float(gp))
        if self.timing:
            self._ev.append([])
        self._mark()
        if self.collectives and not getattr(a, "dense_in_pack", False):
            dist.all_reduce(a.dense_grads(), op=dist.ReduceOp.SUM, group=self.group)
        self._mark()
        if not self.bounded:
            if need is None:
                need, _ = self._agree(a.local_rows(), batch.B)
            if need > self.capacity:
                self.capacity = _round4(int(need * 1.25) + 16)   # the same number on every rank
                self._all = None
        elif a.local_rows() > self.capacity:
            raise RuntimeError(f"rank {self.rank}: this step touches {a.local_rows()} entity rows, more than the capacity "
                               f"{self.capacity} promised to set_capacity(bound=True)")
        cap = self.capacity
        if self.native:
            a.exchange_begin(cap)   # pack into this rank's slot of the gathered buffer + the all-gather, in place
            self._mark()            # ("pack" = pack + the collective's launch)
            if overlap is not None:
                overlap()
            self._mark()
            self._mark()            # (no separate merge)
            a.exchange_finish(opt)  # dense sum + optimiser step with the union of the rows inside the row kernel
            self._mark()
            if not self._first_native_checked:
                self._first_native_checked = True
                self._bounded_wait(float(os.environ.get("KPRN_DP_FIRST_STEP_TIMEOUT", "60")))
            return
        buf = a.pack(cap)  # one packed tensor per rank, same length everywhere
        if self._all is None or self._all.numel() != buf.numel() * self.world or self._all.dtype != buf.dtype:
            self._all = torch.empty(buf.numel() * self.world, dtype=buf.dtype, device=buf.device)
        self._mark()
        work = None
        if self.collectives:
            work = dist.all_gather_into_tensor(self._all, buf, group=self.group, async_op=True)
        else:
            self._all.copy_(buf)
        if overlap is not None:
            overlap()      # compute that hides the exchange (xGMI is otherwise the only thing working right now)
        if work is not None:
            work.wait()    # stream-level wait: the merge below is ordered after the collective
        self._mark()
        a.merge(self._all, self.world, cap)  # union of the rows, summed in rank order
        self._mark()
        a.apply_update(opt)
        self._mark()


def shard_pairs(n_pairs, rank, world):
    """contiguous slice of the pairs of one global minibatch for `rank` (SURVEY 8e partitioning)."""
    per = (n_pairs + world - 1) // world
    lo = min(n_pairs, rank * per)
    hi = min(n_pairs, lo + per)
    return lo, hi
