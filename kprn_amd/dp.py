"""Data-parallel training step over torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

New design -- the reference is single-device.  One process per GPU; each
rank owns a full parameter replica and a disjoint slice of the (user,item) pairs (pairs are
independent units: model/module/MapReduce.lua:24-47 never mixes them).  Per step:

  1. every rank runs zeroGrad + forward + BCE + backward on ITS pairs, with the loss scaled by
     1/B_global (the mean over the global minibatch, nn.BCECriterion sizeAverage);
  2. dense gradients (type/relation tables, LSTM, head: ONE contiguous device buffer, ~0.3 MB
     for D=H=64,L=2) -> one all-reduce(sum);
  3. entity-table gradients are row-sparse -> each rank packs {count, row ids, grad rows} of the rows it
     touched into ONE fixed-capacity buffer, ONE all-gather, then every rank merges all ranks' rows:
     union of the ids by a stable sort, sums in rank order (identical addition order everywhere =>
     replicas stay bit-identical);
  4. the optimiser step runs locally on the summed gradient (MyOptimizer.lua:196-219).

The collective calls only see an "adapter" that exposes the engine's buffers as torch tensors,
so the same code is exercised on CPU (gloo, world_size 2) with a numpy-backed adapter.
"""
import numpy as np
import torch
import torch.distributed as dist


class _DevArray:
    """minimal __cuda_array_interface__ carrier for a raw device pointer owned by libkprn."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 3, "strides": None}


def wrap_device(ptr, n, kind, device):
    typestr = {"f32": "<f4", "i32": "<i4"}[kind]
    return torch.as_tensor(_DevArray(ptr, n, typestr), device=device)


class GpuAdapter:
    """Exposes one kprn Engine's gradient buffers as CUDA tensors (zero-copy)."""

    def __init__(self, engine, device):
        self.e = engine
        self.device = torch.device(device)
        ptr, n = engine.dense_grad_buffer()
        self.dense = wrap_device(ptr, n, "f32", self.device)
        self.de = engine.cfg.de
        self._cap = 0

    def backward(self, batch, class_id, bce_literal, inv_batch):
        self.e.backward(batch, class_id, bce_literal, inv_batch, want_loss=False)

    def dense_grads(self):
        return self.dense

    def local_rows(self):
        return self.e.sparse_grad_capacity()

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
    def __init__(self, adapter, group=None):
        self.a = adapter
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # with a process group (even of size 1: bench.py --force-dp) every collective below is really issued, so the
        # single-GPU dev box exercises the exact call sequence of the N-GPU run
        self.collectives = dist.is_initialized()
        self.capacity = 0
        self._all = None

    def set_capacity(self, local_max_rows):
        """fixed per-rank packing capacity = max over ranks of the largest per-step touched-row count."""
        t = torch.tensor([int(local_max_rows)], dtype=torch.int64)
        if self.collectives:
            if dist.get_backend(self.group) == "nccl":
                t = t.to(self.a.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        self.capacity = max(1, int(t.item()))
        self._all = None
        return self.capacity

    def train_step(self, batch, opt, class_id=1, global_pairs=None, overlap=None):
        """one data-parallel MyOptimizer:trainBatch; `batch` holds THIS rank's pairs.
        overlap: optional callable that ENQUEUES work which does not depend on this step's update (e.g. a scoring pass
        with the pre-update parameters); it runs while the entity-row all-gather is in flight."""
        a = self.a
        gp = global_pairs if global_pairs is not None else batch.B * self.world
        a.zero_pad()  # MyOptimizer.lua:181
        a.backward(batch, class_id, bool(opt.bce_literal), 1.0 / float(gp))
        if self.collectives:
            dist.all_reduce(a.dense_grads(), op=dist.ReduceOp.SUM, group=self.group)
        if self.capacity <= 0:
            self.set_capacity(a.local_rows())
        cap = self.capacity
        buf = a.pack(cap)  # one packed tensor per rank, same length everywhere
        if self._all is None or self._all.numel() != buf.numel() * self.world or self._all.dtype != buf.dtype:
            self._all = torch.empty(buf.numel() * self.world, dtype=buf.dtype, device=buf.device)
        work = None
        if self.collectives:
            work = dist.all_gather_into_tensor(self._all, buf, group=self.group, async_op=True)
        else:
            self._all.copy_(buf)
        if overlap is not None:
            overlap()      # compute that hides the exchange (xGMI is otherwise the only thing working right now)
        if work is not None:
            work.wait()    # stream-level wait: the merge below is ordered after the collective
        a.merge(self._all, self.world, cap)  # union of the rows, summed in rank order
        a.apply_update(opt)


def shard_pairs(n_pairs, rank, world):
    """contiguous slice of the pairs of one global minibatch for `rank` (SURVEY 8e partitioning)."""
    per = (n_pairs + world - 1) // world
    lo = min(n_pairs, rank * per)
    hi = min(n_pairs, lo + per)
    return lo, hi
