"""Data-parallel step (kprn_amd/dp.py) on CPU: world_size 2, gloo, 127.0.0.1.

The collective logic -- loss scaled by the GLOBAL batch, dense all-reduce, fixed-capacity sparse
row all-gather, rank-ordered merge, local optimiser step -- is exercised with a numpy adapter that
gets its gradients from the CPU oracle (test infrastructure).  Two ranks, each fed half of the
pairs, must end exactly where one process fed the whole minibatch ends, and stay bit-identical to
each other.  The GPU adapter shares every line of DataParallel.train_step with this test; the
device-pointer plumbing is covered by tests/test_gpu_parity.py::test_two_replicas_exchange...
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kprn_amd import dp, synth
from oracle.oracle import Oracle, make_cfg, make_opt


class _Batch:
    def __init__(self, idx, labels):
        self.idx, self.labels, self.B = idx, labels, idx.shape[0]


class OracleAdapter:
    """numpy stand-in for dp.GpuAdapter: same methods, gradients from the oracle."""

    def __init__(self, cfg, theta, dense_in_pack=False):
        self.dense_in_pack = dense_in_pack
        self.o = Oracle(cfg, np.float64)
        self.cfg = cfg
        self.theta = theta.copy()
        self.lay = self.o.layout()
        off, shp = self.lay["entity_emb"]
        self.e0, self.e1 = off, off + int(np.prod(shp))
        self.de = cfg.de
        self.device = torch.device("cpu")
        self.g = np.zeros_like(self.theta)
        self.m = np.zeros_like(self.theta)
        self.v = np.zeros_like(self.theta)
        self.t = 0
        self.rows = np.zeros(0, np.int64)
        self._dense = None

    def zero_pad(self):
        self.o.zero_pad(self.theta)

    def backward(self, batch, class_id, bce_literal, inv_batch):
        _, g, _ = self.o.forward_backward(self.theta, batch.idx, batch.labels, class_id, bce_literal, inv_batch)
        self.g = g
        ge = g[self.e0:self.e1].reshape(-1, self.de)
        self.rows = np.flatnonzero(np.any(ge != 0, axis=1))
        dense = np.concatenate([g[:self.e0], g[self.e1:]])
        self._dense = torch.from_numpy(dense)

    def dense_grads(self):
        return self._dense

    def local_rows(self):
        return len(self.rows)

    def batch_rows(self, batch):
        return len(np.unique(batch.idx[..., -2]))

    def pack(self, capacity):
        """one packed float64 tensor {count, ids[cap], rows[cap*de]} (the GPU adapter packs 32-bit words)"""
        ge = self.g[self.e0:self.e1].reshape(-1, self.de)
        nd = self._dense.numel() if self.dense_in_pack else 0
        buf = torch.zeros(1 + capacity * (1 + self.de) + nd, dtype=torch.float64)
        if nd:
            buf[1 + capacity * (1 + self.de):] = self._dense   # the dense arena rides behind the rows
        n = len(self.rows)
        buf[0] = n
        buf[1:1 + n] = torch.from_numpy(self.rows.astype(np.float64))
        buf[1 + capacity:1 + capacity + n * self.de] = torch.from_numpy(ge[self.rows].ravel().copy())
        ge[self.rows] = 0
        return buf

    def merge(self, all_buf, world, capacity):
        ge = self.g[self.e0:self.e1].reshape(-1, self.de)
        nd = self._dense.numel() if self.dense_in_pack else 0
        stride = 1 + capacity * (1 + self.de) + nd
        if nd:
            tails = [all_buf[r * stride + stride - nd:(r + 1) * stride].clone() for r in range(world)]
            acc = tails[0]
            for t in tails[1:]:   # rank order
                acc = acc + t
            self._dense = acc
        for r in range(world):  # rank order => same addition order on every replica
            b = all_buf[r * stride:(r + 1) * stride].numpy()
            n = int(b[0])
            ids = b[1:1 + n].astype(np.int64)
            ge[ids] += b[1 + capacity:1 + capacity + n * self.de].reshape(n, self.de)

    def apply_update(self, opt):
        d = self._dense.numpy()
        self.g[:self.e0] = d[:self.e0]
        self.g[self.e1:] = d[self.e0:]
        self.t += 1
        self.m[:] = opt.beta1 * self.m + (1 - opt.beta1) * self.g
        self.v[:] = opt.beta2 * self.v + (1 - opt.beta2) * self.g * self.g
        step = opt.lr * np.sqrt(1 - opt.beta2 ** self.t) / (1 - opt.beta1 ** self.t)
        self.theta -= step * self.m / (np.sqrt(self.v) + opt.eps)
        self.o.zero_pad(self.theta)

    def new(self, n, dtype):
        return torch.zeros(n, dtype=torch.float64 if dtype == torch.float32 else dtype)


def _cfg():
    return make_cfg(Vt=6, Ve=80, Vr=9, dt=4, de=8, dr=4, H=16, L=2)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = _cfg()
        theta = Oracle(cfg).init_params(3, 0.3)
        idx, labels = synth.make_paths(12, 3, 4, Ve=80, seed=7)
        lo, hi = dp.shard_pairs(12, rank, world)
        a = OracleAdapter(cfg, theta, dense_in_pack=True)   # (the GPU adapter's default: one collective per step; the ragged test below runs the all-reduce form)
        d = dp.DataParallel(a)
        opt = make_opt(method=1, lr=1e-2)
        for _ in range(3):
            d.train_step(_Batch(idx[lo:hi], labels[lo:hi]), opt, 1, global_pairs=12)
        np.save(os.path.join(out_dir, f"theta{rank}.npy"), a.theta)
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_big_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    t0 = np.load(tmp_path / "theta0.npy")
    t1 = np.load(tmp_path / "theta1.npy")
    assert np.array_equal(t0, t1)  # replicas bit-identical
    cfg = _cfg()
    o = Oracle(cfg)
    theta = o.init_params(3, 0.3)
    idx, labels = synth.make_paths(12, 3, 4, Ve=80, seed=7)
    st = o.new_state()
    for _ in range(3):
        o.train_step(theta, st, make_opt(method=1, lr=1e-2), idx, labels)
    np.testing.assert_allclose(t0, theta, rtol=1e-10, atol=1e-13)


def _ragged_worker(rank, world, port, out_dir):
    """ragged shards, batches of varying size, NO capacity bound promised (a capacity frozen at the first step's
    row count aborted later steps; B * world is the wrong global batch for unequal shards)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = _cfg()
        theta = Oracle(cfg).init_params(3, 0.3)
        a = OracleAdapter(cfg, theta)
        d = dp.DataParallel(a, equal_shards=False)
        opt = make_opt(method=1, lr=1e-2)
        caps = []
        for step, n in enumerate(_RAGGED):
            idx, labels = synth.make_paths(n, 3, 4, Ve=80, seed=40 + step)
            cut = (n * 2) // 3          # rank 0: two thirds of the pairs, rank 1: the rest
            sl = slice(0, cut) if rank == 0 else slice(cut, n)
            d.train_step(_Batch(idx[sl], labels[sl]), opt, 1)
            caps.append(d.capacity)
        np.save(os.path.join(out_dir, f"rtheta{rank}.npy"), a.theta)
        np.save(os.path.join(out_dir, f"rcaps{rank}.npy"), np.array(caps))
    finally:
        dist.destroy_process_group()


_RAGGED = [2, 5, 17, 9, 30]   # pairs of the global minibatch, step by step: the touched-row count grows


def test_ragged_shards_growing_capacity_equal_one_big_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_ragged_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    t0 = np.load(tmp_path / "rtheta0.npy")
    t1 = np.load(tmp_path / "rtheta1.npy")
    assert np.array_equal(t0, t1)
    c0, c1 = np.load(tmp_path / "rcaps0.npy"), np.load(tmp_path / "rcaps1.npy")
    assert np.array_equal(c0, c1) and c0[-1] > c0[0]   # the capacity grew, identically on both ranks
    cfg = _cfg()
    o = Oracle(cfg)
    theta = o.init_params(3, 0.3)
    st = o.new_state()
    for step, n in enumerate(_RAGGED):
        idx, labels = synth.make_paths(n, 3, 4, Ve=80, seed=40 + step)
        o.train_step(theta, st, make_opt(method=1, lr=1e-2), idx, labels)
    np.testing.assert_allclose(t0, theta, rtol=1e-10, atol=1e-13)


def test_shard_pairs_partitions():
    for n, w in ((12, 2), (13, 4), (3, 8), (0, 2)):
        spans = [dp.shard_pairs(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
