"""Batch feed: the reference's Batcher / BatcherFileList with the same names, arguments, return
values and iteration order (release/songPathRnn/model/batcher/Batcher.lua,
BatcherFileList.lua), over .torch / .int / .npz path files.

Host side only: batches come back as numpy arrays exactly like the CPU path of the reference
(narrow views of the file's tensors, Batcher.lua:50-51); kprn_amd.optimizer moves them to HBM
(the reference's populateGPUTensor, BatcherFileList.lua:78-96) and keeps them there when the
order is deterministic.
"""
import os

import numpy as np

from . import formats


class Batcher:
    """Batcher(filePath, batchSize, shuffle)  -- Batcher.lua:9-32"""

    def __init__(self, filePath, batchSize, shuffle, rng=None, check_ids=True):
        self.filePath = filePath
        self.labels, self.data, self.classId = formats.load_path_file(filePath, check_ids=check_ids)
        self.doShuffle = bool(shuffle)
        self.labelDimension = 1 if self.labels.ndim == 1 else self.labels.shape[1]
        self.numPaths = self.data.shape[1]
        self.numTokensInPath = self.data.shape[2]
        self.numFeatureTemplates = self.data.shape[3]
        self.rng = rng if rng is not None else np.random.default_rng()
        self.epoch = 0
        self.perm = None   # shuffled order: row perm[i] of the file's tensors stands at position i
        if self.doShuffle:
            self.shuffle()
        self.batchSize = int(batchSize)
        self.curStart = 0

    def shuffle(self):  # Batcher.lua:35-41
        """The reference permutes the tensors themselves (labels:index(1, inds), data:index(1, inds)); here the permutation is
        kept and composed -- same order, but an epoch's reshuffle moves 8 bytes per pair instead of the whole file, and the
        engine gathers each minibatch's rows itself (kprn_batch_feed_rows_async)."""
        if self.doShuffle:
            inds = self.rng.permutation(self.labels.shape[0])
            self.perm = inds if self.perm is None else self.perm[inds]

    def _next_span(self):
        dataSize = self.labels.shape[0]
        start = self.curStart
        if start >= dataSize:
            return None
        end = min(start + self.batchSize, dataSize)
        self.curStart = end
        return start, end

    def getBatch(self):  # Batcher.lua:43-54 -> labels[B], data[B,P,T,F] (views; copies of the permuted rows when shuffled) or None
        span = self._next_span()
        if span is None:
            return None
        start, end = span
        if self.perm is None:
            return self.labels[start:end], self.data[start:end]
        rows = self.perm[start:end]
        return self.labels[rows], self.data[rows]

    def getBatchRows(self):
        """-> (labels of the whole file, data of the whole file, rows of this minibatch) or None: what getBatch would return is
        (labels[rows], data[rows]); nothing is copied"""
        span = self._next_span()
        if span is None:
            return None
        start, end = span
        rows = np.arange(start, end, dtype=np.int64) if self.perm is None else self.perm[start:end]
        return self.labels, self.data, rows

    def reset(self):  # Batcher.lua:56-59
        self.curStart = 0
        self.epoch += 1
        if self.doShuffle:
            self.shuffle()

    def getSizes(self):  # Batcher.lua:62
        return self.labelDimension, self.numPaths, self.numTokensInPath, self.numFeatureTemplates

    def getClassId(self):  # Batcher.lua:65
        return self.classId


class BatcherFileList:
    """BatcherFileList(dataDir, batchSize, shuffle, maxBatches, useCuda, filelist) -- BatcherFileList.lua:10-51.

    getBatch() -> (labels, data, n, classId) or None at the end of an epoch; reset() starts the next.
    Iteration order is the reference's CPU path (BatcherFileList.lua:133-146): files in `index` order,
    each drained completely before the next; `index` is a fresh permutation per epoch when shuffling.
    """

    def __init__(self, dataDir, batchSize, shuffle, maxBatches, useCuda, filelist, seed=None, check_ids=True):
        fileList = os.path.join(dataDir, filelist)
        self.doShuffle = bool(shuffle)
        self.batchSize = int(batchSize)
        self.useCuda = bool(useCuda)
        self.rng = np.random.default_rng(seed)
        self.batchers = []
        with open(fileList) as f:
            for line in f:
                line = line.strip()
                if line:
                    self.batchers.append(Batcher(os.path.join(dataDir, line), batchSize, self.doShuffle, self.rng, check_ids=check_ids))
        self.numBatchers = len(self.batchers)
        self.maxBatches = self.numBatchers  # :37 (the maxBatches argument is ignored by the reference too)
        self._new_epoch_order()

    def _new_epoch_order(self):
        self.startIndex = 1
        self.endIndex = self.maxBatches
        self.index = (self.rng.permutation(self.numBatchers) + 1) if self.doShuffle else np.arange(1, self.numBatchers + 1)
        self.currentIndex = 1

    def reset(self):  # BatcherFileList.lua:99-116
        self._new_epoch_order()
        for b in self.batchers:
            b.reset()

    def max_batch_positions(self):
        """largest B*P*T of any minibatch this list can hand out: an upper bound of the entity rows one step touches
        (the fixed packing capacity of the data-parallel row exchange, kprn_amd/dp.py)"""
        return max((min(self.batchSize, b.labels.shape[0]) * b.numPaths * b.numTokensInPath for b in self.batchers), default=0)

    def getBatchInternal(self, rows=False):  # CPU path, BatcherFileList.lua:133-146
        if self.currentIndex >= self.numBatchers:
            self.currentIndex = 1
        for i in range(self.currentIndex, self.numBatchers + 1):
            batcher = self.batchers[self.index[i - 1] - 1]
            got = batcher.getBatchRows() if rows else batcher.getBatch()
            if got is not None:
                self.currentIndex = i
                n = got[2].shape[0] if rows else got[0].shape[0]
                return got, n, batcher.getClassId(), (self.index[i - 1] - 1, batcher.curStart - n, batcher.epoch)
        return None

    def getBatch(self, with_key=False, rows=False):  # BatcherFileList.lua:169-188
        """-> (labels, data, n, classId[, key]) or None.  rows=True: (labels_all, data_all, rows) of the file instead of the two
        minibatch tensors -- (labels, data, rows, n, classId[, key]) -- for consumers that gather the rows themselves."""
        while self.startIndex <= self.numBatchers:
            got = self.getBatchInternal(rows)
            if got is None:
                self.startIndex = self.endIndex + 1
                self.endIndex = min(self.startIndex + self.maxBatches - 1, self.numBatchers)
                self.currentIndex = self.startIndex
            else:
                arrays, n, classId, key = got
                out = tuple(arrays) + (n, classId)
                return out + (key,) if with_key else out
        return None
