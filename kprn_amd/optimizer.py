"""MyOptimizer: the reference's training loop (release/songPathRnn/model/optimizer/MyOptimizer.lua)
with the same method names and epoch log lines, driving the HIP engine through the C ABI.
"""
import sys
import time

import numpy as np

from . import _ffi


class OptimizerCallback:  # optimizer/OptimizerCallback.lua
    def __init__(self, epochHookFreq, hook, name=""):
        self.epochHookFreq, self.hook, self.name = epochHookFreq, hook, name


class MyOptimizer:
    """MyOptimizer(engine, trainingOptions, opt)   (MyOptimizer.lua:13-72)

    trainingOptions: dict(numEpochs, epochHooks=[OptimizerCallback], minibatchsize)
    opt:             kprn_opt (model.opt_from_flags)
    dp:              optional kprn_amd.dp.DataParallel -- every rank then feeds ITS shard of each minibatch
    """

    def __init__(self, engine, trainingOptions, opt, startIteration=1, gradientStepCounter=100, dp=None, out=sys.stdout):
        assert trainingOptions is not None
        self.engine = engine
        self.trainingOptions = trainingOptions
        self.opt = opt
        self.startIteration = startIteration
        self.gradientStepCounter = gradientStepCounter
        self.totalError = 0.0
        self.dp = dp
        self.out = out
        self._cache = {}  # device-resident batches keyed by (file, offset): only when the order is the same every epoch
        self._slots = [None, None]   # streaming feed (shuffled order): two device slots ...
        self._stage = [None, None]   # ... and their page-locked staging buffers (BatcherFileList.lua:53-60 preallocates likewise)
        for hook in trainingOptions.get("epochHooks", []):  # MyOptimizer.lua:65-70
            if hook.epochHookFreq == 1:
                hook.hook(0)

    def zeroPadTokens(self):  # MyOptimizer.lua:74-93
        self.engine.zero_pad_tokens()

    def _device_batch(self, inputs, targets, key):
        if key is not None and key in self._cache:
            return self._cache[key]
        b = self.engine.batch(inputs, targets)
        if key is not None:
            if len(self._cache) > 4096:
                for old in self._cache.values():
                    old.free()
                self._cache.clear()
            self._cache[key] = b
        return b

    def _feed(self, k, inputs, targets):
        """BatcherFileList:populateGPUTensor (BatcherFileList.lua:78-96): copy the minibatch into preallocated page-locked
        buffers and hand it to the engine's feed stream; slot k is refilled in place."""
        inputs = np.asarray(inputs)
        targets = np.asarray(targets)
        st = self._stage[k]
        if st is None or st[0].size < inputs.size or st[1].size < targets.size:
            st = (self.engine.host_array((max(inputs.size, 1) * 2,), np.int32), self.engine.host_array((max(targets.size, 1) * 2,), np.float32))
            self._stage[k] = st
        hi = st[0][:inputs.size].reshape(inputs.shape)
        hl = st[1][:targets.size].reshape(targets.shape)
        hi[...] = inputs   # (float64 ids of a .torch file are converted here, once per batch)
        hl[...] = targets
        self._slots[k] = self.engine.feed(hi, hl, slot=self._slots[k])
        return self._slots[k]

    def trainBatch(self, inputs, targets, classId=1, key=None, want_loss=True):
        """MyOptimizer.lua:177-221: zeroPad; fEval{zeroGrad, forward, BCE, backward, clip/L2}; optim step; zeroPad."""
        assert inputs is not None
        assert targets is not None
        b = inputs if isinstance(inputs, _ffi.Batch) else self._device_batch(inputs, targets, key)
        if self.dp is not None:
            # ranks may hold shards of different sizes (last batch of a file): the global pair count is agreed per step unless the
            # DataParallel object was built with equal_shards=True
            self.dp.train_step(b, self.opt, classId)
            err = self.engine.read_loss() if want_loss else None
        else:
            err = self.engine.train_step(b, self.opt, classId, want_loss=want_loss)
        if err is not None:
            self.totalError += err
        return err

    def train(self, trainBatcher):  # MyOptimizer.lua:95-169
        prevTime = time.time()
        numProcessed = 0
        print("Making a pass of the data to count the batches", file=self.out)
        totalBatches = 0
        while trainBatcher.getBatch() is not None:
            totalBatches += 1
        print(f"Total num batches {totalBatches}", file=self.out)
        trainBatcher.reset()
        if self.dp is not None and not self.dp.bounded and hasattr(trainBatcher, "max_batch_positions"):
            # a true bound of the rows any step of any rank touches (+ the virtual prefix positions), agreed once
            self.dp.set_capacity(min(trainBatcher.max_batch_positions() + 8, self.engine.cfg.Ve), bound=True)
        i = self.startIteration
        history = []
        while i <= self.trainingOptions["numEpochs"]:
            self.totalError = 0.0
            batch_counter = 0
            gradientStepCounter = 0
            # A shuffled epoch never repeats a batch: each one is streamed -- the next batch's upload + index build run on the
            # engine's feed stream while this batch trains (two slots).  A fixed order keeps the batches resident in HBM.
            streaming = bool(trainBatcher.doShuffle)
            got = trainBatcher.getBatch(with_key=True)
            k = 0
            fed = self._feed(k, got[1], got[0]) if (streaming and got is not None) else None
            while got is not None:
                targets, inputs, num, classId, key = got
                nxt = trainBatcher.getBatch(with_key=True)
                if streaming:
                    cur = fed
                    if nxt is not None:
                        fed = self._feed(k ^ 1, nxt[1], nxt[0])   # queued before this step: runs under it
                    k ^= 1
                    inputs = cur
                batch_counter += 1
                numProcessed += targets.size
                # (file, offset) names the same rows in every epoch of a fixed order -- Batcher.epoch is not part of the key
                cache_key = (key[0], key[1]) if not streaming else None
                self.trainBatch(inputs, targets, classId, cache_key)
                got = nxt
                gradientStepCounter += 1
                if gradientStepCounter % self.gradientStepCounter == 0:
                    avgError = self.totalError / gradientStepCounter
                    print("Printing after %d gradient steps\navg loss in epoch = %f\n" % (self.gradientStepCounter, avgError), file=self.out)
            avgError = self.totalError / max(batch_counter, 1)
            currTime = time.time()
            elapsed = currTime - prevTime
            rate = numProcessed / max(elapsed, 1e-9)
            numProcessed = 0
            prevTime = currTime
            print("\nIter: %d\navg loss in epoch = %f\ntotal elapsed = %f\ntime per batch = %f" %
                  (i, avgError, elapsed, elapsed / max(batch_counter, 1)), file=self.out)
            print("examples/sec = %f" % rate, file=self.out)
            history.append(avgError)
            self.postEpoch()
            for hook in self.trainingOptions.get("epochHooks", []):
                if i % hook.epochHookFreq == 0:
                    hook.hook(i)
            trainBatcher.reset()
            i += 1
        return history

    def postEpoch(self):  # MyOptimizer.lua:171-173
        pass
