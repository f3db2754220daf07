"""MyOptimizer: the reference's training loop (release/songPathRnn/model/optimizer/MyOptimizer.lua)
with the same method names and epoch log lines, driving the HIP engine through the C ABI.
"""
import sys
import time


from . import _ffi


class OptimizerCallback:  # optimizer/OptimizerCallback.lua
    def __init__(self, epochHookFreq, hook, name=""):
        self.epochHookFreq, self.hook, self.name = epochHookFreq, hook, name


class MyOptimizer:
    """MyOptimizer(engine, trainingOptions, opt)   (MyOptimizer.lua:13-72)

    trainingOptions: dict(numEpochs, epochHooks=[OptimizerCallback], minibatchsize)
    opt:             kprn_opt (model.opt_from_flags)
    dp:              optional kprn_amd.dp.DataParallel -- every rank then feeds ITS shard of each minibatch.  Shards of a file's last
                     minibatch are ragged (3 and 2 pairs of 5), so the loss scale 1 / (pairs of the GLOBAL minibatch) is agreed per
                     step: the DataParallel object is switched to equal_shards=False unless trainingOptions["equalShards"] promises
                     that every rank always holds the same number of pairs.
    """
    FEED_AHEAD = 3

    def __init__(self, engine, trainingOptions, opt, startIteration=1, gradientStepCounter=100, dp=None, out=sys.stdout):
        assert trainingOptions is not None
        self.engine = engine
        self.trainingOptions = trainingOptions
        self.opt = opt
        self.startIteration = startIteration
        self.gradientStepCounter = gradientStepCounter
        self.totalError = 0.0
        self.dp = dp
        if dp is not None and not trainingOptions.get("equalShards", False):
            dp.equal_shards = False   # (B_local * world is the global pair count only when every shard has the same size)
        self.out = out
        self._cache = {}  # device-resident batches keyed by (file, offset): only when the order is the same every epoch
        # streaming feed (shuffled order): a ring of device slots refilled in turn (BatcherFileList.lua:53-60 preallocates its GPU
        # tensors likewise); FEED_AHEAD minibatches are in the feed's hands while one trains -- a feed (row gather + index build on
        # the host cores + one upload) takes about as long as a step, so one in flight would not hide it
        self._slots = [None] * (self.FEED_AHEAD + 1)
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

    def trainBatch(self, inputs, targets, classId=1, key=None, want_loss=True):
        """MyOptimizer.lua:177-221: zeroPad; fEval{zeroGrad, forward, BCE, backward, clip/L2}; optim step; zeroPad."""
        assert inputs is not None
        assert targets is not None or isinstance(inputs, _ffi.Batch)
        b = inputs if isinstance(inputs, _ffi.Batch) else self._device_batch(inputs, targets, key)
        if self.dp is not None:
            # ranks may hold shards of different sizes (last batch of a file): the global pair count is agreed per step (see __init__)
            self.dp.train_step(b, self.opt, classId)
            err = self.engine.read_loss() if want_loss else None
        else:
            err = self.engine.train_step(b, self.opt, classId, want_loss=want_loss)
        if err is not None:
            self.totalError += err
        return err

    def _feed_rows(self, k, labels, data, rows):
        """BatcherFileList:populateGPUTensor (BatcherFileList.lua:78-96) for a shuffled order: the engine's feed threads gather
        the minibatch's rows out of the file's arrays into the slot's page-locked upload image; slot k is refilled in place."""
        self._slots[k] = self.engine.feed_rows(data, labels, rows, slot=self._slots[k])
        return self._slots[k]

    def _epoch_error(self):
        """(totalError, steps) of the steps since the last call: summed on the device (kprn_read_loss_sum), so that a step costs
        the host no synchronisation (MyOptimizer.lua:199 adds err to totalError on the host after every step)"""
        s, n = self.engine.loss_sum(reset=True)
        self.totalError += s
        return n

    def train(self, trainBatcher):  # MyOptimizer.lua:95-169
        prevTime = time.time()
        numProcessed = 0
        print("Making a pass of the data to count the batches", file=self.out)
        totalBatches = 0
        while trainBatcher.getBatch(rows=True) is not None:
            totalBatches += 1
        print(f"Total num batches {totalBatches}", file=self.out)
        trainBatcher.reset()
        if self.dp is not None and not self.dp.bounded and hasattr(trainBatcher, "max_batch_positions"):
            # a true bound of the rows any step of any rank touches (+ the virtual prefix positions), agreed once
            self.dp.set_capacity(min(trainBatcher.max_batch_positions() + 8, self.engine.cfg.Ve), bound=True)
        self.engine.set_option("loss_accumulate", "1")
        self.engine.loss_sum(reset=True)
        try:
            return self._epochs(trainBatcher, prevTime)
        finally:   # (an exception out of a step -- a bad id, KPRN_E_INDEX -- must not leave the engine accumulating)
            self.engine.set_option("loss_accumulate", "0")

    def _epochs(self, trainBatcher, prevTime):
        numProcessed = 0
        i = self.startIteration
        history = []
        while i <= self.trainingOptions["numEpochs"]:
            self.totalError = 0.0
            batch_counter = 0
            gradientStepCounter = 0
            # A shuffled epoch never repeats a batch: each one is streamed -- the next batch's row gather, index build and upload
            # run in the engine's feed threads while earlier batches train.  A fixed order keeps the batches resident in HBM.
            streaming = bool(trainBatcher.doShuffle)
            ahead = []                    # minibatches handed to the feed, oldest first: (batcher tuple, slot)
            k = 0
            def fill():                   # keep FEED_AHEAD minibatches in flight behind the one being trained
                nonlocal k
                while len(ahead) < (self.FEED_AHEAD if streaming else 1):   # (+ the one being trained = the ring's slots)
                    g = trainBatcher.getBatch(with_key=True, rows=streaming)
                    if g is None:
                        return
                    ahead.append((g, self._feed_rows(k, g[0], g[1], g[2]) if streaming else None))
                    k = (k + 1) % len(self._slots)
            fill()
            while ahead:
                got, fed = ahead.pop(0)
                if streaming:
                    _, _, _, num, classId, key = got
                    inputs, targets = fed, None
                    fill()                # queued before this step: they run under it
                else:
                    targets, inputs, num, classId, key = got
                    fill()
                batch_counter += 1
                numProcessed += num
                # (file, offset) names the same rows in every epoch of a fixed order -- Batcher.epoch is not part of the key
                cache_key = (key[0], key[1]) if not streaming else None
                self.trainBatch(inputs, targets, classId, cache_key, want_loss=False)
                gradientStepCounter += 1
                if gradientStepCounter % self.gradientStepCounter == 0:
                    self._epoch_error()
                    avgError = self.totalError / gradientStepCounter
                    print("Printing after %d gradient steps\navg loss in epoch = %f\n" % (self.gradientStepCounter, avgError), file=self.out)
            self._epoch_error()
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
