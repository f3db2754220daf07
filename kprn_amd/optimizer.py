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
        self._cache = {}  # device-resident batches keyed by (file, offset, shuffle epoch)
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
                self._cache.clear()
            self._cache[key] = b
        return b

    def trainBatch(self, inputs, targets, classId=1, key=None, want_loss=True):
        """MyOptimizer.lua:177-221: zeroPad; fEval{zeroGrad, forward, BCE, backward, clip/L2}; optim step; zeroPad."""
        assert inputs is not None
        assert targets is not None
        b = inputs if isinstance(inputs, _ffi.Batch) else self._device_batch(inputs, targets, key)
        if self.dp is not None:
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
        i = self.startIteration
        history = []
        while i <= self.trainingOptions["numEpochs"]:
            self.totalError = 0.0
            batch_counter = 0
            gradientStepCounter = 0
            while True:
                got = trainBatcher.getBatch(with_key=True)
                if got is None:
                    break
                targets, inputs, num, classId, key = got
                batch_counter += 1
                numProcessed += targets.size
                cache_key = key if not trainBatcher.doShuffle else None
                self.trainBatch(inputs, targets, classId, cache_key)
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
