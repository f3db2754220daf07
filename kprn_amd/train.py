"""`python -m kprn_amd.train <flags>` == `th model/OneModel.lua <flags>` (release/songPathRnn/run_scripts/train.sh:86).

Same flags (OneModel.lua:27-87), same data layout (dataDir/train.list naming .torch / .int / .npz
files), same epoch log lines, checkpoint "<model>-latest" every saveFrequency epochs
(OneModel.lua:392-408; native format, kprn_save).
"""
import os
import sys

from . import model
from .batcher import BatcherFileList
from .optimizer import MyOptimizer, OptimizerCallback


def main(argv=None):
    params = model.parse_flags(argv)
    if params.createExptDir == 1 and params.exptDir:
        os.makedirs(params.exptDir, exist_ok=True)
        with open(os.path.join(params.exptDir, "config.txt"), "w") as f:  # OneModel.lua:128-170
            for k, v in sorted(vars(params).items()):
                f.write(f"{k}\t{v}\n")
    eng = model.build_engine(params)
    print(model.REDUCER_NAME[model.reducer_of_train_flag(params.topK)])
    print("Using Adam!" if params.useAdam == 1 else "Using adagrad!")
    trainBatcher = BatcherFileList(params.dataDir, params.minibatch, True, 100, params.gpuid != -1, "train.list", seed=params.seed, check_ids=False)   # (the engine validates every id)
    callbacks = []
    if params.model:
        def saver(i):
            path = params.model + "-latest"
            print("saving to " + path)
            eng.save(path)   # the native checkpoint is always written: it is the one this package (and its scoring CLI) is tested to load back
            if params.checkpointFormat in ("t7", "both"):
                model.save_checkpoint_t7(eng, path + ".t7")
        if params.createExptDir == 1:
            callbacks.append(OptimizerCallback(params.saveFrequency, saver, "saving"))
        else:
            print("WARNING! - createExptDir is NOT set!")
    opt = model.opt_from_flags(params)
    optimizer = MyOptimizer(eng, {"numEpochs": params.numEpochs, "epochHooks": callbacks, "minibatchsize": params.minibatch},
                            opt, startIteration=params.startIteration, gradientStepCounter=params.gradientStepCounter)
    optimizer.train(trainBatcher)
    return 0


if __name__ == "__main__":
    sys.exit(main())
