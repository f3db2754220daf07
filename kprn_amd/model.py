"""Model construction from the reference's command-line surface (release/songPathRnn/model/OneModel.lua).

`parse_flags` accepts the torch.CmdLine flags of OneModel.lua:27-87 with the same names and
defaults; `build_engine` turns them into a kprn Engine the way OneModel.lua:204-309 builds
predictor_net / reducer / training_net.  Options the engine does not implement fail loudly with
the library's KPRN_E_UNSUPPORTED (dropout) instead of silently doing something else.
"""
import argparse

from . import _ffi

LABEL_DIMENSION = 46  # OneModel.lua:119


def flag_parser():
    p = argparse.ArgumentParser(prefix_chars="-", add_help=True, description="songPathRnn flags (OneModel.lua:27-87)")
    a = p.add_argument
    a("-dataDir", default=""); a("-minibatch", type=int, default=32); a("-testTimeMinibatch", type=int, default=32)
    a("-numRowsToGPU", type=int, default=1); a("-gpuid", type=int, default=-1); a("-lazyCuda", type=int, default=0)
    a("-relationVocabSize", type=int, default=51503); a("-entityTypeVocabSize", type=int, default=2267)
    a("-entityVocabSize", type=int, default=1540261)
    a("-relationEmbeddingDim", type=int, default=50); a("-entityTypeEmbeddingDim", type=int, default=50)
    a("-entityEmbeddingDim", type=int, default=50)
    a("-numFeatureTemplates", type=int, default=-1); a("-numEntityTypes", type=int, default=-1)
    a("-learningRate", type=float, default=0.001); a("-learningRateDecay", type=float, default=0.0)
    a("-tokenFeatures", type=int, default=1); a("-evaluationFrequency", type=int, default=10)
    a("-model", default=""); a("-exptDir", default=""); a("-initModel", default="")
    a("-paramInit", type=float, default=0.1); a("-startIteration", type=int, default=1); a("-saveFrequency", type=int, default=50)
    a("-embeddingL2", type=float, default=0.0001); a("-l2", type=float, default=0.0001)
    a("-architecture", default="rnn"); a("-numEpochs", type=int, default=300); a("-batchesPerEpoch", type=int, default=500)
    a("-rnnType", default="rnn"); a("-rnnDepth", type=int, default=1); a("-rnnHidSize", type=int, default=50)
    a("-useAdam", type=int, default=0); a("-epsilon", type=float, default=1e-8)
    a("-useGradClip", type=int, default=1); a("-gradClipNorm", type=float, default=5.0)
    a("-gradientStepCounter", type=int, default=100)
    a("-includeEntityTypes", type=int, default=1); a("-includeEntity", type=int, default=-1)
    a("-topK", type=int, default=0); a("-K", type=int, default=5)
    a("-package_path", default=""); a("-createExptDir", type=int, default=1)
    a("-useReLU", type=int, default=1); a("-rnnInitialization", type=int, default=1); a("-regularize", type=int, default=1)
    a("-numLayers", type=int, default=1); a("-useDropout", type=int, default=0); a("-dropout", type=float, default=0.0)
    # engine-side additions (not in the reference)
    a("-seed", type=int, default=12345); a("-entityUpdate", type=int, default=0)
    a("-checkpointFormat", default="native", choices=["native", "t7", "both"],
      help="the native checkpoint is always written at <model>-latest; t7 / both ALSO write <model>-latest.t7, the parameters in a Torch7 {embeddingLayer, predictor_net} container (the reference writes its container at <model>-latest itself)")
    return p


def parse_flags(argv=None):
    return flag_parser().parse_args(argv)


RNN_TYPES = {"lstm": 0, "rnn": 1, "gru": 2}


def reducer_of_train_flag(topK):
    """OneModel.lua:284-293: -topK 1 -> TopK+Mean, 2 -> LogSumExp, ANYTHING ELSE -> nn.Max"""
    return {1: 1, 2: 2}.get(int(topK), 0)


def reducer_of_score_flag(top_k):
    """test_from_checkpoint.lua:69-79: -top_k 0 -> nn.Max, 2 -> LogSumExp, ANYTHING ELSE -> TopK+Mean"""
    return {0: 0, 2: 2}.get(int(top_k), 1)


REDUCER_NAME = {0: "Reducer is max pool", 1: "Reducer is topK", 2: "Reducer is LogSumExp"}

NATIVE_MAGIC = b"KPRNAMD1"


def load_checkpoint(eng, path):
    """-initModel / -model_path: the native format (kprn_save) or a reference Torch7 checkpoint {embeddingLayer, predictor_net}
    (OneModel.lua:392-400), told apart by the file's first bytes"""
    with open(path, "rb") as f:
        head = f.read(8)
    if head == NATIVE_MAGIC:
        eng.load(path)
        return "native"
    from . import formats
    params = formats.checkpoint_params(path, num_layers=eng.cfg.L)
    lay = eng.layout()
    missing = [n for n in lay if n not in params]
    if missing:
        raise _ffi.KprnError(_ffi.E_IO, f"{path}: checkpoint lacks {missing}")
    for n in lay:
        eng.set_param(n, params[n])
    return "t7"


def save_checkpoint_t7(eng, path):
    """PARAMETER EXCHANGE in the reference's container: a Torch7 table {embeddingLayer, predictor_net} (OneModel.lua:392-400) whose nn.* objects
    carry the weight tensors under the reference's field names, in module order -- what formats.checkpoint_params (and a Lua script that walks
    the tables) reads back.  NOT a runnable checkpoint for eval/test_from_checkpoint.lua: the Element-Research rnn modules in it have no
    recurrentModule graph, sharedClones or step state, so model:forward would fail; rebuild the model with OneModel.lua's constructor and copy
    the tensors in."""
    from . import formats
    formats.write_checkpoint(path, {n: eng.get_param(n) for n in eng.layout()}, num_entity_types=eng.cfg.num_types, use_relu=eng.cfg.use_relu)


def build_engine(params, rank=0, world=1, device_id=None, stream=None):
    """OneModel.lua:204-309.  The embedding variants of OneModel.lua:207-219: a table that is left out (-includeEntityTypes 0 /
    -includeEntity 0) is a table of width 0 to the engine -- x_t = [types | relations], [entities | relations] or [relations],
    parameters in the same flat order with the absent table contributing nothing."""
    inc_types, inc_ent = params.includeEntityTypes == 1, params.includeEntity == 1
    if params.numEntityTypes > params.numFeatureTemplates:
        raise _ffi.KprnError(_ffi.E_ARG, "assert(numEntityTypes <= numFeatureTemplates) (OneModel.lua:107)")
    if params.useDropout != 0:
        raise _ffi.KprnError(_ffi.E_UNSUPPORTED, "dropout is off in every shipped config (config.sh:48) and is not built")
    if params.rnnType not in RNN_TYPES:
        raise _ffi.KprnError(_ffi.E_ARG, f"rnnType must be lstm, rnn or gru, got {params.rnnType}")
    if device_id is None:
        device_id = max(0, params.gpuid)
    eng = _ffi.Engine(params.entityTypeVocabSize, params.entityVocabSize, params.relationVocabSize,
                      params.entityTypeEmbeddingDim if inc_types else 0, params.entityEmbeddingDim if inc_ent else 0, params.relationEmbeddingDim,
                      params.rnnHidSize, params.numLayers, F=params.numFeatureTemplates, num_types=params.numEntityTypes,
                      C_=LABEL_DIMENSION, reducer=getattr(params, "reducer", reducer_of_train_flag(params.topK)), K=params.K, rnn_type=RNN_TYPES[params.rnnType],
                      use_relu=params.useReLU, rnn_init=params.rnnInitialization,
                      device_id=device_id, rank=rank, world=world, param_init=params.paramInit, seed=params.seed, stream=stream)
    if params.initModel:
        load_checkpoint(eng, params.initModel)  # OneModel.lua:277-282
    return eng


def opt_from_flags(params):
    """optConfig / optInfo of OneModel.lua:340-384 as a kprn_opt."""
    return _ffi.make_opt(method=1 if params.useAdam == 1 else 0, lr=params.learningRate, beta1=0.9, beta2=0.999,
                         eps=params.epsilon, lr_decay=params.learningRateDecay, regularize=params.regularize,
                         use_grad_clip=params.useGradClip, grad_clip_norm=params.gradClipNorm, l2=params.l2,
                         bce_literal=0, entity_update=params.entityUpdate)
