"""Scoring entry point: release/songPathRnn/eval/test_from_checkpoint.lua.

Loads a checkpoint, runs model:forward on every batch of the test list (minibatch 512, no
shuffle, :47-49,57) and writes `counter \\t %.5f score \\t label` with a global 0-based counter
(:110-118).  Line order = list order x in-file row order: the downstream join is positional
(eval/combine_result.py:24-27).
"""
import time

from .batcher import BatcherFileList


def lua_number(x):
    """how Lua 5.1 concatenates a number into a string: "%.14g" (labels print as 1 / 0)."""
    return "%.14g" % float(x)


def score_lines(engine, batcher, class_id=1):
    counter = 0
    while True:
        got = batcher.getBatch()
        if got is None:
            break
        labs, inputs, count, _classId = got
        preds = engine.forward(engine.batch(inputs), class_id)["probs"]  # nn.Select(2,1) is fixed in the script (:82)
        for i in range(count):
            yield "%d\t%.5f\t%s\n" % (counter, preds[i], lua_number(labs[i]))
            counter += 1


def test_from_checkpoint(engine, input_dir, test_list, out_file, minibatch=512, log=None):
    """engine: built with the same -top_k reducer the script would rebuild (:69-79)."""
    batcher = BatcherFileList(input_dir, minibatch, False, 1000, True, test_list)
    start = time.time()
    n = 0
    with open(out_file, "w") as f:
        for line in score_lines(engine, batcher, 1):
            f.write(line)
            n += 1
    if log:
        print("total cost time:", time.time() - start, file=log)
    return n
