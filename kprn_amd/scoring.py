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


ENGINE_BATCH_PATHS = 65536   # paths per engine call: the script's minibatch (512 pairs) is a memory knob of the Torch7 graph, not
                             # part of the result -- pairs are scored independently, so the engine takes them in chip-sized groups
FEED_AHEAD = 2


def score_batches(engine, batcher, class_id=1):
    """-> (labels [n], probabilities [n]) per group of pairs, in list x in-file order.  With the HIP engine every group is a
    label-less feed slot (upload + identical-prefix plan built by the host threads under the previous group's kernels), scored
    asynchronously; the probabilities of group i come back while group i + 1 runs, so the caller's formatting overlaps too."""
    streaming = hasattr(engine, "feed") and hasattr(engine, "forward_async")
    saved = [(b, b.batchSize) for b in getattr(batcher, "batchers", [])]
    try:
        if streaming:
            for b, _ in saved:       # chip-sized groups per bucket file (the caller's batch sizes are put back when the generator ends)
                b.batchSize = max(b.batchSize, ENGINE_BATCH_PATHS // max(1, b.numPaths))
        yield from _score_batches(engine, batcher, class_id, streaming)
    finally:
        for b, size in saved:
            b.batchSize = size


def _score_batches(engine, batcher, class_id, streaming):
    if not streaming:
        while True:
            got = batcher.getBatch()
            if got is None:
                return
            labs, inputs, count, _classId = got
            yield labs, engine.forward(engine.batch(inputs), class_id)["probs"]  # nn.Select(2,1) is fixed in the script (:82)
    slots = [None] * (FEED_AHEAD + 2)
    k = 0
    fed = []           # (slot, labels, count): fed, not yet scored
    running = None     # (labels, count): scored asynchronously, probabilities not yet read
    done = False
    while True:
        while not done and len(fed) < FEED_AHEAD:
            got = batcher.getBatch()
            if got is None:
                done = True
                break
            labs, inputs, count, _classId = got
            slots[k] = engine.feed(inputs, None, slot=slots[k])
            fed.append((slots[k], labs, count))
            k = (k + 1) % len(slots)
        out = None
        if running is not None:
            out = (running[0], engine.read_probs(running[1]))
        running = None
        if fed:
            slot, labs, count = fed.pop(0)
            engine.forward_async(slot, class_id)
            running = (labs, count)
        if out is not None:
            yield out
        if running is None and not fed and done:
            return


def score_lines(engine, batcher, class_id=1):
    counter = 0
    for labs, preds in score_batches(engine, batcher, class_id):
        for i in range(len(labs)):
            yield "%d\t%.5f\t%s\n" % (counter, preds[i], lua_number(labs[i]))
            counter += 1


def write_scores(engine, batcher, f, class_id=1):
    """the script's output loop (:110-118) into the binary file f; lines formatted by the host cores (kprn_format_score_lines)"""
    from . import _ffi
    counter = 0
    for labs, preds in score_batches(engine, batcher, class_id):
        f.write(_ffi.format_score_lines(counter, preds, labs))
        counter += len(labs)
    return counter


def test_from_checkpoint(engine, input_dir, test_list, out_file, minibatch=512, log=None, rank=0, world=1, barrier=None):
    """engine: built with the same -top_k reducer the script would rebuild (:69-79).

    Data-parallel scoring (new; the reference is single-device): pairs are independent units, so the FILES of the test list are
    sharded over the ranks in contiguous ranges (dp.shard_pairs), every rank scores its files with no collective and writes
    `<out_file>.part<rank>`; after `barrier()` (torch.distributed.barrier in a real run) rank 0 concatenates the parts in rank
    order = list order and renumbers the global 0-based counter, so that `out_file` is byte-identical to the single-rank file
    (the downstream join with test.list.entity is positional, eval/combine_result.py:24-27)."""
    if world <= 1:
        batcher = BatcherFileList(input_dir, minibatch, False, 1000, True, test_list, check_ids=False)   # (the engine validates every id)
        start = time.time()
        n = 0
        with open(out_file, "wb") as f:
            n = write_scores(engine, batcher, f, 1)
        if log:
            print("total cost time:", time.time() - start, file=log)
        return n
    import os
    import tempfile
    from .dp import shard_pairs
    with open(os.path.join(input_dir, test_list)) as f:
        files = [l.strip() for l in f if l.strip()]
    lo, hi = shard_pairs(len(files), rank, world)
    start = time.time()
    n = 0
    part = f"{out_file}.part{rank}"
    with open(part, "wb") as out:
        if hi > lo:
            # a list file of this rank's shard, next to the original (paths in it stay relative to input_dir)
            fd, shard_list = tempfile.mkstemp(prefix=f".{os.path.basename(test_list)}.rank{rank}.", dir=input_dir)
            try:
                with os.fdopen(fd, "w") as sl:
                    sl.write("\n".join(files[lo:hi]) + "\n")
                batcher = BatcherFileList(input_dir, minibatch, False, 1000, True, os.path.basename(shard_list), check_ids=False)
                n = write_scores(engine, batcher, out, 1)
            finally:
                os.unlink(shard_list)
    if barrier is not None:
        barrier()
    if rank == 0:
        counter = 0
        with open(out_file, "w") as f:
            for r in range(world):
                with open(f"{out_file}.part{r}") as pf:
                    for line in pf:
                        _, rest = line.split("\t", 1)
                        f.write("%d\t%s" % (counter, rest))
                        counter += 1
        for r in range(world):
            os.unlink(f"{out_file}.part{r}")
        if log:
            print("total cost time:", time.time() - start, file=log)
    return n
