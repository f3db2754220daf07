"""Ranking evaluation chain after scoring, Python 3 restatement of
release/songPathRnn/eval/combine_result.py, resort.py and eval_score.py.
"""
import heapq
import math


def combine_result(entity_lines, result_lines):
    """combine_result.py:24-27: positional join -> "user \\t item \\t label \\t score"."""
    out = []
    for e, r in zip(entity_lines, result_lines):
        el = e.strip().split("\t")
        rl = r.strip().split("\t")
        out.append(el[1] + "\t" + el[2] + "\t" + rl[-1] + "\t" + rl[-2] + "\n")
    return out


def resort(combined_lines, user_ids):
    """resort.py:22-43: keep users in user_ids; sort by (int(user), -score); score re-printed via str(float)."""
    users = set(u.strip() for u in user_ids)
    items = []
    for line in combined_lines:
        ll = line.strip().split("\t")
        if ll[0] in users:
            items.append((int(ll[0]), ll[1], ll[2], float(ll[3])))
    items.sort(key=lambda x: (x[0], -x[-1]))
    return [str(i[0]) + "\t" + i[1] + "\t" + str(i[2]) + "\t" + str(i[3]) + "\n" for i in items]


def hit_ndcg(scores, k):
    """eval_score.py:20-46: the positive is index 0 among 1 + 100 candidates; heapq.nlargest keeps the first-seen
    element on ties, so the positive wins ties; sum(scores) == 0 -> (0, 0); ndcg = ln2 / ln(rank + 2)."""
    if sum(scores) == 0:
        return 0.0, 0.0
    top = heapq.nlargest(k, range(len(scores)), key=lambda i: scores[i])
    if 0 in top:
        rank = top.index(0)
        return 1.0, math.log(2) / math.log(rank + 2)
    return 0.0, 0.0


def eval_samples(score_of, samples, ks=range(1, 16)):
    """eval_score.py:97-129.  score_of: dict (user,item)->score; samples: iterable of (user, pos_item, [neg items]).
    A sample is skipped if the positive or any negative is unscored (:101-109)."""
    hits = {k: 0.0 for k in ks}
    ndcgs = {k: 0.0 for k in ks}
    n = 0
    for user, pos, negs in samples:
        keys = [(user, pos)] + [(user, x) for x in negs]
        if any(kv not in score_of for kv in keys):
            continue
        sc = [score_of[kv] for kv in keys]
        n += 1
        for k in ks:
            h, d = hit_ndcg(sc, k)
            hits[k] += h
            ndcgs[k] += d
    if n == 0:
        return {k: 0.0 for k in ks}, {k: 0.0 for k in ks}, 0
    return {k: hits[k] / n for k in ks}, {k: ndcgs[k] / n for k in ks}, n
