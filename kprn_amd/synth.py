"""Seeded synthetic KG path sets in the songPathRnn batch layout.

The reference ships no .torch path files, so the bench and the parity
tests use path sets with the statistics measured on the shipped sample
(release/songPathRnn/data/input/positive_matrix_sample.tsv.translated): paths per pair
~ 1 + Geometric (mean 1.75, capped at 28), real step counts 4 (73 %) or 6 (27 %),
LEFT padding to T (movie_data_format.py:250-254), first step = the pair's user, last step
= the pair's item with relation #END_RELATION (movie_data_format.py:280-284), ids 1-based
(int2torch.lua:60-63).  Layout [pairs, P, T, F] with F = numTypes type cols, entity,
relation (FeatureEmbedding.lua:51,88,31).

Vocabulary convention (release/songPathRnn/data/vocab/*.txt shifted by +1):
  types:     1..Vt-2 real, Vt-1 = #PAD_TOKEN, Vt = #UNK  (entity_type_id.txt)
  relations: 1..Vr-3 real, Vr-2 = #UNK, Vr-1 = #PAD_TOKEN, Vr = #END_RELATION (all_relation_id.txt)
  entities:  1..Ve-2 real, Ve-1 = #UNK_ENTITY, Ve = #PAD_TOKEN (format_entity_pair.py:13)
"""
import numpy as np

SEED = 12345  # OneModel.lua:115


def zipf_ids(rng, n, vmax, a=1.05):
    """Zipf(a)-distributed ids in 1..vmax (KG hubs), by inverse-CDF on a truncated power law."""
    u = rng.random(n)
    # continuous approximation of a truncated zeta distribution
    e = 1.0 - a
    x = ((vmax ** e - 1.0) * u + 1.0) ** (1.0 / e)
    return np.clip(x.astype(np.int64), 1, vmax).astype(np.int32)


def make_paths(n_pairs, P, T, F=3, Vt=6, Ve=10000, Vr=9, num_types=1, seed=SEED, real_len=None):
    """One bucket: n_pairs pairs, exactly P paths each -> (idx int32 [n_pairs,P,T,F], labels f32 [n_pairs])."""
    rng = np.random.default_rng(seed)
    N = n_pairs * P
    idx = np.empty((n_pairs, P, T, F), dtype=np.int32)
    pad_type, pad_ent, pad_rel, end_rel = Vt - 1, Ve, Vr - 1, Vr
    # real length per path
    if real_len is None:
        choices = np.array([min(4, T), T], dtype=np.int32)
        ell = choices[(rng.random(N) >= 0.73).astype(np.int32)]
    else:
        ell = np.full(N, real_len, dtype=np.int32)
    ell = ell.reshape(n_pairs, P)
    users = zipf_ids(rng, n_pairs, Ve - 2)
    items = zipf_ids(rng, n_pairs, Ve - 2)
    types = rng.integers(1, max(2, Vt - 1), size=(n_pairs, P, T, num_types), dtype=np.int32)
    ents = zipf_ids(rng, N * T, Ve - 2).reshape(n_pairs, P, T)
    rels = rng.integers(1, max(2, Vr - 2), size=(n_pairs, P, T), dtype=np.int32)
    t = np.arange(T, dtype=np.int32)[None, None, :]
    first = (T - ell)[:, :, None]           # index of the first real step
    is_pad = t < first
    is_first = t == first
    is_last = t == (T - 1)
    ents = np.where(is_first, users[:, None, None], ents)
    ents = np.where(is_last, items[:, None, None], ents)
    rels = np.where(is_last, end_rel, rels)
    ents = np.where(is_pad, pad_ent, ents)
    rels = np.where(is_pad, pad_rel, rels)
    types = np.where(is_pad[..., None], pad_type, types)
    tcol0 = F - num_types - 2
    idx[..., :] = pad_type  # any untouched leading feature cols (F > numTypes + 2) are ignored by the model
    idx[..., tcol0:tcol0 + num_types] = types
    idx[..., F - 2] = ents
    idx[..., F - 1] = rels
    labels = (rng.random(n_pairs) < 0.5).astype(np.float32)
    return idx, labels


def draw_num_paths(rng, n_pairs, pmax=28):
    """P = min(1 + Geom(0.57), pmax): mean ~1.75 as on the shipped sample."""
    return np.minimum(rng.geometric(0.57, size=n_pairs), pmax).astype(np.int32)


def make_bucketed(n_paths_target, T, F=3, Vt=6, Ve=10000, Vr=9, num_types=1, seed=SEED, pmax=28):
    """A whole synthetic path set bucketed by #paths like movie_data_format.py:301-314:
    returns {P: (idx[n_P,P,T,F], labels[n_P])} whose total path count is ~ n_paths_target."""
    rng = np.random.default_rng(seed)
    n_pairs = max(1, int(round(n_paths_target / 1.75)))
    Ps = draw_num_paths(rng, n_pairs, pmax)
    out = {}
    for P in np.unique(Ps):
        cnt = int((Ps == P).sum())
        out[int(P)] = make_paths(cnt, int(P), T, F, Vt, Ve, Vr, num_types, seed=seed + 1000 + int(P))
    return out
