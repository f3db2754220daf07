"""-m gpu: the time-split tile hand-over of the fused D = H = 64 BPTT launches (kprn_amd/csrc/lstm_fused_common.h ho_plan, lstm_fused_bwd.hip; option
"tile_handover" = 2 (default: workgroup b paired with b + G / 2) | 1 (b with G - 1 - b) | 0 (whole tiles only)).

The persistent kernels deal 64-path tiles round-robin and a left-padded path set gives tiles of T and T - 2 executed steps: the workgroups' step sums
differ by whole steps and the launch lasts as long as the heaviest.  With the option on, the two workgroups of a pair move the first d steps (in the
backward's own time order: T-1 .. T-d) of the heavy one's first tile between them, the tile's recurrent state (dh, dc) going through a slot in global
memory.  What must hold:
  * the forward does not take part: every path score bit-identical with the option on and off;
  * the backward changes only which workgroup's partial sum a row's weight-gradient contribution lands in: dx / entity gradients per row identical,
    weight gradients equal to fp32 re-association;
  * against the f64 oracle inside the usual bars (model/OneModel.lua:236,268-275; optimizer/MyOptimizer.lua:177-221);
  * a batch whose tiles are all alike and divide the grid evenly moves nothing."""
import numpy as np
import pytest

from kprn_amd import _ffi, synth
from oracle.oracle import Oracle, make_cfg, make_opt

pytestmark = pytest.mark.gpu
SHAPE = dict(Vt=6, Ve=30000, Vr=9, dt=16, de=32, dr=16, H=64, L=2)
T = 6


def rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b)) / max(1e-30, np.max(np.abs(b))))


def mk(seed=3, handover=2, plan=True, L=2):
    shape = dict(SHAPE, L=L)
    eng = _ffi.Engine(shape["Vt"], shape["Ve"], shape["Vr"], shape["dt"], shape["de"], shape["dr"], shape["H"], shape["L"])
    eng.set_option("small_tiles", "0")
    eng.set_option("prefix_plan", "1" if plan else "0")
    eng.set_option("tile_handover", str(int(handover)))
    o64 = Oracle(make_cfg(**shape), np.float64)
    theta = o64.init_params(seed, 0.1).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    return eng, o64, theta


def run(eng, idx, labels):
    b = eng.batch(idx, labels)
    out = eng.forward(b, 1, want=("path_scores", "probs"))
    loss = eng.backward(b, 1)
    return b, out["path_scores"].copy(), out["probs"].copy(), loss, eng.get_flat_grads().copy()


# (pairs, P, plan, real_len): 300 / 500 whole tiles of 6 steps on 256 workgroups (44 / 244 heavy ones, whole-tile differences: d = 3),
# the bench's 4 / 6-step mix behind the identical-prefix plan, a ragged last tile, L = 1
CASES = [(19200, 1, False, 6, 2), (16000, 2, False, None, 2), (65536 // 4, 4, True, None, 2), (21001, 1, True, None, 2), (300 * 64 - 7, 1, True, None, 1)]


@pytest.mark.parametrize("pairs,P,plan,real_len,L", CASES)
def test_hand_over_on_equals_off(pairs, P, plan, real_len, L):
    idx, labels = synth.make_paths(pairs, P, T, Ve=SHAPE["Ve"], seed=pairs % 1000 + P, real_len=real_len)
    res = {}
    for on in (2, 1, 0):
        eng, _, _ = mk(handover=on, plan=plan, L=L)
        b, ps, probs, loss, g = run(eng, idx, labels)
        st = b.handover_stats
        if on:
            # tiles do change hands, and the longest workgroup gets shorter (a pairing reaches every heavy workgroup only while at most half of them
            # are heavy: the 500-tile case has 244 heavy ones of 256 -- 12 pairs form, the longest workgroup stays)
            assert st[0] > 0 and st[1] >= st[0] and st[3] <= st[2] and (st[3] < st[2] or pairs * P == 32000), st
        else:
            assert st[0] == 0 and st[2] == st[3], st
        # a second pass + backward over the same batch: the slots' epochs move on, nothing stale is picked up
        b2, ps2, probs2, loss2, g2 = run(eng, idx, labels)
        assert np.array_equal(ps, ps2) and loss == loss2
        res[on] = (ps, probs, loss, g, eng.layout())
        eng.close()
    for on in (2, 1):
        assert np.array_equal(res[on][0], res[0][0])        # forward: bit-identical scores
        assert np.array_equal(res[on][1], res[0][1])
        assert res[on][2] == res[0][2]
        g1, g0, lay = res[on][3].astype(np.float64), res[0][3].astype(np.float64), res[on][4]
        for nm, (off, shp) in lay.items():
            n = int(np.prod(shp))
            # a row's dx is the same arithmetic whichever workgroup runs it; the weight gradients are other partial sums, and the entity rows of hub
            # entities are summed with atomics across the gather-reduce's segments in any order (run to run, with or without the hand-over)
            assert rel_inf(g1[off:off + n], g0[off:off + n]) <= 1e-5, (on, nm)


@pytest.mark.parametrize("pairs,P,plan", [(5000, 4, True), (19200, 1, False)])
def test_hand_over_against_the_f64_oracle(pairs, P, plan):
    eng, o64, theta = mk(handover=2, plan=plan)
    idx, labels = synth.make_paths(pairs, P, T, Ve=SHAPE["Ve"], seed=77, real_len=None if plan else 6)
    b, ps, probs, loss, g = run(eng, idx, labels)
    assert b.handover_stats[0] > 0
    ops, _, oprobs = o64.forward(theta, idx)
    assert rel_inf(ps, ops) < 3e-6
    np.testing.assert_allclose(probs, oprobs[:, 0], rtol=1e-5)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1.0, abs(ol))
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < 2e-4, nm
    eng.close()


def test_training_steps_with_hand_over_match_whole_tiles():
    """six Adam steps + a scoring pass on the side stream beside each training forward (the bench's step): parameters equal to the whole-tile engine's
    to fp32 re-association of the weight gradients"""
    batches = [synth.make_paths(3000 + 500 * k, 4, T, Ve=SHAPE["Ve"], seed=900 + k) for k in range(3)]
    finals = []
    for on in (2, 0):
        eng, _, _ = mk(handover=on)
        eng.set_option("score_overlap", "1")
        opt = _ffi.make_opt(method=1, lr=1e-3)
        bs = [eng.batch(i, l) for i, l in batches]
        probs = []
        for k in range(6):
            b = bs[k % 3]
            eng.forward_async(b, 1)
            eng.train_step(b, opt)
            probs.append(eng.read_probs(b.B).copy())
        finals.append((eng.get_flat_params().astype(np.float64), probs))
        eng.close()
    assert float(np.max(np.abs(finals[0][0] - finals[1][0]))) < 2e-6
    for p1, p0 in zip(finals[0][1], finals[1][1]):
        np.testing.assert_allclose(p1, p0, rtol=2e-5)


def test_nothing_moves_when_the_tiles_divide_evenly():
    eng, _, _ = mk(handover=2, plan=False)
    idx, labels = synth.make_paths(512 * 64, 1, T, Ve=SHAPE["Ve"], seed=5, real_len=6)
    st = eng.batch(idx, labels).handover_stats
    assert st[0] == 0 and st[2] == st[3] == 2 * (2 * T - 1), st
    eng.close()


def test_one_layer_more_tiles_than_workgroups():
    """found by this file's L = 1 case in round 6: with ONE layer a slot of the fused forward has a single barrier, and a workgroup's second tile let wave 3
    (which has no share of the 46-class head) run a whole slot ahead of the others and read h_{t-1} rows they had not written -- scores off by 3e-4, run to run,
    in batches of more than 256 tiles.  Against the f64 oracle, twice (model/OneModel.lua:236,268-275)."""
    eng, o64, theta = mk(handover=2, plan=True, L=1)
    idx, labels = synth.make_paths(270 * 64 + 5, 1, T, Ve=SHAPE["Ve"], seed=31)
    ops, _, _ = o64.forward(theta, idx)
    for rep in range(2):
        b = eng.batch(idx, labels)
        ps = eng.forward(b, 1, want=("path_scores",))["path_scores"]
        assert rel_inf(ps, ops) < 3e-6, rep
    eng.set_option("prefix_plan", "0")
    ps0 = eng.forward(eng.batch(idx, labels), 1, want=("path_scores",))["path_scores"]
    assert rel_inf(ps0, ops) < 3e-6
    eng.close()


@pytest.mark.parametrize("Tt,real_len,plan", [(2, 2, False), (6, 2, True), (3, 2, True)])
def test_two_step_tiles_more_tiles_than_workgroups(Tt, real_len, plan):
    """round 6: a tile's ids reach LDS by DMA, requested in the first slot of the tile BEFORE and read at the top of that tile's last slot (forward;
    lstm_fused_common.h ids_stage_dma), and by the tile before's start in the bottom BPTT launch.  With two-step tiles the last slot is the very next one:
    the only case in which the waves must meet once more in between.  T = 2, and longer paths whose common left padding leaves two executed steps
    (tile_k = T - 2), 300 tiles on 256 workgroups, twice; scores and every gradient against the f64 oracle (model/OneModel.lua:236,268-275)."""
    eng, o64, theta = mk(handover=2, plan=plan)
    idx, labels = synth.make_paths(300 * 64 - 3, 1, Tt, Ve=SHAPE["Ve"], seed=41 + Tt, real_len=real_len)
    ops, _, _ = o64.forward(theta, idx)
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    for rep in range(2):
        b, ps, probs, loss, g = run(eng, idx, labels)
        assert rel_inf(ps, ops) < 3e-6, rep
        assert abs(loss - ol) < 1e-5 * max(1.0, abs(ol))
        for nm, (off, shp) in eng.layout().items():
            n = int(np.prod(shp))
            assert rel_inf(g[off:off + n], og[off:off + n]) < 2e-4, (nm, rep)
    eng.close()


@pytest.mark.parametrize("Tt,F,nT,L", [(16, 3, 1, 2), (7, 6, 2, 2), (5, 4, 2, 1)])
def test_dma_id_planes_long_paths_and_two_type_slots(Tt, F, nT, L):
    """the LDS-DMA id staging at its edges: T = MAXT_LDS (the planes are full), two type slots per step (the CAddTable branch of the gather reads the batch's
    own id tensor beside the planes: net/FeatureEmbedding.lua:55), unused leading feature columns (F = 6), one layer; 290 tiles on 256 workgroups (second tiles
    take the staged-ahead path), forward and every gradient against the f64 oracle."""
    shape = dict(SHAPE, L=L)
    eng = _ffi.Engine(shape["Vt"], shape["Ve"], shape["Vr"], shape["dt"], shape["de"], shape["dr"], shape["H"], L, F=F, num_types=nT)
    eng.set_option("small_tiles", "0")
    o64 = Oracle(make_cfg(F=F, numTypes=nT, **shape), np.float64)
    theta = o64.init_params(5, 0.1).astype(np.float32).astype(np.float64)
    eng.set_flat_params(theta.astype(np.float32))
    idx, labels = synth.make_paths(290 * 64 - 9, 1, Tt, F=F, Ve=SHAPE["Ve"], num_types=nT, seed=60 + Tt)
    b, ps, probs, loss, g = run(eng, idx, labels)
    ops, _, _ = o64.forward(theta, idx)
    assert rel_inf(ps, ops) < 3e-6
    ol, og, _ = o64.forward_backward(theta, idx, labels)
    assert abs(loss - ol) < 1e-5 * max(1.0, abs(ol))
    for nm, (off, shp) in eng.layout().items():
        n = int(np.prod(shp))
        assert rel_inf(g[off:off + n], og[off:off + n]) < 2e-4, nm
    eng.close()
