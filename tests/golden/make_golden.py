"""Regenerates the committed fixtures under tests/golden/.  Run in the build container
(needs /root/reference for the eval-chain slices; the oracle vectors need only oracle/).

  python tests/golden/make_golden.py

1. c1_small.npz / c2_small.npz -- inputs, weights and float64 outputs of the CPU oracle (forward,
   gradients, 3 Adam steps) on the plumbing config C1 (T=3, d=16, L=1) and a C2-shaped config
   (T=6, D=H=64, L=2).  The oracle itself is pinned against PyTorch autograd + finite differences
   (tests/test_oracle.py); the reference (Lua/Torch7) cannot run here -- parity unpinned.
2. eval_chain/ -- the first 300 lines of the reference's own aligned fixtures
   release/songPathRnn/eval/config1/{test_sample.res,test_combine_sample.txt},
   release/songPathRnn/data/output/test_sample.list.entity and the first user block of
   test_combine_sorted_sample.txt: known answers for combine_result.py / resort.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from kprn_amd import synth  # noqa: E402
from oracle.oracle import Oracle, make_cfg, make_opt  # noqa: E402


def oracle_case(name, Ve, dt, de, dr, H, L, T, pairs, P, seed):
    cfg = make_cfg(Vt=6, Ve=Ve, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L)
    o = Oracle(cfg, np.float64)
    theta = o.init_params(seed, 0.1).astype(np.float32).astype(np.float64)
    idx, labels = synth.make_paths(pairs, P, T, Ve=Ve, seed=seed + 1)
    ps, pooled, probs = o.forward(theta, idx)
    loss, grad, p = o.forward_backward(theta, idx, labels)
    th = theta.copy()
    st = o.new_state()
    losses = [o.train_step(th, st, make_opt(method=1, lr=1e-2), idx, labels)[0] for _ in range(3)]
    np.savez_compressed(os.path.join(HERE, name), cfg=np.array([6, Ve, 9, dt, de, dr, 3, 1, H, L, 46, 2, 5], np.int32),
                        theta=theta.astype(np.float32), idx=idx, labels=labels, path_scores=ps, pooled=pooled,
                        probs=probs, loss=np.float64(loss), grad=grad, theta_after3=th, losses3=np.array(losses))


def rnn_case(name, Ve, dt, de, dr, H, L, T, pairs, P, seed, use_relu):
    """rnnType rnn (the shipped config.sh default): Recurrence + MaskZero, pads masked"""
    cfg = make_cfg(Vt=6, Ve=Ve, Vr=9, dt=dt, de=de, dr=dr, H=H, L=L, rnn_type=1, use_relu=use_relu)
    o = Oracle(cfg, np.float64)
    theta = o.init_params(seed, 0.2).astype(np.float32).astype(np.float64)
    o.zero_pad(theta)
    idx, labels = synth.make_paths(pairs, P, T, Ve=Ve, seed=seed + 1)
    ps, pooled, probs = o.forward(theta, idx)
    loss, grad, p = o.forward_backward(theta, idx, labels)
    np.savez_compressed(os.path.join(HERE, name), cfg=np.array([6, Ve, 9, dt, de, dr, 3, 1, H, L, 46, 2, 5, 1, use_relu], np.int32),
                        theta=theta.astype(np.float32), idx=idx, labels=labels, path_scores=ps, pooled=pooled,
                        probs=probs, loss=np.float64(loss), grad=grad)


def eval_slices():
    ref = "/root/reference/release/songPathRnn"
    out = os.path.join(HERE, "eval_chain")
    os.makedirs(out, exist_ok=True)
    for src, dst in ((f"{ref}/eval/config1/test_sample.res", "test_sample.res"),
                     (f"{ref}/data/output/test_sample.list.entity", "test_sample.list.entity"),
                     (f"{ref}/eval/config1/test_combine_sample.txt", "test_combine_sample.txt")):
        with open(src) as f, open(os.path.join(out, dst), "w") as g:
            for i, line in enumerate(f):
                if i >= 300:
                    break
                g.write(line)
    # one complete user block of the sorted file (sorted by (int(user), -score), resort.py:38)
    with open(f"{ref}/eval/config1/test_combine_sorted_sample.txt") as f, open(os.path.join(out, "sorted_first_users.txt"), "w") as g:
        users, cnt = [], {}
        for line in f:
            u = line.split("\t")[0]
            if u not in users:
                if len(users) == 2:
                    break
                users.append(u)
            cnt[u] = cnt.get(u, 0) + 1
            if cnt[u] <= 250:  # the first 250 rows of each of the first two users
                g.write(line)


if __name__ == "__main__":
    oracle_case("c1_small.npz", 200, 4, 8, 4, 16, 1, 3, 24, 2, 11)
    oracle_case("c2_small.npz", 400, 16, 32, 16, 64, 2, 6, 16, 3, 21)
    rnn_case("c1_rnn_small.npz", 200, 8, 16, 8, 32, 2, 6, 20, 3, 31, 1)
    if os.path.isdir("/root/reference"):
        eval_slices()
    print(sorted(os.listdir(HERE)))
