"""Independent second implementation of the songPathRnn graph on PyTorch-CPU autograd.

Used ONLY to cross-check oracle/kprn_oracle.c (SURVEY.md section 8c mitigation (i)): torch's
own LSTM kernel with gates permuted FastLSTM [i,g,f,o] -> torch [i,f,g,o], single bias.
"""
import numpy as np
import torch


def split_params(orc, theta):
    lay = orc.layout()
    out = {}
    for nm, (off, shp) in lay.items():
        n = int(np.prod(shp))
        out[nm] = torch.tensor(np.asarray(theta[off:off + n]).reshape(shp), dtype=torch.float64, requires_grad=True)
    return out


def _perm(w, H):
    # rows [i,g,f,o] -> [i,f,g,o]
    return torch.cat([w[0:H], w[2 * H:3 * H], w[H:2 * H], w[3 * H:4 * H]], dim=0)


def forward(orc, prm, idx, reducer=2, K=5):
    c = orc.cfg
    idx = torch.as_tensor(np.asarray(idx), dtype=torch.long)
    B, P, T, F = idx.shape
    flat = idx.view(B * P, T, F) - 1
    nT = c.numTypes
    typ = sum(prm["type_emb"][flat[:, :, F - nT - 2 + k]] for k in range(nT))
    ent = prm["entity_emb"][flat[:, :, F - 2]]
    rel = prm["relation_emb"][flat[:, :, F - 1]]
    x = torch.cat([typ, ent, rel], dim=2)  # [N,T,D]
    H = c.H
    h_in = x
    for l in range(c.L):
        lstm = torch.nn.LSTM(h_in.shape[2], H, num_layers=1, batch_first=True).double()
        Wi, bi, Wo = prm[f"lstm{l + 1}.i2g.weight"], prm[f"lstm{l + 1}.i2g.bias"], prm[f"lstm{l + 1}.o2g.weight"]
        # functional call so autograd reaches our leaves
        params = {"weight_ih_l0": _perm(Wi, H), "weight_hh_l0": _perm(Wo, H),
                  "bias_ih_l0": _perm(bi, H), "bias_hh_l0": torch.zeros(4 * H, dtype=torch.float64)}
        h_in, _ = torch.func.functional_call(lstm, params, (h_in,))
    hT = h_in[:, -1, :]
    s = hT @ prm["out.weight"].t() + prm["out.bias"]  # [N,C]
    s3 = s.view(B, P, c.C)
    if reducer == 2:
        y = torch.logsumexp(s3, dim=1)
    elif reducer == 0:
        y = s3.max(dim=1).values
    else:
        kk = min(K, P)
        y = s3.topk(kk, dim=1).values.mean(dim=1)
    return x, s, y, torch.sigmoid(y)


def loss_and_grads(orc, theta, idx, labels, class_id=1, reducer=2, K=5):
    prm = split_params(orc, theta)
    x, s, y, p = forward(orc, prm, idx, reducer, K)
    pc = p[:, class_id - 1]
    t = torch.as_tensor(np.asarray(labels), dtype=torch.float64)
    eps = 1e-12
    loss = -(t * torch.log(pc + eps) + (1 - t) * torch.log(1 - pc + eps)).mean()
    loss.backward()
    g = np.zeros(orc.n)
    for nm, (off, shp) in orc.layout().items():
        n = int(np.prod(shp))
        gr = prm[nm].grad
        g[off:off + n] = 0 if gr is None else gr.numpy().ravel()
    return float(loss.detach()), g, s.detach().numpy(), p.detach().numpy()
