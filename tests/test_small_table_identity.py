"""not gpu: the small-table identity of DESIGN.md 3.4d (lstm_bf16.hip k_onehot_T / k_small_tables_finish, kernels_basic.hip k_onehot_cols /
k_small_tables_finish_f32), replayed in numpy.

With x[t, n] = [ Wt[type] | We[entity] | Wr[relation] ] (FeatureEmbedding.lua:112-121) and dA the gradient of the layer-0 pre-activations
a = x W_i2g^T + ..., the reference forms dx = dA W_i2g and scatter-adds its type / relation thirds into the tables, and dW_i2g = dA^T x over the full
width.  The engine forms ONE extra column block G = dA^T [S_r | S_t] of the dW product (S = one-hot selector columns, relation rows first, then Vr +
type) and finishes from it:
    dW_i2g[:, relation columns] = G_r Wr         dWr = G_r^T W_i2g[:, relation columns]
    dW_i2g[:, type columns]     = G_t Wt         dWt = G_t^T W_i2g[:, type columns]
so that dx is needed for the entity slice only.  Both sides in float64 here: the identity is exact algebra, the test pins the block layout."""
import numpy as np


def test_table_and_weight_gradients_from_one_hot_columns_equal_the_dx_route():
    rng = np.random.default_rng(5)
    T, N, H = 5, 37, 6
    Vt, Vr, Ve, dt, de, dr = 6, 9, 50, 4, 8, 4
    D = dt + de + dr
    Wt, We, Wr = rng.normal(size=(Vt, dt)), rng.normal(size=(Ve, de)), rng.normal(size=(Vr, dr))
    Wi = rng.normal(size=(4 * H, D))
    typ, ent, rel = rng.integers(0, Vt, (T, N)), rng.integers(0, Ve, (T, N)), rng.integers(0, Vr, (T, N))
    dA = rng.normal(size=(T, N, 4 * H))
    x = np.concatenate([Wt[typ], We[ent], Wr[rel]], axis=2)                      # [T, N, D]
    # the dx route (what the reference's modules do: LookupTable:accGradParameters scatter-adds rows of dx)
    dx = dA @ Wi                                                                 # [T, N, D]
    gWt, gWe, gWr = np.zeros_like(Wt), np.zeros_like(We), np.zeros_like(Wr)
    np.add.at(gWt, typ, dx[:, :, :dt])
    np.add.at(gWe, ent, dx[:, :, dt:dt + de])
    np.add.at(gWr, rel, dx[:, :, dt + de:])
    dWi = np.einsum("tng,tnd->gd", dA, x)                                        # [4H, D]
    # the engine's route: selector columns [S_r | S_t] appended to the dW product's other operand; relation rows first, type rows at Vr + type
    S = np.zeros((T, N, 128))
    S[np.arange(T)[:, None], np.arange(N)[None, :], rel] = 1.0
    S[np.arange(T)[:, None], np.arange(N)[None, :], Vr + typ] += 1.0
    Z = np.concatenate([x[:, :, dt:dt + de], S], axis=2)                         # [x_e | S]: the merged product's column blocks (h_{t-1} omitted here)
    Ct = np.einsum("tng,tnz->gz", dA, Z)                                         # [4H, de + 128]
    G_r, G_t = Ct[:, de:de + Vr], Ct[:, de + Vr:de + Vr + Vt]
    assert np.all(Ct[:, de + Vr + Vt:] == 0.0)                                   # (unused selector columns stay empty)
    dWi_e = Ct[:, :de]
    np.testing.assert_allclose(dWi_e, dWi[:, dt:dt + de], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(G_r @ Wr, dWi[:, dt + de:], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(G_t @ Wt, dWi[:, :dt], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(G_r.T @ Wi[:, dt + de:], gWr, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(G_t.T @ Wi[:, :dt], gWt, rtol=1e-12, atol=1e-12)
    # ... and the only part of dx the engine still forms is the entity slice
    gWe2 = np.zeros_like(We)
    np.add.at(gWe2, ent, dA @ Wi[:, dt:dt + de])
    np.testing.assert_allclose(gWe2, gWe, rtol=1e-12, atol=1e-12)
