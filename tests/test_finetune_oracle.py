"""CPU: the torch restatement of the fine-tune head (oracle/finetune_torch.py) against the fixture produced by running the
reference class itself under import stand-ins (tests/golden/make_golden_finetune.py)."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load_golden(goal=False):
    from oracle import finetune_torch as O
    g = np.load(os.path.join(HERE, "golden", "finetune_tiny_goal.npz" if goal else "finetune_tiny.npz"))
    L, wv, wt, embed, hid, na = [int(v) for v in g["cfg"]]
    cfg = O.HeadConfig(layers=L, width_v=wv, width_t=wt, embed=embed, hidden=hid, n_actions=na, gamma=float(g["gamma"]),
                       logit_scale=float(g["logit_scale"]), goal_conditioned=goal)
    P = {k[6:]: g[k] for k in g.files if k.startswith("param:")}
    G = {k[5:]: g[k] for k in g.files if k.startswith("grad:")}
    batch = (g["img_inter"], g["img_final"], None if goal else g["txt_inter"], None if goal else g["txt_final"], g["r"], g["action"])
    return cfg, P, G, batch, g


def test_goal_conditioned_variant_matches_the_reference_class():
    """clip_multiscale_adapter.py:208-212,224-230 (goal_conditioned=True): fixture = the reference class itself with that switch on
    (make_golden_finetune.py --goal).  Loss, both components, every gradient; the text head's parameters get NO gradient."""
    from oracle import finetune_torch as O
    cfg, P, G, batch, g = load_golden(goal=True)
    assert int(g["goal_conditioned"]) == 1 and batch[0].shape[0] == 4
    got, aux = O.grads(P, cfg, batch)
    assert abs(aux["loss"] - float(g["loss"])) < 2e-5 and abs(aux["vip_loss"] - float(g["vip_loss"])) < 2e-5
    assert abs(float(P["lambda_id"]) * aux["id_loss"] - float(g["lambda_id_times_id_loss"])) < 2e-5
    for k in G:
        scale = max(np.abs(G[k]).max(), 1e-6)
        assert np.abs(got[k] - G[k]).max() / scale < 2e-4, k
    text_side = {k for k in P if k.startswith("text_")}
    assert text_side <= set(aux["no_grad"]) and all(not np.any(G[k]) for k in text_side)


def test_param_tree_matches_reference_state_dict():
    from oracle import finetune_torch as O
    cfg, P, _, _, _ = load_golden()
    shapes = O.param_shapes(cfg)
    assert set(shapes) == set(P)
    for k, s in shapes.items():
        assert tuple(P[k].shape) == tuple(s), k


def test_loss_and_components_match_reference():
    from oracle import finetune_torch as O
    cfg, P, _, batch, g = load_golden()
    _, aux = O.grads(P, cfg, batch)
    assert abs(aux["loss"] - float(g["loss"])) < 2e-5
    assert abs(aux["vip_loss"] - float(g["vip_loss"])) < 2e-5
    assert abs(float(P["lambda_id"]) * aux["id_loss"] - float(g["lambda_id_times_id_loss"])) < 2e-5


def test_gradients_match_reference_autograd():
    from oracle import finetune_torch as O
    cfg, P, G, batch, _ = load_golden()
    got, _ = O.grads(P, cfg, batch)
    for k in G:
        scale = max(np.abs(G[k]).max(), 1e-6)
        assert np.abs(got[k] - G[k]).max() / scale < 2e-4, k  # the reference ran in float32


def test_vip_exponent_broadcasts_to_a_matrix():
    """r is [B,1] and the scores [B]: the reference's mean runs over B*B terms (clip_multiscale_adapter.py:216-220)."""
    from oracle import finetune_torch as O
    cfg, P, _, batch, _ = load_golden()
    out = O.forward(O.to_torch(P), cfg, *[torch.as_tensor(b, dtype=torch.float64) for b in batch[:5]], torch.as_tensor(batch[5]))
    s = out["scores"].numpy()
    r = batch[4].astype(np.float64) - 1.0
    mat = np.exp(-(r[:, None] + cfg.gamma * s[2][None, :] - s[1][None, :]))
    vip = (1 - cfg.gamma) * -s[0].mean() + np.log(1e-8 + mat.mean())
    assert abs(vip - float(out["vip_loss"])) < 1e-10
    per_sample = np.exp(-(r + cfg.gamma * s[2] - s[1])).mean()
    assert abs(np.log(1e-8 + per_sample) - np.log(1e-8 + mat.mean())) > 1e-6  # a per-sample reading would differ


@pytest.mark.parametrize("use_id", [True, False])
def test_adamw_matches_torch(use_id):
    """incl. use_id_loss off: the inverse model and lambda_id then have .grad None and torch.optim.AdamW leaves them untouched
    (no decoupled decay either) -- ADVICE r1."""
    import dataclasses
    from oracle import finetune_torch as O
    cfg, P, _, batch, _ = load_golden()
    cfg = dataclasses.replace(cfg, use_id=use_id)
    Pt = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in P.items()}
    opt = torch.optim.AdamW(list(Pt.values()), lr=1e-3, weight_decay=0.01)
    for _ in range(3):
        opt.zero_grad()
        out = O.forward(Pt, cfg, *[torch.as_tensor(b, dtype=torch.float64) for b in batch[:5]], torch.as_tensor(batch[5]))
        out["loss"].backward()
        opt.step()
    got, _ = O.train_steps(P, cfg, [batch], 1e-3, 0.01, 3)
    for k in P:
        assert np.abs(got[k] - Pt[k].detach().numpy()).max() < 1e-9, k
    if not use_id:
        assert all(np.array_equal(got[k], np.asarray(P[k], np.float64)) for k in P if k.startswith("inverse_layer.") or k == "lambda_id")
