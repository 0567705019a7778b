"""Oracle (test infrastructure): the ARP-DT policy forward / loss / train step, restated in torch
(autograd supplies the backward that jax.value_and_grad supplies in the reference).

Follows, for the shipped configuration (transfer_type=m3ae_*, use_adapter=True, use_text=False,
use_discrete_action=True, one image key):

  * adapter MLP, ReLU on BOTH layers .............. /root/reference/arp_dt/models/adapter/layers.py:6-30
  * y = sigmoid(residual_weight) * adapter(x) + (1 - sigmoid(.)) * x   arp_dt/ARPDT.py:466-472
  * reshape [B,T,tokens*dim] -> Dense(emb) -> tanh  arp_dt/ARPDT.py:475-484 (declared :141)
  * action Embed, rtg Dense(1 -> emb, no bias) .... arp_dt/ARPDT.py:278-293 (declared :102-109)
  * token order [image, rtg, action], causal mask . arp_dt/ARPDT.py:159-200
  * Transformer / Block / Attention / FeedForward . arp_dt/layers.py:11-166 (pre-LN, flax LayerNorm eps
    1e-6, qkv+out Dense WITH bias, FFN WITHOUT bias, nn.gelu = tanh approximation, scale hd^-0.5,
    masked fill finfo(f32).min)
  * heads: action from the rtg-token rows 1::3, return from the image-token rows 0::3;
    Dense(emb)+ReLU+Dense(n, no bias); the "5-member ensemble" is one module five times .. ARPDT.py:94-99,203-222
  * losses: CE averaged over ALL B*T*n_actions elements, acc, MSE ................ ARPDT.py:238-261,498-507
  * L2 penalty weight_decay*0.5*sum ||p||^2 over params with ndim > 1; aux keys .. main_procgen.py:105-126
  * pmean of grads; optax clip_by_global_norm(c) -> adamw(b1 .9, b2 .999, eps 1e-8) whose decay mask is
    all-False (no_decay_list is empty -> decay applies to nothing) ............... main_procgen.py:128-139,490-507

Parameters are keyed by their Flax tree path flattened with '/' (SURVEY.md Appendix C); kernels are
[in, out] as in Flax.  Parity status: UNPINNED by the reference (flax/jax are absent and the reference has
no tests); pinned against fp64 finite differences and an independent numpy forward in tests/test_oracle_policy.py.
"""
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class PolicyConfig:
    emb: int = 128
    depth: int = 2
    heads: int = 8
    mlp_ratio: int = 4
    n_actions: int = 15
    window: int = 4
    enc_tokens: int = 257
    enc_dim: int = 768
    use_adapter: bool = True
    lambda_ret: float = 1.0
    weight_decay: float = 5e-5
    clip_norm: float = 10.0
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8
    alibi_bias: bool = False  # config.alibi_bias (ARPDT.py:88 -> layers.py:74-78); off in the shipped configuration


def param_shapes(cfg):
    """Flax tree of the shipped policy, '/'-flattened (Appendix C)."""
    E, D, H = cfg.emb, cfg.enc_dim, cfg.mlp_ratio * cfg.emb
    s = {}
    if cfg.use_adapter:
        for i in (0, 1):
            s[f"AdapterMLP_0/Dense_{i}/kernel"] = (D, D)
            s[f"AdapterMLP_0/Dense_{i}/bias"] = (D,)
        s["residual_weight"] = (1,)
    s["image_text_input/kernel"] = (cfg.enc_tokens * D, E)
    s["image_text_input/bias"] = (E,)
    s["action_input/embedding"] = (cfg.n_actions, E)
    s["rtg_input/kernel"] = (1, E)
    for i in range(cfg.depth):
        p = f"policy/Block_{i}/"
        for ln in ("LayerNorm_0", "LayerNorm_1"):
            s[p + ln + "/scale"] = (E,)
            s[p + ln + "/bias"] = (E,)
        s[p + "Attention_0/Dense_0/kernel"] = (E, 3 * E)
        s[p + "Attention_0/Dense_0/bias"] = (3 * E,)
        s[p + "Attention_0/Dense_1/kernel"] = (E, E)
        s[p + "Attention_0/Dense_1/bias"] = (E,)
        s[p + "FeedForward_0/fc1/kernel"] = (E, H)
        s[p + "FeedForward_0/fc2/kernel"] = (H, E)
    s["policy/LayerNorm_0/scale"] = (E,)
    s["policy/LayerNorm_0/bias"] = (E,)
    for head, n in (("action_outputs_0", cfg.n_actions), ("return_outputs_0", 1)):
        s[head + "/layers_0/kernel"] = (E, E)
        s[head + "/layers_0/bias"] = (E,)
        s[head + "/layers_2/kernel"] = (E, n)
    return s


def num_params(cfg):
    return int(sum(np.prod(v) for v in param_shapes(cfg).values()))


def alibi_slopes(n):
    """arp_dt/layers.py:97-110 (_get_attention_slopes): 2^(-8 (i + 1) / n) for a power-of-two head count, otherwise the slopes of the next lower power
    of two followed by every other slope of the next higher one."""
    import math

    def pow2(m):
        start = 2 ** (-(2 ** -(math.log2(m) - 3)))
        return [start * start ** i for i in range(m)]

    if math.log2(n).is_integer():
        return pow2(n)
    c = 2 ** math.floor(math.log2(n))
    return pow2(c) + alibi_slopes(2 * c)[0::2][: n - c]


def forward(P, cfg, enc, action, rtg):
    """P: dict name -> tensor.  enc [B,T,tokens,dim], action int64 [B,T], rtg [B,T,1].
    Returns dict(action_pred [B,T,n_actions], return_pred [B,T,1], loss, acc, trans_loss, return_loss)."""
    B, T = action.shape
    E = cfg.emb
    x = enc.reshape(B * T * cfg.enc_tokens, cfg.enc_dim)
    if cfg.use_adapter:
        a = F.relu(x @ P["AdapterMLP_0/Dense_0/kernel"] + P["AdapterMLP_0/Dense_0/bias"])
        a = F.relu(a @ P["AdapterMLP_0/Dense_1/kernel"] + P["AdapterMLP_0/Dense_1/bias"])
        res = torch.sigmoid(P["residual_weight"])
        x = res * a + (1 - res) * x
    img = torch.tanh(x.reshape(B, T, -1) @ P["image_text_input/kernel"] + P["image_text_input/bias"])
    act = P["action_input/embedding"][action]
    rt = rtg @ P["rtg_input/kernel"]
    tok = torch.cat([img, rt, act], dim=-1).reshape(B, 3 * T, E)
    L = 3 * T
    mask = torch.tril(torch.ones(L, L, dtype=torch.bool, device=tok.device))
    hd = E // cfg.heads
    h = tok
    for i in range(cfg.depth):
        p = f"policy/Block_{i}/"
        y = F.layer_norm(h, (E,), P[p + "LayerNorm_0/scale"], P[p + "LayerNorm_0/bias"], 1e-6)
        qkv = y @ P[p + "Attention_0/Dense_0/kernel"] + P[p + "Attention_0/Dense_0/bias"]
        q, k, v = (t.reshape(B, L, cfg.heads, hd).transpose(1, 2) for t in qkv.split(E, dim=-1))
        att = (q @ k.transpose(-2, -1)) * hd ** -0.5
        if getattr(cfg, "alibi_bias", False):  # layers.py:74-78: slopes[:, None, None] * arange(n)[None, None, :] -- a bias on the KEY index
            sl = torch.tensor(alibi_slopes(cfg.heads), dtype=att.dtype, device=att.device)
            att = att + sl[None, :, None, None] * torch.arange(L, dtype=att.dtype, device=att.device)[None, None, None, :]
        att = att.masked_fill(~mask, torch.finfo(att.dtype).min).softmax(-1)
        y = (att @ v).transpose(1, 2).reshape(B, L, E)
        h = h + y @ P[p + "Attention_0/Dense_1/kernel"] + P[p + "Attention_0/Dense_1/bias"]
        y = F.layer_norm(h, (E,), P[p + "LayerNorm_1/scale"], P[p + "LayerNorm_1/bias"], 1e-6)
        y = F.gelu(y @ P[p + "FeedForward_0/fc1/kernel"], approximate="tanh") @ P[p + "FeedForward_0/fc2/kernel"]
        h = h + y
    h = F.layer_norm(h, (E,), P["policy/LayerNorm_0/scale"], P["policy/LayerNorm_0/bias"], 1e-6)
    a_in, r_in = h[:, 1::3], h[:, 0::3]  # num_obs_token = 1, 3 tokens per step
    head = lambda z, n: F.relu(z @ P[n + "/layers_0/kernel"] + P[n + "/layers_0/bias"]) @ P[n + "/layers_2/kernel"]
    logits = head(a_in, "action_outputs_0")
    ret = head(r_in, "return_outputs_0")
    onehot = F.one_hot(action, cfg.n_actions).to(logits.dtype)
    trans = (-onehot * F.log_softmax(logits, -1)).mean()
    acc = (logits.argmax(-1) == action).to(logits.dtype).mean()
    rloss = ((ret - rtg) ** 2).mean()
    return dict(action_pred=logits, return_pred=ret, loss=trans + cfg.lambda_ret * rloss, acc=acc, trans_loss=trans,
                return_loss=rloss)


def loss_and_aux(P, cfg, enc, action, rtg):
    """loss_fn of create_train_step (main_procgen.py:105-126)."""
    out = forward(P, cfg, enc, action, rtg)
    l2 = sum((p ** 2).sum() for p in P.values() if p.ndim > 1)
    pen = cfg.weight_decay * 0.5 * l2
    loss = out["loss"] + pen
    aux = dict(loss=loss, acc=out["acc"] * 100, trans_loss=out["trans_loss"], return_loss=out["return_loss"],
               weight_penalty=pen, weight_l2=l2)
    return loss, aux, out


def grads(P, cfg, enc, action, rtg):
    Pr = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    loss, aux, out = loss_and_aux(Pr, cfg, enc, action, rtg)
    loss.backward()
    g = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in Pr.items()}
    return g, {k: float(v.detach()) for k, v in aux.items()}, {k: v.detach() for k, v in out.items()}


def init_state(P):
    return dict(params={k: v.clone() for k, v in P.items()}, mu={k: torch.zeros_like(v) for k, v in P.items()},
                nu={k: torch.zeros_like(v) for k, v in P.items()}, step=0)


def train_step(state, cfg, shards, lr_fn):
    """One reference train step on a list of per-device batches (pmap shards): pmean of (loss, aux,
    grads) over devices, clip_by_global_norm, adam (decoupled decay masked off), main_procgen.py:128-139."""
    P = state["params"]
    gs, auxs = [], []
    for enc, action, rtg in shards:
        g, aux, _ = grads(P, cfg, enc, action, rtg)
        gs.append(g)
        auxs.append(aux)
    n = len(shards)
    g = {k: sum(gi[k] for gi in gs) / n for k in P}
    aux = {k: sum(a[k] for a in auxs) / n for k in auxs[0]}
    gnorm = torch.sqrt(sum((v ** 2).sum() for v in g.values()))
    if gnorm >= cfg.clip_norm:
        g = {k: v / gnorm * cfg.clip_norm for k, v in g.items()}
    step = state["step"]
    lr = float(lr_fn(step))
    t = step + 1
    new = dict(params={}, mu={}, nu={}, step=t)
    for k in P:
        mu = cfg.b1 * state["mu"][k] + (1 - cfg.b1) * g[k]
        nu = cfg.b2 * state["nu"][k] + (1 - cfg.b2) * g[k] ** 2
        mhat = mu / (1 - cfg.b1 ** t)
        nhat = nu / (1 - cfg.b2 ** t)
        new["params"][k] = P[k] - lr * mhat / (torch.sqrt(nhat) + cfg.eps)
        new["mu"][k], new["nu"][k] = mu, nu
    aux["train_state_step"] = step
    aux["learning_rate"] = lr
    aux["grad_norm"] = float(gnorm)
    return new, aux


def forward_numpy(Pn, cfg, enc, action, rtg):
    """Independent numpy forward (no torch ops) used to cross-check ``forward``."""
    B, T = action.shape
    E, hd = cfg.emb, cfg.emb // cfg.heads
    x = enc.reshape(B * T * cfg.enc_tokens, cfg.enc_dim).astype(np.float64)
    g = lambda k: np.asarray(Pn[k], np.float64)
    if cfg.use_adapter:
        a = np.maximum(x @ g("AdapterMLP_0/Dense_0/kernel") + g("AdapterMLP_0/Dense_0/bias"), 0)
        a = np.maximum(a @ g("AdapterMLP_0/Dense_1/kernel") + g("AdapterMLP_0/Dense_1/bias"), 0)
        res = 1 / (1 + np.exp(-g("residual_weight")))
        x = res * a + (1 - res) * x
    img = np.tanh(x.reshape(B, T, -1) @ g("image_text_input/kernel") + g("image_text_input/bias"))
    tok = np.concatenate([img, rtg.astype(np.float64) @ g("rtg_input/kernel"), g("action_input/embedding")[action]], -1)
    h = tok.reshape(B, 3 * T, E)
    L = 3 * T

    def ln(z, s, b):
        mu = z.mean(-1, keepdims=True)
        return (z - mu) / np.sqrt(((z - mu) ** 2).mean(-1, keepdims=True) + 1e-6) * s + b

    for i in range(cfg.depth):
        p = f"policy/Block_{i}/"
        y = ln(h, g(p + "LayerNorm_0/scale"), g(p + "LayerNorm_0/bias"))
        qkv = y @ g(p + "Attention_0/Dense_0/kernel") + g(p + "Attention_0/Dense_0/bias")
        q, k, v = (t.reshape(B, L, cfg.heads, hd).transpose(0, 2, 1, 3) for t in np.split(qkv, 3, -1))
        s = q @ k.transpose(0, 1, 3, 2) * hd ** -0.5
        s = np.where(np.tril(np.ones((L, L), bool)), s, -np.inf)
        s = np.exp(s - s.max(-1, keepdims=True))
        s /= s.sum(-1, keepdims=True)
        y = (s @ v).transpose(0, 2, 1, 3).reshape(B, L, E)
        h = h + y @ g(p + "Attention_0/Dense_1/kernel") + g(p + "Attention_0/Dense_1/bias")
        y = ln(h, g(p + "LayerNorm_1/scale"), g(p + "LayerNorm_1/bias")) @ g(p + "FeedForward_0/fc1/kernel")
        y = 0.5 * y * (1 + np.tanh(np.sqrt(2 / np.pi) * (y + 0.044715 * y ** 3)))
        h = h + y @ g(p + "FeedForward_0/fc2/kernel")
    h = ln(h, g("policy/LayerNorm_0/scale"), g("policy/LayerNorm_0/bias"))
    head = lambda z, n: np.maximum(z @ g(n + "/layers_0/kernel") + g(n + "/layers_0/bias"), 0) @ g(n + "/layers_2/kernel")
    return head(h[:, 1::3], "action_outputs_0"), head(h[:, 0::3], "return_outputs_0")
