"""CPU oracle (TEST INFRASTRUCTURE ONLY) for SURVEY row N2: the trainable head of the CLIP multi-scale adapter
fine-tune step, restated in torch (float64 by default) with autograd for the gradients.

Follows /root/reference/finetune_module/clip_multiscale_adapter.py:
  * encode_image :134-149  -- per-block CLS features (detached by the hooks, utils.py:6-18) -> bias-free Linear ->
                              concat with the CLIP image feature -> res*f + (1-res)*AdapterMLP(f) -> F.normalize
                              (note: here res weights the ORIGINAL feature, the opposite of ARPDT.py:466-472)
  * encode_text  :151-175  -- same on the EOT-token features of the text tower
  * forward      :177-250  -- VIP loss (:214-220; `r` is [B,1] and the scores are [B], so the exponent BROADCASTS to a
                              [B,B] matrix before the mean -- kept), inverse-dynamics CE on concat[a1,t,a2,t] (:232-237),
                              total = vip + lambda_id * id with lambda_id a learnable scalar used as is (:103,246)
  * AdapterMLP   layers.py:6-60 with num_layers = 2: Linear -> Identity -> ReLU -> Linear
and finetune.py:139-141 for the optimiser: CLIP frozen, torch.optim.AdamW(lr, weight_decay) over everything else.

Pinned by tests/golden/finetune_tiny.npz: outputs and gradients of the REFERENCE CLASS ITSELF executed in the build
container on a stub CLIP (tests/golden/make_golden_finetune.py).  The reference has no tests of its own for this
path: parity unpinned by the reference beyond that fixture.
"""
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class HeadConfig:
    layers: int = 12        # CLIP blocks whose CLS / EOT features are concatenated (clip_model.transformer.layers)
    width_v: int = 768      # vision tower width
    width_t: int = 512      # text tower width (= embed = input_dim = output_dim in the reference's constructor defaults)
    embed: int = 512
    hidden: int = 1024      # hidden_dim; the adapters use hidden*(layers+1), the inverse model uses hidden
    n_actions: int = 15
    gamma: float = 0.98
    logit_scale: float = float(np.log(1 / 0.07))
    use_vip: bool = True
    use_id: bool = True
    goal_conditioned: bool = False  # clip_multiscale_adapter.py:208-212,224-230: image3 stands where the prompt stands

    @property
    def d_img(self):
        return self.layers * self.width_v

    @property
    def d_txt(self):
        return self.layers * self.width_t

    @property
    def feat(self):
        return self.layers * self.width_t + self.embed

    @property
    def adapter_hidden(self):
        return self.hidden * (self.layers + 1)


def param_shapes(cfg):
    """torch state_dict names of CLIPMultiscaleAdapter without clip_model.* (Linear weights are [out, in])."""
    Fd, Hd = cfg.feat, cfg.adapter_hidden
    s = {"image_intermediate_linear.weight": (cfg.d_txt, cfg.d_img), "text_intermediate_linear.weight": (cfg.d_txt, cfg.d_txt)}
    for a in ("image_adapter", "text_adapter"):
        s[f"{a}.layers.0.weight"] = (Hd, Fd)
        s[f"{a}.layers.0.bias"] = (Hd,)
        s[f"{a}.layers.3.weight"] = (Fd, Hd)
        s[f"{a}.layers.3.bias"] = (Fd,)
    s["inverse_layer.layers.0.weight"] = (cfg.hidden, 4 * Fd)
    s["inverse_layer.layers.0.bias"] = (cfg.hidden,)
    s["inverse_layer.layers.3.weight"] = (cfg.n_actions, cfg.hidden)
    s["inverse_layer.layers.3.bias"] = (cfg.n_actions,)
    s["image_residual_weight"] = ()
    s["text_residual_weight"] = ()
    s["lambda_id"] = ()
    return s


def init_params(cfg, seed=0, scale=1.0):
    """Seeded stand-in init (the reference uses orthogonal weights, zero biases, 4.0 residual weights, lambda_id = ln(1/0.07))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    P = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith("residual_weight"):
            P[k] = np.float32(4.0) + np.zeros(shp, np.float32)
        elif k == "lambda_id":
            P[k] = np.float32(np.log(1 / 0.07)) + np.zeros(shp, np.float32)
        elif k.endswith(".bias"):
            P[k] = (0.02 * rng.standard_normal(shp)).astype(np.float32)
        else:
            P[k] = (scale * rng.standard_normal(shp) / np.sqrt(shp[1])).astype(np.float32)
    return P


def _adapter(P, name, x):
    h = F.relu(F.linear(x, P[f"{name}.layers.0.weight"], P[f"{name}.layers.0.bias"]))
    return F.linear(h, P[f"{name}.layers.3.weight"], P[f"{name}.layers.3.bias"])


def _encode(P, which, inter, final):
    u = F.linear(inter, P[f"{which}_intermediate_linear.weight"])
    f = torch.cat([u, final], dim=-1)
    res = torch.sigmoid(P[f"{which}_residual_weight"])
    y = res * f + (1.0 - res) * _adapter(P, f"{which}_adapter", f)
    return F.normalize(y, dim=-1)


def forward(P, cfg, img_inter, img_final, txt_inter, txt_final, r, action):
    """img_inter [3,B,d_img], img_final [3,B,embed] (frames 0,1,2 of each sample), txt_inter [B,d_txt], txt_final [B,embed],
    r [B] as stored in the batch (the loss uses r - 1), action [B] int64.  goal_conditioned: FOUR image groups, txt_* = None."""
    a = [_encode(P, "image", img_inter[k], img_final[k]) for k in range(3)]
    if cfg.goal_conditioned:
        # :208-212 -- img_inter / img_final carry FOUR groups (image0..image3); the scores are negative distances to the goal frame's
        # adapted feature, which also takes the prompt's two slots of the inverse-model input (:224-230); no text tower, no logit scale
        t = _encode(P, "image", img_inter[3], img_final[3])
        s = [-torch.linalg.norm(t - a[k], dim=-1) for k in range(3)]
    else:
        t = _encode(P, "text", txt_inter, txt_final)
        scale = float(np.exp(cfg.logit_scale))
        s = [scale * (a[k] * t).sum(-1) for k in range(3)]
    rr = (r - 1.0).reshape(-1, 1)  # [B,1] against [B] scores: a [B,B] exponent, exactly as the reference broadcasts
    vip = (1 - cfg.gamma) * -s[0].mean() + torch.log(1e-8 + torch.mean(torch.exp(-(rr + cfg.gamma * s[2] - s[1]))))
    c = torch.cat([a[1], t, a[2], t], dim=-1)
    h = F.relu(F.linear(c, P["inverse_layer.layers.0.weight"], P["inverse_layer.layers.0.bias"]))
    logits = F.linear(h, P["inverse_layer.layers.3.weight"], P["inverse_layer.layers.3.bias"])
    idl = F.cross_entropy(logits, action)
    loss = 0.0
    if cfg.use_vip:
        loss = loss + vip
    if cfg.use_id:
        loss = loss + P["lambda_id"] * idl
    return {"loss": loss, "vip_loss": vip, "id_loss": idl, "scores": torch.stack(s), "logits": logits, "adapted_image": torch.stack(a),
            "adapted_text": t}


def to_torch(P, dtype=torch.float64, requires_grad=False):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad) for k, v in P.items()}


def grads(P, cfg, batch, dtype=torch.float64):
    Pt = to_torch(P, dtype, requires_grad=True)
    out = forward(Pt, cfg, *[None if b is None else torch.as_tensor(b, dtype=dtype) for b in batch[:5]], torch.as_tensor(batch[5], dtype=torch.long))
    out["loss"].backward()
    g = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).detach().numpy() for k, v in Pt.items()}
    aux = {k: float(out[k].detach()) for k in ("loss", "vip_loss", "id_loss")}
    aux["no_grad"] = tuple(k for k, v in Pt.items() if v.grad is None)  # torch.optim.AdamW leaves these untouched
    return g, aux


def adamw_step(P, M, V, G, step, lr, weight_decay, b1=0.9, b2=0.999, eps=1e-8, no_grad=()):
    """torch.optim.AdamW semantics (decoupled decay on every parameter THAT HAS A GRADIENT -- one whose .grad is None is skipped
    entirely: no decay, no moment update -- bias correction with t = step + 1)."""
    t = step + 1
    out_p, out_m, out_v = {}, {}, {}
    for k in P:
        if k in no_grad:
            out_p[k], out_m[k], out_v[k] = np.asarray(P[k], np.float64), M[k], V[k]
            continue
        p = np.asarray(P[k], np.float64) * (1.0 - lr * weight_decay)
        m = b1 * np.asarray(M[k], np.float64) + (1 - b1) * G[k]
        v = b2 * np.asarray(V[k], np.float64) + (1 - b2) * G[k] * G[k]
        denom = np.sqrt(v) / np.sqrt(1 - b2 ** t) + eps
        out_p[k] = p - (lr / (1 - b1 ** t)) * m / denom
        out_m[k], out_v[k] = m, v
    return out_p, out_m, out_v


def train_steps(P, cfg, batches, lr, weight_decay, n_steps):
    M = {k: np.zeros_like(np.asarray(v, np.float64)) for k, v in P.items()}
    V = {k: np.zeros_like(np.asarray(v, np.float64)) for k, v in P.items()}
    P = {k: np.asarray(v, np.float64) for k, v in P.items()}
    aux = []
    for i in range(n_steps):
        g, a = grads(P, cfg, batches[i % len(batches)])
        aux.append(a)
        P, M, V = adamw_step(P, M, V, g, i, lr, weight_decay, no_grad=a["no_grad"])
    return P, aux
