"""Seeded synthetic parameters and batches for the ARP-DT policy.

Dense kernels lecun-normal (std fan_in^-0.5), biases perturbed away from 0 so parity tests exercise
them, action Embed N(0, 1/emb), LayerNorm scale ~1 / bias ~0, residual_weight = 4.0 (ARPDT.py:145-150).
Keys are the Flax tree paths flattened with '/' (SURVEY.md Appendix C); kernels are [in, out]."""
import numpy as np


def policy_param_shapes(cfg):
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


def policy_params(cfg, seed=0, dtype=np.float32):
    rng = np.random.default_rng(seed)
    P = {}
    for name, shape in policy_param_shapes(cfg).items():
        if name == "residual_weight":
            v = np.full(shape, 4.0)
        elif name.endswith("/scale"):
            v = 1.0 + 0.05 * rng.standard_normal(shape)
        elif name.endswith("/bias"):
            v = 0.02 * rng.standard_normal(shape)
        elif name.endswith("embedding"):
            v = rng.standard_normal(shape) / np.sqrt(shape[1])
        else:
            v = rng.standard_normal(shape) / np.sqrt(shape[0])
        P[name] = v.astype(dtype)
    return P


def policy_batch(cfg, batch, seed=0, dtype=np.float32):
    """enc ~ N(0,1) [B,T,tokens,dim], action ~ U{0..n_actions-1} int32 [B,T], rtg ~ U(0,1) [B,T,1]."""
    rng = np.random.default_rng(seed)
    enc = rng.standard_normal((batch, cfg.window, cfg.enc_tokens, cfg.enc_dim)).astype(dtype)
    action = rng.integers(0, cfg.n_actions, (batch, cfg.window)).astype(np.int32)
    rtg = rng.random((batch, cfg.window, 1)).astype(dtype)
    return enc, action, rtg


def m3ae_param_shapes(cfg):
    """Flax tree of the M3AE image encoder used by forward_representation (m3ae/model.py:370-430,471-496)."""
    D, P, H = cfg.width, cfg.patch, cfg.mlp_ratio * cfg.width
    s = {"cls_token": (1, 1, D), "encoder_image_type_embedding": (1, 1, D), "image_embedding/kernel": (P * P * 3, D),
         "image_embedding/bias": (D,)}
    for i in range(cfg.layers):
        p = f"encoder/Block_{i}/"
        for ln in ("LayerNorm_0", "LayerNorm_1"):
            s[p + ln + "/scale"] = (D,)
            s[p + ln + "/bias"] = (D,)
        s[p + "Attention_0/Dense_0/kernel"] = (D, 3 * D)
        s[p + "Attention_0/Dense_0/bias"] = (3 * D,)
        s[p + "Attention_0/Dense_1/kernel"] = (D, D)
        s[p + "Attention_0/Dense_1/bias"] = (D,)
        s[p + "TransformerMLP_0/fc1/kernel"] = (D, H)
        s[p + "TransformerMLP_0/fc1/bias"] = (H,)
        s[p + "TransformerMLP_0/fc2/kernel"] = (H, D)
        s[p + "TransformerMLP_0/fc2/bias"] = (D,)
    s["encoder/LayerNorm_0/scale"] = (D,)
    s["encoder/LayerNorm_0/bias"] = (D,)
    return s


def m3ae_params(cfg, seed=0, dtype=np.float32):
    """Seeded random-init encoder weights (xavier-like kernels, small biases, LN ~ identity)."""
    rng = np.random.default_rng(seed)
    P = {}
    for name, shape in m3ae_param_shapes(cfg).items():
        if name.endswith("/scale"):
            v = 1.0 + 0.05 * rng.standard_normal(shape)
        elif name.endswith("/bias") or name in ("cls_token", "encoder_image_type_embedding"):
            v = 0.02 * rng.standard_normal(shape)
        else:
            v = rng.standard_normal(shape) * np.sqrt(2.0 / (shape[0] + shape[1]))
        P[name] = v.astype(dtype)
    return P


def normalized_frames(n, res=256, seed=0, dtype=np.float32):
    """Frames as the training pipeline hands them to the encoder: float NHWC, normalised with the Procgen
    statistics of the reference's augmentation (main_procgen.py:232-276)."""
    from .synth import procgen_like_frames
    x = procgen_like_frames(n, res, res, seed=seed).astype(np.float32) / 255.0
    mean = np.array([0.5762, 0.5503, 0.5213], np.float32)
    std = np.array([0.3207, 0.3169, 0.3307], np.float32)
    return ((x - mean) / std).astype(dtype)
