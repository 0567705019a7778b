"""Oracle (test infrastructure): forward_representation of the frozen M3AE image encoder (row N1), numpy.

Follows /root/reference/arp_dt/models/m3ae/model.py:
  * patchify "b (h p1) (w p2) c -> b (h w) (p1 p2 c)" ............ arp_dt/ARPDT.py:111-116
  * image_embedding Dense + 2-D sincos pos-emb + image type embedding, CLS without pos-emb;
    no text branch (use_text=False for ARP-DT) ....................... m3ae/model.py:471-496
  * get_2d_sincos_pos_embed ("w goes first") ......................... m3ae/model.py:95-136
  * Block: LN(eps 1e-6) -> Attention(qkv Dense WITH bias, softmax(q k^T * hd^-0.5), out Dense with bias)
    -> residual; LN -> fc1(bias) -> nn.gelu (tanh approximation) -> fc2(bias) -> residual . m3ae/model.py:200-283
  * Transformer: depth blocks then a final LayerNorm over ALL tokens ... m3ae/model.py:286-312
  * "base" = width 768, depth 12, heads 12 ........................... m3ae/model.py:935-941
Parameters are keyed by their Flax tree path flattened with '/', kernels [in, out].
Parity status: UNPINNED by the reference (flax absent, no tests); pinned against HuggingFace ViTModel
(gelu_pytorch_tanh, eps 1e-6) in tests/test_oracle_m3ae.py.
"""
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class EncConfig:
    patch: int = 16
    width: int = 768
    layers: int = 12
    heads: int = 12
    mlp_ratio: int = 4
    img_res: int = 256

    @property
    def grid(self):
        return self.img_res // self.patch

    @property
    def tokens(self):
        return self.grid * self.grid + 1


def param_shapes(cfg):
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


def sincos_1d(embed_dim, pos):
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1).astype(np.float64), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim, length):
    gs = int(round(length ** 0.5))
    assert gs * gs == length
    g = np.arange(gs, dtype=np.float64)
    grid = np.stack(np.meshgrid(g, g), axis=0).reshape(2, 1, gs, gs)  # here w goes first
    return np.concatenate([sincos_1d(embed_dim // 2, grid[0]), sincos_1d(embed_dim // 2, grid[1])], axis=1)


def patchify(x, p):
    n, H, W, c = x.shape
    return x.reshape(n, H // p, p, W // p, p, c).transpose(0, 1, 3, 2, 4, 5).reshape(n, (H // p) * (W // p), p * p * c)


def _ln(z, s, b):
    mu = z.mean(-1, keepdims=True)
    return (z - mu) / np.sqrt(((z - mu) ** 2).mean(-1, keepdims=True) + 1e-6) * s + b


def _gelu_tanh(y):
    return 0.5 * y * (1 + np.tanh(np.sqrt(2 / np.pi) * (y + 0.044715 * y ** 3)))


def forward_representation(P, cfg, images):
    """images: float [n, res, res, 3] (already normalised).  Returns [n, tokens, width] (fp64)."""
    g = lambda k: np.asarray(P[k], np.float64)
    D, hd = cfg.width, cfg.width // cfg.heads
    x = patchify(np.asarray(images, np.float64), cfg.patch)
    n, L, _ = x.shape
    x = x @ g("image_embedding/kernel") + g("image_embedding/bias") + sincos_2d(D, L) + g("encoder_image_type_embedding")[0]
    x = np.concatenate([np.broadcast_to(g("cls_token"), (n, 1, D)), x], axis=1)
    T = L + 1
    for i in range(cfg.layers):
        p = f"encoder/Block_{i}/"
        y = _ln(x, g(p + "LayerNorm_0/scale"), g(p + "LayerNorm_0/bias"))
        qkv = (y @ g(p + "Attention_0/Dense_0/kernel") + g(p + "Attention_0/Dense_0/bias")).reshape(n, T, 3, cfg.heads, hd)
        q, k, v = (qkv[:, :, j].transpose(0, 2, 1, 3) for j in range(3))
        s = q @ k.transpose(0, 1, 3, 2) * hd ** -0.5
        s = np.exp(s - s.max(-1, keepdims=True))
        s /= s.sum(-1, keepdims=True)
        y = (s @ v).transpose(0, 2, 1, 3).reshape(n, T, D)
        x = x + y @ g(p + "Attention_0/Dense_1/kernel") + g(p + "Attention_0/Dense_1/bias")
        y = _ln(x, g(p + "LayerNorm_1/scale"), g(p + "LayerNorm_1/bias"))
        y = _gelu_tanh(y @ g(p + "TransformerMLP_0/fc1/kernel") + g(p + "TransformerMLP_0/fc1/bias"))
        x = x + y @ g(p + "TransformerMLP_0/fc2/kernel") + g(p + "TransformerMLP_0/fc2/bias")
    return _ln(x, g("encoder/LayerNorm_0/scale"), g("encoder/LayerNorm_0/bias"))
