"""Oracle (test infrastructure): CLIP ViT image tower + text tower + scaled cosine similarity, numpy.

The reference's labelling pass (/root/reference/arp_dt/label_reward.py:132-146) calls the
third-party PyTorch package ``clip`` (openai/CLIP, pinned at commit
d50d76daa670286dd6cacf3bcd80b5e4823fc8e1 in /root/reference/requirements.txt:17), which is absent
from /root/reference.  This file restates that package's published algorithm, following the
reference's own in-tree Flax mirror of it line by line:

  * LayerNorm eps 1e-5 ........................ arp_dt/models/openai/layers.py:9
  * QuickGELU x*sigmoid(1.702x) ............... arp_dt/models/openai/layers.py:12-13
  * MLP c_fc -> QuickGELU -> c_proj ........... arp_dt/models/openai/layers.py:223-232
  * ResidualAttentionBlock (pre-LN) ........... arp_dt/models/openai/layers.py:235-250
  * VisionTransformer (conv1 no bias, CLS,
    pos-emb, ln_pre, blocks, ln_post(CLS), proj) arp_dt/models/openai/layers.py:274-336
  * TextEncoder (embed + pos, causal blocks,
    ln_final, EOT row = argmax(text), proj) ... arp_dt/models/openai/layers.py:339-370
  * L2-normalise ............................... arp_dt/models/openai/layers.py:429-439
  * reward = logits_per_text[0]
           = exp(logit_scale) * <txt_n, img_n> . arp_dt/label_reward.py:140-146
  * weight names / layouts (openai state dict) . arp_dt/models/openai/model.py:220-314

Parity status: UNPINNED by reference-held vectors (the reference has none).  Pinned instead by
``tests/test_oracle.py`` against HuggingFace ``CLIPModel`` (quick_gelu) on the same weights.
"""
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class ClipConfig:
    patch: int = 32
    width: int = 768
    layers: int = 12
    heads: int = 12
    embed: int = 512
    img_res: int = 224
    txt_width: int = 512
    txt_layers: int = 12
    txt_heads: int = 8
    ctx: int = 77
    vocab: int = 49408

    @property
    def grid(self):
        return self.img_res // self.patch

    @property
    def tokens(self):
        return self.grid * self.grid + 1


VIT_B32 = ClipConfig(patch=32)  # arp_dt/models/openai/model.py:60-69
VIT_B16 = ClipConfig(patch=16)  # arp_dt/models/openai/model.py:70-79


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def quick_gelu(x):
    return x * (1.0 / (1.0 + np.exp(-1.702 * x)))


def _softmax(s):
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    return e / e.sum(-1, keepdims=True)


def mha(x, w_in, b_in, w_out, b_out, heads, causal, fp8=False):
    """torch ``nn.MultiheadAttention`` self-attention: fused in-proj [3D, D], rows ordered q,k,v.
    fp8: the product's optional e4m3 projections (csrc/tower.h::tower_attn_fp8): ln_1 output x 32 and the attention output x 16 as
    e4m3, both weight matrices x a per-tensor power of two as e4m3; the attention itself stays on the 16-bit path."""
    n, t, d = x.shape
    hd = d // heads
    if fp8:
        si = _pow2_scale(w_in)
        qkv = (quant_e4m3(x * 32.0) @ quant_e4m3(w_in * si).T) / (32.0 * si) + b_in
    else:
        qkv = x @ w_in.T + b_in
    q, k, v = np.split(qkv, 3, axis=-1)
    sh = lambda a: a.reshape(n, t, heads, hd).transpose(0, 2, 1, 3)
    q, k, v = sh(q), sh(k), sh(v)
    s = (q * (hd ** -0.5)) @ k.transpose(0, 1, 3, 2)
    if causal:
        s = np.where(np.triu(np.ones((t, t), bool), 1), -np.inf, s)
    p = _softmax(s)
    o = (p @ v).transpose(0, 2, 1, 3).reshape(n, t, d)
    if fp8:
        so = _pow2_scale(w_out)
        return (quant_e4m3(o * 16.0) @ quant_e4m3(w_out * so).T) / (16.0 * so) + b_out
    return o @ w_out.T + b_out


def quant_e4m3(x):
    """Round to OCP e4m3fn (4 exponent bits, bias 7, 3 mantissa bits, subnormals down to 2^-9, NO infinities: saturates at 448),
    round-to-nearest-even -- what v_cvt_pk_fp8_f32 behind a clamp produces.  Emulation for the product's optional fp8 MLP mode."""
    x = np.asarray(x, np.float64)
    a = np.minimum(np.abs(x), 448.0)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -20)))
    e = np.clip(e, -6, 8)                    # below 2^-6 the spacing stays 2^-9 (subnormals)
    step = 2.0 ** (e - 3)
    q = np.round(a / step) * step            # numpy rounds half to even
    return np.sign(x) * np.minimum(q, 448.0)


def _pow2_scale(w):
    return 2.0 ** np.floor(np.log2(240.0 / np.abs(w).max()))


def resblock(x, W, pre, heads, causal, mlp_fp8=False, attn_fp8=False):
    g = lambda k: W[pre + k]
    h = layer_norm(x, g("ln_1.weight"), g("ln_1.bias"))
    x = x + mha(h, g("attn.in_proj_weight"), g("attn.in_proj_bias"), g("attn.out_proj.weight"),
                g("attn.out_proj.bias"), heads, causal, fp8=attn_fp8)
    h = layer_norm(x, g("ln_2.weight"), g("ln_2.bias"))
    if mlp_fp8:
        # the product's fp8 MLP (csrc/tower.h::tower_mlp_fp8): e4m3 activations x 32 / x 16, e4m3 weights x a per-tensor power of two
        w1, w2 = g("mlp.c_fc.weight"), g("mlp.c_proj.weight")
        s1, s2 = _pow2_scale(w1), _pow2_scale(w2)
        u = (quant_e4m3(h * 32.0) @ quant_e4m3(w1 * s1).T) / (32.0 * s1) + g("mlp.c_fc.bias")
        h = quant_e4m3(quick_gelu(u) * 16.0)
        return x + (h @ quant_e4m3(w2 * s2).T) / (16.0 * s2) + g("mlp.c_proj.bias")
    h = quick_gelu(h @ g("mlp.c_fc.weight").T + g("mlp.c_fc.bias"))
    return x + h @ g("mlp.c_proj.weight").T + g("mlp.c_proj.bias")


def encode_image(W, cfg, x_nchw, return_tokens=False, mlp_fp8=False, attn_fp8=False):
    """x_nchw: float [n,3,R,R] already normalised.  Returns un-normalised features [n, embed]."""
    n = x_nchw.shape[0]
    P, G, D = cfg.patch, cfg.grid, cfg.width
    # conv P x P stride P, no bias == GEMM over patches; k index = c*P*P + py*P + px
    p = x_nchw.reshape(n, 3, G, P, G, P).transpose(0, 2, 4, 1, 3, 5).reshape(n, G * G, 3 * P * P)
    x = p @ W["visual.conv1.weight"].reshape(D, -1).T
    cls = np.broadcast_to(W["visual.class_embedding"], (n, 1, D))
    x = np.concatenate([cls, x], axis=1) + W["visual.positional_embedding"]
    x = layer_norm(x, W["visual.ln_pre.weight"], W["visual.ln_pre.bias"])
    for i in range(cfg.layers):
        # (the product keeps the class-token-only last block on 16-bit operands: its row count is below the fp8 kernel's tile)
        x = resblock(x, W, f"visual.transformer.resblocks.{i}.", cfg.heads, causal=False, mlp_fp8=mlp_fp8 and i < cfg.layers - 1,
                     attn_fp8=attn_fp8 and i < cfg.layers - 1)
    if return_tokens:
        return x
    c = layer_norm(x[:, 0], W["visual.ln_post.weight"], W["visual.ln_post.bias"])
    return c @ W["visual.proj"]


def encode_text(W, cfg, tokens):
    """tokens: int [p, ctx].  Returns un-normalised features [p, embed]."""
    tokens = np.asarray(tokens)
    x = W["token_embedding.weight"][tokens] + W["positional_embedding"]
    for i in range(cfg.txt_layers):
        x = resblock(x, W, f"transformer.resblocks.{i}.", cfg.txt_heads, causal=True)
    x = layer_norm(x, W["ln_final.weight"], W["ln_final.bias"])
    eot = tokens.argmax(-1)
    return x[np.arange(x.shape[0]), eot] @ W["text_projection"]


def l2n(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def rewards_from_features(W, img_feat, txt_feat):
    """``logits_per_text[0]`` of openai CLIP.forward (label_reward.py:141-145)."""
    scale = np.exp(W["logit_scale"])
    return (scale * (l2n(txt_feat) @ l2n(img_feat).T))[0]


def cast_weights(W, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in W.items()}


def compute_reward(W, cfg, frames_u8, tokens, use_crop=False, dtype=np.float64):
    """The reference's ``compute_reward`` closure end to end (label_reward.py:132-146)."""
    from . import preprocess as pp

    Wd = cast_weights(W, dtype)
    x = pp.preprocess(frames_u8, use_crop=use_crop).astype(dtype)
    f = encode_image(Wd, cfg, x)
    t = encode_text(Wd, cfg, tokens)
    return rewards_from_features(Wd, f, t).astype(np.float32)
