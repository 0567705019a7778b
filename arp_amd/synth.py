"""Seeded synthetic inputs: random-init CLIP weights (openai state-dict names), Procgen-like
frames and prompt token ids.

There is no network, hence no pretrained ``ViT-B-32.pt`` and no BPE vocabulary; benchmarks and
parity tests run on seeded random-init weights of the real architecture and on synthetic token ids
(SURVEY.md section 8d).  Weight names and layouts are the openai/CLIP state-dict ones the
reference's converter consumes (/root/reference/arp_dt/models/openai/model.py:220-314), so a real
checkpoint's ``state_dict()`` can be passed to :class:`arp_amd.clip.ClipLabeller` unchanged.
"""
import numpy as np

SOT, EOT = 49406, 49407  # openai/CLIP tokenizer specials (label_reward.py:136-138 -> clip.tokenize)


def clip_weights(cfg, seed=0, dtype=np.float32):
    """openai-style init.  Biases / LayerNorm affine parameters
    are perturbed away from 0 / 1 so that parity tests exercise them."""
    rng = np.random.default_rng(seed)
    W = {}

    def nrm(shape, std):
        return (rng.standard_normal(shape) * std).astype(dtype)

    def ln(prefix, d):
        W[prefix + ".weight"] = (1.0 + 0.05 * rng.standard_normal(d)).astype(dtype)
        W[prefix + ".bias"] = nrm((d,), 0.02)

    def tower(prefix, d, layers):
        attn_std = d ** -0.5
        proj_std = d ** -0.5 * (2 * layers) ** -0.5
        fc_std = (2 * d) ** -0.5
        for i in range(layers):
            p = f"{prefix}resblocks.{i}."
            ln(p + "ln_1", d)
            W[p + "attn.in_proj_weight"] = nrm((3 * d, d), attn_std)
            W[p + "attn.in_proj_bias"] = nrm((3 * d,), 0.02)
            W[p + "attn.out_proj.weight"] = nrm((d, d), proj_std)
            W[p + "attn.out_proj.bias"] = nrm((d,), 0.02)
            ln(p + "ln_2", d)
            W[p + "mlp.c_fc.weight"] = nrm((4 * d, d), fc_std)
            W[p + "mlp.c_fc.bias"] = nrm((4 * d,), 0.02)
            W[p + "mlp.c_proj.weight"] = nrm((d, 4 * d), proj_std)
            W[p + "mlp.c_proj.bias"] = nrm((d,), 0.02)

    D, P = cfg.width, cfg.patch
    W["visual.conv1.weight"] = nrm((D, 3, P, P), (3 * P * P) ** -0.5)
    W["visual.class_embedding"] = nrm((D,), D ** -0.5)
    W["visual.positional_embedding"] = nrm((cfg.tokens, D), D ** -0.5)
    ln("visual.ln_pre", D)
    tower("visual.transformer.", D, cfg.layers)
    ln("visual.ln_post", D)
    W["visual.proj"] = nrm((D, cfg.embed), D ** -0.5)

    T = cfg.txt_width
    W["token_embedding.weight"] = nrm((cfg.vocab, T), 0.02)
    W["positional_embedding"] = nrm((cfg.ctx, T), 0.01)
    tower("transformer.", T, cfg.txt_layers)
    ln("ln_final", T)
    W["text_projection"] = nrm((T, cfg.embed), T ** -0.5)
    W["logit_scale"] = np.asarray(np.log(100.0), dtype=dtype)  # pretrained CLIP value
    return W


def procgen_like_frames(n, h=256, w=256, seed=0, noise=True):
    """uint8 NHWC frames: blocky 'sprite' structure plus optional per-pixel noise."""
    rng = np.random.default_rng(seed)
    cell = 16
    base = rng.integers(0, 256, (n, (h + cell - 1) // cell, (w + cell - 1) // cell, 3), dtype=np.uint8)
    x = np.repeat(np.repeat(base, cell, axis=1), cell, axis=2)[:, :h, :w]
    if noise:
        x = (x.astype(np.int16) + rng.integers(-24, 25, x.shape, dtype=np.int16)).clip(0, 255).astype(np.uint8)
    return np.ascontiguousarray(x)


def noise_frames(n, h=256, w=256, seed=0):
    """``default_rng(seed).integers(0,256,(n,h,w,3),uint8)`` -- BASELINE config 1/2 frames."""
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)


def prompt_tokens(n_prompts=1, length=8, ctx=77, vocab=49408, seed=0):
    """Synthetic ``clip.tokenize`` output: [SOT, ids..., EOT, 0-pad] as int32 [n_prompts, ctx]."""
    rng = np.random.default_rng(seed)
    t = np.zeros((n_prompts, ctx), np.int32)
    for i in range(n_prompts):
        ln_i = length if np.isscalar(length) else length[i]
        t[i, 0] = min(SOT, vocab - 2)
        t[i, 1 : 1 + ln_i] = rng.integers(1, min(SOT, vocab - 2), ln_i)
        t[i, 1 + ln_i] = min(EOT, vocab - 1)
    return t
