"""Oracle (test infrastructure): torch-CPU restatement of the labelling pass, fp32 -- the CPU
baseline timed beside the GPU number (bench.py ``cpu_baseline``, kind "port") and a second
independent cross-check of oracle/clip_np.py.

It runs the pass the way the reference's CPU branch does (/root/reference/arp_dt/label_reward.py:89,
132-146): a per-frame PIL transform loop (:109-121,134) followed by one batched fp32 forward of the
openai/CLIP architecture (nn.MultiheadAttention semantics, QuickGELU, LayerNorm eps 1e-5).
Parity status: unpinned by reference-held vectors; agrees with clip_np.py (tests/test_oracle.py::test_torch_port_matches_numpy_oracle).
"""
import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

MEAN = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(3, 1, 1)
STD = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(3, 1, 1)


def pil_transform(frame_u8, use_crop=False, n_px=224):
    """One frame through the reference transform (PIL resize, the per-frame loop of :134)."""
    im = Image.fromarray(frame_u8)
    if use_crop:
        w, h = im.size
        c = w // 2
        left, top = int(round((w - c) / 2.0)), int(round((h - c) / 2.0))
        im = im.crop((left, top, left + c, top + c))
        im = im.resize((n_px, n_px), Image.BICUBIC)
    else:
        w, h = im.size
        if h <= w:
            oh, ow = n_px, int(n_px * w / h)
        else:
            oh, ow = int(n_px * h / w), n_px
        im = im.resize((ow, oh), Image.BICUBIC)
        left, top = int(round((ow - n_px) / 2.0)), int(round((oh - n_px) / 2.0))
        im = im.crop((left, top, left + n_px, top + n_px))
    x = torch.from_numpy(np.array(im.convert("RGB"))).permute(2, 0, 1).float().div(255.0)
    return (x - MEAN) / STD


def _block(x, W, pre, heads, mask):
    g = lambda k: W[pre + k]
    n, t, d = x.shape
    h = F.layer_norm(x, (d,), g("ln_1.weight"), g("ln_1.bias"), 1e-5)
    qkv = F.linear(h, g("attn.in_proj_weight"), g("attn.in_proj_bias"))
    q, k, v = qkv.view(n, t, 3, heads, d // heads).permute(2, 0, 3, 1, 4)
    a = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
    a = a.transpose(1, 2).reshape(n, t, d)
    x = x + F.linear(a, g("attn.out_proj.weight"), g("attn.out_proj.bias"))
    h = F.layer_norm(x, (d,), g("ln_2.weight"), g("ln_2.bias"), 1e-5)
    h = F.linear(h, g("mlp.c_fc.weight"), g("mlp.c_fc.bias"))
    h = h * torch.sigmoid(1.702 * h)
    return x + F.linear(h, g("mlp.c_proj.weight"), g("mlp.c_proj.bias"))


def to_torch(W):
    return {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in W.items()}


def finetune_transform(frames_u8, n_px=224):
    """finetune_module/clip_multiscale_adapter.py:120-132 without the random ColorJitter: float, torchvision tensor resize
    (= bilinear, align_corners False, no antialias; only when BOTH sides differ from 224), / 255, normalise."""
    x = torch.from_numpy(np.asarray(frames_u8)).permute(0, 3, 1, 2).float()
    if x.shape[2] != n_px and x.shape[3] != n_px:
        x = F.interpolate(x, size=(n_px, n_px), mode="bilinear", align_corners=False)
    x = x / 255.0
    return (x - MEAN.view(1, 3, 1, 1)) / STD.view(1, 3, 1, 1)


@torch.no_grad()
def encode_image_multiscale(W, cfg, x):
    """CLIP image feature plus the CLS token after every resblock, concatenated block 0..L-1 -- what the reference's forward
    hooks collect (finetune_module/utils.py:6-18, clip_multiscale_adapter.py:137-142)."""
    d = cfg.width
    x = F.conv2d(x, W["visual.conv1.weight"], stride=cfg.patch)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([W["visual.class_embedding"].expand(x.shape[0], 1, d), x], 1) + W["visual.positional_embedding"]
    x = F.layer_norm(x, (d,), W["visual.ln_pre.weight"], W["visual.ln_pre.bias"], 1e-5)
    inter = []
    for i in range(cfg.layers):
        x = _block(x, W, f"visual.transformer.resblocks.{i}.", cfg.heads, None)
        inter.append(x[:, 0])
    c = F.layer_norm(x[:, 0], (d,), W["visual.ln_post.weight"], W["visual.ln_post.bias"], 1e-5)
    return torch.cat(inter, -1), c @ W["visual.proj"]


@torch.no_grad()
def encode_text_multiscale(W, cfg, tokens):
    """Text feature plus the EOT-token row after every resblock (clip_multiscale_adapter.py:160-165)."""
    tokens = torch.as_tensor(np.asarray(tokens), dtype=torch.long)
    x = W["token_embedding.weight"][tokens] + W["positional_embedding"]
    t = x.shape[1]
    mask = torch.full((t, t), float("-inf")).triu_(1)
    rows, eot = torch.arange(x.shape[0]), tokens.argmax(-1)
    inter = []
    for i in range(cfg.txt_layers):
        x = _block(x, W, f"transformer.resblocks.{i}.", cfg.txt_heads, mask)
        inter.append(x[rows, eot])
    x = F.layer_norm(x, (cfg.txt_width,), W["ln_final.weight"], W["ln_final.bias"], 1e-5)
    return torch.cat(inter, -1), x[rows, eot] @ W["text_projection"]


@torch.no_grad()
def encode_image(W, cfg, x):
    d = cfg.width
    x = F.conv2d(x, W["visual.conv1.weight"], stride=cfg.patch)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([W["visual.class_embedding"].expand(x.shape[0], 1, d), x], 1) + W["visual.positional_embedding"]
    x = F.layer_norm(x, (d,), W["visual.ln_pre.weight"], W["visual.ln_pre.bias"], 1e-5)
    for i in range(cfg.layers):
        x = _block(x, W, f"visual.transformer.resblocks.{i}.", cfg.heads, None)
    c = F.layer_norm(x[:, 0], (d,), W["visual.ln_post.weight"], W["visual.ln_post.bias"], 1e-5)
    return c @ W["visual.proj"]


@torch.no_grad()
def encode_text(W, cfg, tokens):
    tokens = torch.as_tensor(np.asarray(tokens), dtype=torch.long)
    x = W["token_embedding.weight"][tokens] + W["positional_embedding"]
    t = x.shape[1]
    mask = torch.full((t, t), float("-inf")).triu_(1)
    for i in range(cfg.txt_layers):
        x = _block(x, W, f"transformer.resblocks.{i}.", cfg.txt_heads, mask)
    x = F.layer_norm(x, (cfg.txt_width,), W["ln_final.weight"], W["ln_final.bias"], 1e-5)
    return x[torch.arange(x.shape[0]), tokens.argmax(-1)] @ W["text_projection"]


@torch.no_grad()
def compute_reward(W, cfg, frames_u8, tokens, use_crop=False, text_feat=None):
    """frames -> rewards exactly as label_reward.py:132-146 does on its CPU branch."""
    x = torch.stack([pil_transform(f, use_crop) for f in frames_u8])
    img = encode_image(W, cfg, x)
    txt = encode_text(W, cfg, tokens) if text_feat is None else text_feat
    img = img / img.norm(dim=1, keepdim=True)
    txt = txt / txt.norm(dim=1, keepdim=True)
    return (W["logit_scale"].exp() * txt @ img.t())[0].numpy()
