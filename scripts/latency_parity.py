#!/usr/bin/env python3
"""Row N4 parity evidence: rewards of the latency path (single-frame calls, and calls of 3 frames) against the fp64 oracle over a few
dozen frames, both CLIP models, f16 and bf16 operands, plus the heavy-tailed weight set of tests/test_clip_gpu.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import clip, synth
from oracle import clip_np as C

def run(name, nframes, W=None, tag=""):
    cfg = clip.MODELS[name]
    ocfg = C.ClipConfig(patch=cfg.patch)
    Wt = synth.clip_weights(ocfg, seed=0) if W is None else W
    fr = synth.procgen_like_frames(nframes, seed=11)
    tok = synth.prompt_tokens(1, 8, seed=2)
    ref = C.compute_reward(Wt, ocfg, fr, tok)
    for mode in ("f16", "bf16"):
        m = clip.ClipLabeller(cfg, Wt, mode=mode, n_streams=1).set_text(tok)
        one = np.concatenate([m.label(fr[i:i + 1]) for i in range(nframes)])
        three = np.concatenate([m.label(fr[i:i + 3]) for i in range(0, nframes, 3)])
        m.close()
        e1, e3 = np.abs(one - ref) / 100.0, np.abs(three - ref) / 100.0
        print(f"{name}{tag} {mode}: {nframes} frames, single-frame calls: cosine error max {e1.max():.2e} p99 {np.quantile(e1, 0.99):.2e} mean {e1.mean():.2e}; "
              f"3-frame calls: max {e3.max():.2e}", flush=True)

run("ViT-B/32", 48)
run("ViT-B/16", 24)
# heavy-tailed weights (outlier LayerNorm gains x 30, QuickGELU inputs in the tens, massive-activation channels)
cfg = clip.MODELS["ViT-B/32"]
W = synth.clip_weights(C.ClipConfig(patch=cfg.patch), seed=0)
rng = np.random.default_rng(7)
for i in range(12):
    p = f"visual.transformer.resblocks.{i}."
    for ln in ("ln_1", "ln_2"):
        W[p + ln + ".weight"][rng.choice(768, 6, replace=False)] *= 30.0
    b = W[p + "mlp.c_fc.bias"]
    b[rng.choice(3072, 24, replace=False)] = rng.choice([-20.0, 20.0], 24).astype(np.float32)
W["visual.ln_pre.weight"][rng.choice(768, 4, replace=False)] *= 30.0
och, osign = rng.choice(768, 3, replace=False), np.array([1.0, -1.0, 1.0], np.float32)
for i in range(12):
    W[f"visual.transformer.resblocks.{i}.mlp.c_proj.bias"][och] += 15.0 * osign
run("ViT-B/32", 24, W, " heavy-tailed")
