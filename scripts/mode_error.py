"""Cosine error of every operand mode against the numpy oracle (tests-side tool; not part of the product path).
usage: python scripts/mode_error.py [n_frames]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from arp_amd import clip, synth
from oracle import clip_np as C

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for name in ("ViT-B/32", "ViT-B/16"):
    cfg = clip.MODELS[name]
    ocfg = C.ClipConfig(patch=cfg.patch)
    Wt = synth.clip_weights(ocfg, seed=0)
    fr = synth.procgen_like_frames(n if name == "ViT-B/32" else max(2, n // 4), seed=1)
    tok = synth.prompt_tokens(1, 8, seed=2)
    ref = C.compute_reward(Wt, ocfg, fr, tok)
    for mode in ("f32", "f16", "bf16"):
        m = clip.ClipLabeller(cfg, Wt, mode=mode).set_text(tok)
        got = m.label(fr)
        err = np.abs(got - ref) / 100.0
        print(f"{name} {mode}: frames {len(fr)} cosine err max {err.max():.3e} rms {np.sqrt((err ** 2).mean()):.3e}", flush=True)
        m.close()
