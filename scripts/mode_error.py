"""Cosine error of every operand mode against the numpy oracle (tests-side tool; not part of the product path).
usage: python scripts/mode_error.py [n_frames_b32] [n_frames_b16]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from arp_amd import clip, synth
from oracle import clip_np as C

n32 = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n16 = int(sys.argv[2]) if len(sys.argv) > 2 else max(2, n32 // 4)
for name, n in (("ViT-B/32", n32), ("ViT-B/16", n16)):
    if n <= 0:
        continue
    cfg = clip.MODELS[name]
    ocfg = C.ClipConfig(patch=cfg.patch)
    Wt = synth.clip_weights(ocfg, seed=0)
    fr = np.concatenate([synth.procgen_like_frames(n - n // 2, seed=1), synth.noise_frames(n // 2, 256, 256, seed=7)])
    tok = synth.prompt_tokens(1, 8, seed=2)
    ref = np.concatenate([C.compute_reward(Wt, ocfg, fr[i : i + 16], tok) for i in range(0, n, 16)])
    for mode in ("f32", "f16", "bf16"):
        m = clip.ClipLabeller(cfg, Wt, mode=mode).set_text(tok)
        got = m.label(fr)
        err = np.abs(got - ref) / 100.0
        print(f"{name} {mode}: frames {len(fr)} cosine err max {err.max():.3e} p99 {np.quantile(err, 0.99):.3e} rms {np.sqrt((err ** 2).mean()):.3e} "
              f"mean signed {((got - ref) / 100.0).mean():+.2e}; |cos| of the rewards up to {np.abs(ref).max() / 100:.3f}", flush=True)
        m.close()
