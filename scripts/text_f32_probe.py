"""Where does the common-mode part of the 16-bit modes' reward error come from?  Rewards recomputed on the host from 16-bit image
features and (a) the same mode's text feature, (b) the f32 mode's text feature, against the oracle."""
import sys
import numpy as np
sys.path.insert(0, ".")
from arp_amd import clip, synth
from oracle import clip_np as C

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = clip.MODELS["ViT-B/32"]
ocfg = C.ClipConfig(patch=cfg.patch)
Wt = synth.clip_weights(ocfg, seed=0)
fr = np.concatenate([synth.procgen_like_frames(n - n // 2, seed=1), synth.noise_frames(n // 2, 256, 256, seed=7)])
tok = synth.prompt_tokens(1, 8, seed=2)
ref = np.concatenate([C.compute_reward(Wt, ocfg, fr[i : i + 16], tok) for i in range(0, n, 16)]) / 100.0
m32 = clip.ClipLabeller(cfg, Wt, mode="f32").set_text(tok)
t32 = m32.text_features()[0]
m32.close()
for mode in ("f16", "bf16"):
    m = clip.ClipLabeller(cfg, Wt, mode=mode).set_text(tok)
    img = m.encode_image(fr, normalize=True)
    t = m.text_features()[0]
    for name, tt in (("own text", t), ("f32 text", t32)):
        e = img @ tt - ref
        print(f"{mode} image features x {name}: max {np.abs(e).max():.2e} rms {np.sqrt((e ** 2).mean()):.2e} mean {e.mean():+.2e}")
    m.close()
