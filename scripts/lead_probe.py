import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from arp_amd import clip, synth
cfg = clip.MODELS["ViT-B/32"]
n = 1024
m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="f16", max_batch=n, n_streams=2).set_text(synth.prompt_tokens(1, 8, seed=2))
base = synth.procgen_like_frames(64, seed=3)
fr = np.ascontiguousarray(np.tile(base, (n // 64, 1, 1, 1)))
r0 = None
for _ in range(3):
    r = m.label(fr)
t0 = time.perf_counter()
for _ in range(10):
    r = m.label(fr)
print(f"ARP_CLIP_LEAD={os.environ.get('ARP_CLIP_LEAD','0')}: {(time.perf_counter()-t0)/10*1e3:.2f} ms per synchronous 1024-frame call; checksum {float(r.sum()):.6f}")
m.close()
