#!/usr/bin/env python3
"""Row N4: where a single-frame reward's latency goes -- per call-site device time (HIP events) beside the wall time of the call."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import clip, synth

for name in ("ViT-B/16", "ViT-B/32"):
    cfg = clip.MODELS[name]
    m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="f16", max_batch=64, n_streams=1).set_text(synth.prompt_tokens(1, 8, seed=2))
    fr = synth.procgen_like_frames(64, seed=3)
    big = m.label(fr)  # 64 frames: the throughput kernels
    scale = float(np.exp(synth.clip_weights(cfg, seed=0)["logit_scale"]))
    for n in (1, 4, 8):
        small = np.concatenate([m.label(fr[i:i + n]) for i in range(0, 16, n)])
        print(f"{name} n={n}: latency path vs the 64-frame pass, cosine difference max {np.abs(small - big[:16]).max() / scale:.2e}", flush=True)
        for _ in range(5):
            m.label(fr[:n])
        t0 = time.perf_counter()
        for _ in range(50):
            m.label(fr[:n])
        wall = (time.perf_counter() - t0) / 50 * 1e3
        m.profile(True); m.profile_reset()
        for _ in range(20):
            m.label(fr[:n])
        s = m.profile_read(); m.profile(False)
        tot = sum(v["ms"] for v in s.values()) / 20
        calls = sum(v["calls"] for v in s.values()) / 20
        top = sorted(((k, v["ms"] / 20, v["calls"] / 20) for k, v in s.items()), key=lambda t: -t[1])[:10]
        print(f"{name} n={n}: wall {wall:.3f} ms per call; device {tot:.3f} ms in {calls:.0f} launches; " + ", ".join(f"{k} {ms*1e3:.0f}us/{c:.0f}" for k, ms, c in top), flush=True)
    m.close()
