#!/usr/bin/env python3
"""Labelling rate against frames per call (HBM-resident frames, one sync per call): where the latency path, the 128x128 kernels and
the 256x256 kernels hand over."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import clip, synth
name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/32"
cfg = clip.MODELS[name]
m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="f16", max_batch=1024, n_streams=2).set_text(synth.prompt_tokens(1, 8, seed=2))
base = synth.procgen_like_frames(64, seed=3)
fr = np.ascontiguousarray(np.tile(base, (16, 1, 1, 1)))
d_fr = clip.DeviceBuffer(fr.nbytes); d_fr.upload(fr)
d_rw = clip.DeviceBuffer(1024 * 4)
for n in (1, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024):
    for _ in range(3):
        m.label_device_async(d_fr, n, 256, 256, d_rw); m.sync()
    reps = 30 if n < 256 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        m.label_device_async(d_fr, n, 256, 256, d_rw); m.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name} {n:5d} frames per call: {dt * 1e3:7.3f} ms  {n / dt / 1e3:7.1f} k frames/s", flush=True)
m.close()
