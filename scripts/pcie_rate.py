#!/usr/bin/env python3
"""PCIe-inclusive labelling rate: host uint8 frames in, host rewards out, through arp_clip_label (the S2 seam with host buffers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import clip, synth

cfg = clip.MODELS["ViT-B/32"]
NS = int(os.environ.get("ARP_NS", "2"))
m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), max_batch=1024, n_streams=NS).set_text(synth.prompt_tokens(1, 8, seed=2))
for n in (64, 256, 1024, 2048):
    fr = synth.procgen_like_frames(n, seed=3)
    m.label(fr)
    t0 = time.perf_counter()
    reps = max(2, 2048 // n)
    for _ in range(reps):
        m.label(fr)
    dt = (time.perf_counter() - t0) / reps
    print(f"arp_clip_label host->host: n={n:5d}  {dt*1e3:8.2f} ms  {n/dt:9.0f} frames/s", flush=True)
buf = clip.DeviceBuffer(1024 * 196608)
fr = synth.procgen_like_frames(1024, seed=3)
buf.upload(fr)
t0 = time.perf_counter()
for _ in range(5):
    buf.upload(fr)
dt = (time.perf_counter() - t0) / 5
print(f"pageable H2D of 1024 frames (201 MB): {dt*1e3:.2f} ms = {fr.nbytes/dt/1e9:.1f} GB/s")
m.close()
