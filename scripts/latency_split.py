#!/usr/bin/env python3
"""Row N4: a single-frame reward call taken apart -- the device span of the replayed pass (HIP events around it, frames resident),
the wall time of the same with a stream sync per call, and the wall time of the host-fed call (upload + pass + download)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import clip, synth

for name in sys.argv[1:] or ("ViT-B/32", "ViT-B/16"):
    cfg = clip.MODELS[name]
    m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="f16", max_batch=64, n_streams=1).set_text(synth.prompt_tokens(1, 8, seed=2))
    fr = synth.procgen_like_frames(4, seed=3)
    d_fr = clip.DeviceBuffer(fr[:1].nbytes); d_fr.upload(fr[:1])
    d_rw = clip.DeviceBuffer(4)
    e0, e1 = clip.Event(), clip.Event()
    for _ in range(5):
        m.label_device_async(d_fr, 1, 256, 256, d_rw); m.sync()
    reps = 200
    m.record(e0)
    for _ in range(reps):
        m.label_device_async(d_fr, 1, 256, 256, d_rw)
    m.record(e1); m.sync()
    dev = clip.elapsed_ms(e0, e1) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        m.label_device_async(d_fr, 1, 256, 256, d_rw); m.sync()
    sync_wall = (time.perf_counter() - t0) / reps * 1e3
    for _ in range(5):
        m.label(fr[:1])
    t0 = time.perf_counter()
    for _ in range(reps):
        m.label(fr[:1])
    host_wall = (time.perf_counter() - t0) / reps * 1e3
    print(f"{name} single frame: device span {dev:.3f} ms back to back | resident frame + stream sync per call {sync_wall:.3f} ms | "
          f"host frame in, host reward out {host_wall:.3f} ms", flush=True)
    m.close()
