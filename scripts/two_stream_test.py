#!/usr/bin/env python3
"""Experiment: does running two half-batches on two streams (two handles) beat one full batch?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import _ffi, clip, synth

cfg = clip.VIT_B32
W = synth.clip_weights(cfg, seed=0)
tok = synth.prompt_tokens(1, 8, seed=2)
def mk(n):
    m = clip.ClipLabeller(cfg, W, mode="bf16", max_batch=n).set_text(tok)
    fr = synth.noise_frames(n, seed=1)
    d = clip.DeviceBuffer(fr.nbytes).upload(fr)
    r = clip.DeviceBuffer(n * 4)
    return m, d, r
for split in (1, 2, 4):
    n = 1024 // split
    hs = [mk(n) for _ in range(split)]
    def step():
        for m, d, r in hs:
            m.label_device_async(d, n, 256, 256, r)
    for _ in range(3):
        step()
    for m, _, _ in hs: m.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    for m, _, _ in hs: m.sync()
    dt = (time.perf_counter() - t0) / 10
    print(f"{split} stream(s) x {n} frames: {dt*1e3:.2f} ms per 1024 frames -> {1024/dt:.0f} frames/s", flush=True)
    for m, _, _ in hs: m.close()
