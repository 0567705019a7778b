#!/usr/bin/env python3
"""Where does a pipelined host-fed labelling call (arp_clip_label_submit / _collect) lose time against resident frames?
Per iteration: time inside submit(), inside collect(), and the whole; for pageable and pinned sources; beside the same loop on
frames that already sit in HBM (label_device_async per call, one sync per call, and back to back)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import clip, synth

cfg = clip.MODELS["ViT-B/32"]
n = 1024
m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="f16", max_batch=n, n_streams=2).set_text(synth.prompt_tokens(1, 8, seed=2))
base = synth.procgen_like_frames(64, seed=3)
bufs = [np.ascontiguousarray(np.tile(base, (n // 64, 1, 1, 1))) for _ in range(3)]
d_fr = clip.DeviceBuffer(bufs[0].nbytes); d_fr.upload(bufs[0])
d_rw = clip.DeviceBuffer(n * 4)
for _ in range(3):
    m.label_device_async(d_fr, n, 256, 256, d_rw); m.sync()
iters = 20
t0 = time.perf_counter()
for _ in range(iters):
    m.label_device_async(d_fr, n, 256, 256, d_rw)
m.sync()
print(f"resident, back to back: {(time.perf_counter() - t0) / iters * 1e3:.2f} ms per call")
t0 = time.perf_counter()
for _ in range(iters):
    m.label_device_async(d_fr, n, 256, 256, d_rw); m.sync()
print(f"resident, sync per call: {(time.perf_counter() - t0) / iters * 1e3:.2f} ms per call")
for pinned in (False, True):
    if pinned:
        for b in bufs:
            clip.ClipLabeller.pin_host(b)
    m.label_submit(0, bufs[0]); m.label_collect(0)
    ts, tc = [], []
    t_all = time.perf_counter()
    m.label_submit(0, bufs[0])
    for i in range(1, iters + 1):
        a = time.perf_counter()
        if i < iters:
            m.label_submit(i % 2, bufs[i % 3])
        b = time.perf_counter()
        m.label_collect((i - 1) % 2)
        c = time.perf_counter()
        ts.append(b - a); tc.append(c - b)
    tot = (time.perf_counter() - t_all) / iters * 1e3
    print(f"{'pinned' if pinned else 'pageable'} submit/collect: {tot:.2f} ms per call ({n / tot:.1f} k frames/s); inside submit {np.median(ts) * 1e3:.2f} ms, "
          f"inside collect {np.median(tc) * 1e3:.2f} ms", flush=True)
    t0 = time.perf_counter()
    for i in range(iters):
        m.label(bufs[i % 3])
    print(f"{'pinned' if pinned else 'pageable'} synchronous arp_clip_label: {(time.perf_counter() - t0) / iters * 1e3:.2f} ms per call")
m.close()
