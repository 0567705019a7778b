"""Where does the host-fed labelling call lose its 3 ms?  Times (a) the bare H2D of 1024 frames from pageable / pinned memory on an idle GPU,
(b) the same upload while a labelling pass runs on HBM-resident frames (does the copy overlap compute?), (c) arp_clip_label itself."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import clip, synth, _ffi

cfg = clip.MODELS["ViT-B/32"]
m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="f16", max_batch=1024, n_streams=2).set_text(synth.prompt_tokens(1, 8, seed=2))
fr = synth.noise_frames(1024, 256, 256, seed=1000)
dev = clip.DeviceBuffer(fr.nbytes).upload(fr)
up = clip.DeviceBuffer(fr.nbytes)
rew = clip.DeviceBuffer(1024 * 4)

def t_upload(n=5):
    up.upload(fr)
    t = time.perf_counter()
    for _ in range(n):
        up.upload(fr)
    return (time.perf_counter() - t) / n * 1e3

def t_pass(n=5):
    m.label_device_async(dev, 1024, 256, 256, rew); m.sync()
    t = time.perf_counter()
    for _ in range(n):
        m.label_device_async(dev, 1024, 256, 256, rew)
    m.sync()
    return (time.perf_counter() - t) / n * 1e3

for pin in (0, 1):
    if pin:
        m.pin_host(fr)
    a = t_upload()
    b = t_pass()
    # upload from a second thread while the passes run
    res = {}
    def bg():
        res["up"] = t_upload(8)
    th = threading.Thread(target=bg); th.start()
    c = t_pass(8)
    th.join()
    lab = None
    m.label(fr)
    t = time.perf_counter()
    for _ in range(5):
        m.label(fr)
    lab = (time.perf_counter() - t) / 5 * 1e3
    print(f"{'pinned' if pin else 'pageable'}: upload alone {a:.2f} ms ({fr.nbytes / a / 1e6:.1f} GB/s), pass alone {b:.2f} ms, "
          f"together: upload {res['up']:.2f} ms / pass {c:.2f} ms, arp_clip_label {lab:.2f} ms", flush=True)
    if pin:
        m.unpin_host(fr)
m.close()
