"""bench.py --path h5's three passes with ARP_LABEL_TIMING=1: where the FIRST pass over a file (27.3 k frames/s in BENCH_r05) differs from the later ones
(39.6 k).  Needs a GPU.  python scripts/h5_first_pass.py [rows]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, ".")
os.environ["ARP_LABEL_TIMING"] = "1"
import torch  # noqa: F401
from arp_amd import clip, h5store, label_reward as L, synth
rows, tlen, F = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 256, 8
path = os.path.join(tempfile.gettempdir(), f"arp_h5_first_{os.getpid()}.hdf5")
with h5store.H5Store(path, "w") as f:
    f.attrs["env_name"] = "coinrun"
    for s0 in range(0, rows, tlen):
        n = min(tlen, rows - s0)
        rng = np.random.default_rng(s0)
        base = rng.integers(0, 6, (n, 16, 16, 1)).repeat(4, 1).repeat(4, 2) * 40 + rng.integers(0, 3, (n, 64, 64, 3)) * 5
        fr = base.astype(np.uint8).repeat(4, 1).repeat(4, 2)
        idx = np.clip(np.arange(n)[:, None] + np.arange(-F + 1, 1)[None, :], 0, None)
        d = np.zeros((n, F), np.float32); d[-1, -1] = 1
        if s0 == 0:
            f.create_dataset("ob", data=fr[idx], compression="gzip", chunks=(1, F, 256, 256, 3), maxshape=(None, F, 256, 256, 3))
            f.create_dataset("done", data=d, compression="gzip", chunks=(1, F), maxshape=(None, F))
        else:
            for k, v in (("ob", fr[idx]), ("done", d)):
                ds = f[k]; n0 = ds.shape[0]; ds.resize(n0 + n, axis=0); ds[n0:] = v
cfg = clip.MODELS["ViT-B/32"]
m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="f16", device=0)
tok = synth.prompt_tokens(1, 8, seed=2)
keys = ("ob_clip_reward", "ob_clip_pos_rtg")
for rep in range(3):  # pass 0 = the process's first labelling call (lazy GPU init); its datasets are deleted: pass 1 is again a first pass over the file
    t = time.perf_counter()
    L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=path, clip_model=m, tokens=tok)
    dt = time.perf_counter() - t
    print(f"== pass {rep}: {dt * 1e3:.1f} ms, {rows / dt:.0f} frames/s", flush=True)
    if rep == 0:
        with h5store.H5Store(path, "a") as f:
            for k in keys:
                del f[k]
for thr in (16, 32, 64, 128):  # inflate threads of arp_h5_inflate_last_frames (ARP_H5_THREADS; default 32)
    os.environ["ARP_H5_THREADS"] = str(thr)
    for kind in ("file-first", "later"):
        if kind == "file-first":
            with h5store.H5Store(path, "a") as f:
                for k in keys:
                    del f[k]
        ts = []
        for _ in range(1 if kind == "file-first" else 3):
            t = time.perf_counter()
            L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=path, clip_model=m, tokens=tok)
            ts.append(time.perf_counter() - t)
        print(f"== threads {thr:3d} {kind}: {min(ts) * 1e3:.1f} ms, {rows / min(ts):.0f} frames/s", flush=True)
m.close()
os.remove(path)
