"""Row N3 measured end to end: a recorder-style HDF5 file (gzip chunks of one 8-frame row) -> rewards written back.
usage: python scripts/h5_label_rate.py [rows] [traj_len]   (needs a GPU; writes under /tmp)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from arp_amd import clip, h5store, label_reward as L, synth

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
tlen = int(sys.argv[2]) if len(sys.argv) > 2 else 256
F = 8
path = "/tmp/arp_h5_rate.hdf5"
t0 = time.perf_counter()
with h5store.H5Store(path, "w") as f:
    f.attrs["env_name"] = "coinrun"
    done_all = []
    for s in range(0, rows, tlen):
        n = min(tlen, rows - s)
        # Procgen renders 64x64 natively; a 256x256 observation is that picture enlarged: 4x4 blocks of equal pixels, a handful of
        # colours per region (what gzip sees in a real file; synth.procgen_like_frames adds per-pixel noise and does not compress)
        rng = np.random.default_rng(s)
        base = rng.integers(0, 6, (n, 16, 16, 1)).repeat(4, 1).repeat(4, 2) * 40 + rng.integers(0, 3, (n, 64, 64, 3)) * 5
        fr = base.astype(np.uint8).repeat(4, 1).repeat(4, 2)
        idx = np.clip(np.arange(n)[:, None] + np.arange(-F + 1, 1)[None, :], 0, None)
        d = np.zeros((n, F), np.float32); d[-1, -1] = 1
        if s == 0:
            f.create_dataset("ob", data=fr[idx], compression="gzip", chunks=(1, F, 256, 256, 3), maxshape=(None, F, 256, 256, 3))
            f.create_dataset("done", data=d, compression="gzip", chunks=(1, F), maxshape=(None, F))
        else:
            for k, v in (("ob", fr[idx]), ("done", d)):
                ds = f[k]; n0 = ds.shape[0]; ds.resize(n0 + n, axis=0); ds[n0:] = v
print(f"wrote {rows} rows ({os.path.getsize(path) / 1e6:.0f} MB on disk, {rows * F * 196608 / 1e6:.0f} MB raw) in {time.perf_counter() - t0:.1f} s", flush=True)

with h5store.H5Store(path, "r") as f:
    d = f["ob"]
    n = min(rows, 512)
    t = time.perf_counter(); a = d[0:n, -1]; t_ref = time.perf_counter() - t
    print(f"reference pattern g['ob'][traj, -1] (every row's chunk inflated by the library, one thread): {n / t_ref:.0f} frames/s")
    for th in (1, 8, 32, 64):
        t = time.perf_counter(); b = d.read_last_frames(0, n, threads=th, stacked=False, native=False); t1 = time.perf_counter() - t
        t = time.perf_counter(); c = d.read_last_frames(0, min(n, tlen), threads=th, native=False); t2 = time.perf_counter() - t
        assert np.array_equal(a, b) and np.array_equal(a[: len(c)], c)
        t = time.perf_counter(); e = d.read_last_frames(0, min(n, tlen), native_threads=th); t3 = time.perf_counter() - t
        assert np.array_equal(a[: len(e)], e)
        print(f"threads {th:2d}: Python pool, per-row chunks {n / t1:.0f} frames/s; one chunk per {F} rows {len(c) / t2:.0f} frames/s; "
              f"C++ threads (arp_h5_inflate_last_frames), one chunk per {F} rows {len(e) / t3:.0f} frames/s")

cfg = clip.MODELS["ViT-B/32"]
m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), device=0)
tok = synth.prompt_tokens(1, 8, seed=2)
for rep in range(2):
    t = time.perf_counter()
    L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=path, clip_model=m, tokens=tok)
    dt = time.perf_counter() - t
    print(f"label_reward(data_path=...) end to end, pass {rep}: {rows / dt:.0f} frames/s ({dt:.2f} s for {rows} rows)")
with h5store.H5Store(path, "r") as f:
    r = f["ob_clip_reward"][...]
    direct = m.label(f["ob"][0:64, -1])
    assert np.array_equal(r[:64, -1], direct), "file rewards differ from labelling the same frames directly"
    print("rewards in the file == rewards of the same frames labelled directly:", True, r.shape)
m.close()
os.remove(path)
