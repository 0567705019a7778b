"""Stage times of label_reward on an HDF5 file (open / bounds / read / label / write); needs a GPU."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from arp_amd import clip, h5store, label_reward as L, synth
rows, tlen, F = 8192, 512, 8
path = "/tmp/p.hdf5"
with h5store.H5Store(path, "w") as f:
    for s in range(0, rows, tlen):
        n = tlen
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
cfg = clip.MODELS["ViT-B/32"]
m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), device=0).set_text(synth.prompt_tokens(1, 8, seed=2))
for rep in range(2):
    t0 = time.perf_counter()
    st = h5store.H5Store(path, "a")
    t1 = time.perf_counter()
    b = L.trajectory_bounds(st)
    t2 = time.perf_counter()
    ds = st["ob"]
    groups = [[(i * 512, i * 512 + 512), (i * 512 + 512, i * 512 + 1024)] for i in range(0, 16, 2)]
    tr = tl = 0
    for g in groups:
        t = time.perf_counter(); fr = ds.read_last_frames_spans(g); tr += time.perf_counter() - t
        t = time.perf_counter(); r = m.label(fr); tl += time.perf_counter() - t
    t3 = time.perf_counter()
    if os.environ.get("ARP_CPROFILE") == "1" and rep == 1:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        res = L.label_store(st, m, L.make_compute_reward("clip"))
        pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(12)
    else:
        res = L.label_store(st, m, L.make_compute_reward("clip"))
    t4 = time.perf_counter()
    L.write_results(st, res, True, F)
    t5 = time.perf_counter()
    st.close()
    t6 = time.perf_counter()
    print(f"open {t1-t0:.3f} bounds {t2-t1:.3f} serial read {tr:.3f} serial label {tl:.3f} label_store {t4-t3:.3f} write {t5-t4:.3f} close {t6-t5:.3f}")
