"""Oracle leg of bench.py: time the torch-CPU port of the labelling pass on this host's cores.

Run as a child process (``python -m oracle.cpu_baseline``) so that torch's bundled ROCm libraries
never share a process with libarp_hip.so.  Prints one JSON object.
"""
import argparse
import json
import os
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-B/32")
    ap.add_argument("--target-seconds", type=float, default=15.0)
    ap.add_argument("--max-frames", type=int, default=1024)
    a = ap.parse_args()
    import numpy as np
    import torch

    from arp_amd import synth
    from oracle import clip_np, clip_torch

    ncpu = os.cpu_count() or 1
    cfg = clip_np.ClipConfig(patch=32 if a.model == "ViT-B/32" else 16)
    W = clip_torch.to_torch(synth.clip_weights(cfg, seed=0))
    tok = synth.prompt_tokens(1, 8, seed=2)
    frames = synth.noise_frames(a.max_frames, seed=0)
    txt = clip_torch.encode_text(W, cfg, tok)  # cached, as the GPU path caches it (excluded from both timings)
    bs = 64  # the reference sends one trajectory per call; 64-frame batches keep the CPU GEMMs efficient
    # torch's intra-op pool does not scale to hundreds of threads on these shapes: probe a few pool
    # sizes on one batch each (after an untimed warm-up) and keep the fastest
    torch.set_num_threads(min(ncpu, 16))
    clip_torch.compute_reward(W, cfg, frames[:8], tok, text_feat=txt)
    best = None
    for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        clip_torch.compute_reward(W, cfg, frames[:bs], tok, text_feat=txt)
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (th, dt)
    cores = best[0]
    torch.set_num_threads(cores)
    n = int(max(bs, min(a.max_frames, bs * a.target_seconds / max(best[1], 1e-3)))) // bs * bs
    t0 = time.perf_counter()
    out = [clip_torch.compute_reward(W, cfg, frames[i : i + bs], tok, text_feat=txt) for i in range(0, n, bs)]
    dt = time.perf_counter() - t0
    r = np.concatenate(out)
    print(json.dumps({"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
                      "sample": f"{n} of the bench's synthetic 256x256x3 frames, {a.model}, torch-CPU fp32 port incl. the "
                                f"per-frame PIL bicubic transform (label_reward.py:134), batches of {bs}, {cores} torch threads "
                                f"(fastest of 8/16/32/64 on a {ncpu}-cpu host), {dt:.1f} s",
                      "checksum": float(np.abs(r).sum())}))


if __name__ == "__main__":
    main()
