"""S2 seam sweep (host frames in, host rewards out): frames/s of ClipLabeller.label at 1024 frames for a number of upload parts
(ARP_CLIP_HOST_PARTS, read once per process: one child process per setting) x streams, pageable and pinned host memory.
Usage on the GPU box: python scripts/seam_sweep.py            (spawns the children BEFORE touching the GPU)"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(streams):
    import numpy as np
    from arp_amd import clip, synth
    cfg = clip.MODELS["ViT-B/32"]
    m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="f16", max_batch=1024, n_streams=streams).set_text(synth.prompt_tokens(1, 8, seed=2))
    fr = synth.noise_frames(1024, 256, 256, seed=1000)
    out = {}
    for pin in (0, 1):
        if pin:
            m.pin_host(fr)
        for _ in range(3):
            m.label(fr)
        t = time.perf_counter()
        for _ in range(8):
            m.label(fr)
        out["pinned" if pin else "pageable"] = round(1024 * 8 / (time.perf_counter() - t))
        if pin:
            m.unpin_host(fr)
    m.close()
    print(json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]))
    else:
        for streams in (2, 3):
            for parts in (streams, 2 * streams, 3 * streams, 4 * streams):
                env = dict(os.environ, ARP_CLIP_HOST_PARTS=str(parts))
                r = subprocess.run([sys.executable, os.path.abspath(__file__), str(streams)], env=env, capture_output=True, text=True)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                print(f"streams {streams} parts {parts}: {line[-1] if line else r.stderr[-300:]}", flush=True)
