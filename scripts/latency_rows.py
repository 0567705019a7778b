#!/usr/bin/env python3
"""Row N4: where does the latency path stop paying?  Wall time of host-fed calls of n frames with ARP_SKINNY_ROWS = 1 (off) and 1024."""
import os, sys, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from arp_amd import clip, synth
    out = {}
    for name in ("ViT-B/32", "ViT-B/16"):
        cfg = clip.MODELS[name]
        m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="f16", max_batch=64, n_streams=1).set_text(synth.prompt_tokens(1, 8, seed=2))
        fr = synth.procgen_like_frames(32, seed=3)
        for n in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20):
            if n * cfg.tokens > 1024:
                continue
            for _ in range(4):
                m.label(fr[:n])
            t0 = time.perf_counter()
            for _ in range(60):
                m.label(fr[:n])
            out[f"{name} n={n}"] = (time.perf_counter() - t0) / 60 * 1e3
        m.close()
    print(json.dumps(out))
    sys.exit(0)
res = {}
for rows in ("1", "1024"):
    env = dict(os.environ, ARP_SKINNY_ROWS=rows)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
    res[rows] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
for k in res["1"]:
    a, b = res["1"][k], res["1024"][k]
    print(f"{k:16s} rows {int(k.split('=')[1]) * (50 if 'B/32' in k else 197):5d}: throughput kernels {a:.3f} ms | latency path {b:.3f} ms | {'+' if b < a else '-'}{abs(a - b) / a * 100:.0f} %")
