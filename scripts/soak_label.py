#!/usr/bin/env python3
"""Soak: one handle, calls of every kind interleaved for a while -- single frames (captured graphs, pinned staging), a few frames,
full batches (two streams), the asynchronous pair, a prompt change in between -- every result compared with the first result of the same
call.  Catches stale captures, workspace growth under live graphs, slot reuse."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import clip, synth

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 45.0
name = sys.argv[2] if len(sys.argv) > 2 else "ViT-B/32"
cfg = clip.MODELS[name]
m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="f16", max_batch=1024, n_streams=2)
toks = [synth.prompt_tokens(3, [8, 6, 9], seed=2), synth.prompt_tokens(2, [5, 7], seed=9)]  # several prompts each: the online list-of-prompts (mean) branch is in the mix
base = synth.procgen_like_frames(64, seed=3)
big = np.ascontiguousarray(np.tile(base, (16, 1, 1, 1)))
ref = {}
def check(key, val):
    if key not in ref:
        ref[key] = val.copy()
    elif not np.array_equal(ref[key], val):
        raise SystemExit(f"MISMATCH at {key}: max diff {np.abs(ref[key] - val).max()}")
rng = np.random.default_rng(0)
t0, it, prompt = time.time(), 0, 0
m.set_text(toks[prompt])
while time.time() - t0 < seconds:
    kind = rng.integers(0, 9)
    if kind == 0:
        i = int(rng.integers(0, 64)); check((prompt, "one", i), m.label(base[i:i + 1]))
    elif kind == 1:
        n = int(rng.integers(2, 21)); check((prompt, "few", n), m.label(base[:n]))
    elif kind == 2:
        n = int(rng.choice([21, 64, 130, 300])); check((prompt, "mid", n), m.label(big[:n]))
    elif kind == 3:
        check((prompt, "big"), m.label(big))
    elif kind == 4:
        m.label_submit(0, big); m.label_submit(1, big[:512])
        if rng.integers(0, 2):  # other calls while both slots are in flight
            check((prompt, "one", 5), m.label(base[5:6])); check((prompt, "mid", 64), m.label(big[:64]))
        check((prompt, "big"), m.label_collect(0)); check((prompt, "half"), m.label_collect(1))
    elif kind == 5:
        prompt ^= 1; m.set_text(toks[prompt])
    elif kind == 7:  # round 4: the rollout loop's other reward functions on the same handle (captured passes keyed by the prompt reduction)
        from arp_amd import label_reward as L
        i = int(rng.integers(0, 64))
        check((prompt, "mean", i), L.get_torch_clip_reward(m, base[i], ["x"] * len(toks[prompt])))
        check((prompt, "one", i), L.get_torch_clip_reward(m, base[i], "x"))
    elif kind == 8:
        from arp_amd import label_reward as L
        i = int(rng.integers(0, 8))
        check(("goal", i), np.float64(L.get_torch_clip_goal_conditioned_reward(m, base[i], base[63])).reshape(1))
    else:
        check((prompt, "crop", 1), m.label(base[:1], use_crop=True)); check((prompt, "enc", 3), m.encode_image(base[:3]))
    it += 1
print(f"{name}: {it} interleaved calls in {time.time() - t0:.0f} s, {len(ref)} distinct calls, every repeat bit-identical")
m.close()
