#!/usr/bin/env python3
"""Online (rollout) path, SURVEY row N4: single-frame reward latency and batch-1 greedy_action latency."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import clip, synth, label_reward as LR

for name in ("ViT-B/32", "ViT-B/16"):
    cfg = clip.MODELS[name]
    m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode="bf16", max_batch=64).set_text(synth.prompt_tokens(1, 8, seed=2))
    fr = synth.procgen_like_frames(8, seed=3)
    for n in (1, 8):
        LR.get_torch_clip_reward(m, fr[:n])
        t0 = time.perf_counter()
        for _ in range(50):
            LR.get_torch_clip_reward(m, fr[:n])
        print(f"{name} get_torch_clip_reward n={n}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms", flush=True)
    m.close()
from arp_amd import synth_policy as S
from arp_amd.train import PolicyConfig, PolicyTrainer
pc = PolicyConfig()
tr = PolicyTrainer(pc, mode="bf16")
tr.set_params(S.policy_params(pc, seed=0))
enc, act, rtg = S.policy_batch(pc, 1, seed=1)
tr.greedy_action(enc, act, rtg)
t0 = time.perf_counter()
for _ in range(50):
    tr.greedy_action(enc, act, rtg)
print(f"greedy_action batch 1 (encodings in): {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
tr.close()
