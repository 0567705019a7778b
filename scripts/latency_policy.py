#!/usr/bin/env python3
"""Row N4 / P13: greedy_action at batch 1 taken apart -- upload of the [1, T, 257, 768] f32 window, the forward, the download."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import synth_policy as S
from arp_amd.train import PolicyConfig, PolicyTrainer

cfg = PolicyConfig(lambda_ret=0.01)
for mode in ("f16", "f32"):
    tr = PolicyTrainer(cfg, mode=mode)
    tr.set_params(S.policy_params(cfg, seed=0))
    enc, act, rtg = S.policy_batch(cfg, 1, seed=5)
    for _ in range(5):
        tr.greedy_action(enc, act, rtg)
    def t(f, n=200):
        t0 = time.perf_counter()
        for _ in range(n):
            f()
        return (time.perf_counter() - t0) / n * 1e3
    print(f"{mode}: greedy_action {t(lambda: tr.greedy_action(enc, act, rtg)):.3f} ms | set_batch {t(lambda: tr.set_batch(enc, act, rtg)):.3f} ms | "
          f"forward (+ download) {t(lambda: tr.forward()):.3f} ms", flush=True)
    tr.close()
