#!/usr/bin/env python3
"""Soak: one trainer, prefetched train steps with validation steps and greedy actions in between (three kinds of captured chains on two
batch slots, an uploader thread beside the captures), twice from the same state: the two loss trajectories must agree bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import synth_policy as S
from arp_amd.train import PolicyConfig, TrainState, create_train_step, create_val_step, prefetch_to_device

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = PolicyConfig(lambda_ret=0.01)
P = S.policy_params(cfg, seed=0)
batches = []
for i in range(6):
    enc, act, rtg = S.policy_batch(cfg, 8, seed=40 + i)
    batches.append({"image": {"ob": enc}, "action": act, "rtg": {"ob": rtg}})
one = S.policy_batch(cfg, 1, seed=99)

def run():
    state = TrainState.create(cfg, P, mode="f16")
    fn, vfn = create_train_step(cfg, lambda s: 5e-4, cfg.weight_decay), create_val_step(cfg)
    rng = np.array([0, 7], np.uint32)
    out = []
    gen = (batches[i % 6] for i in range(steps))
    src = gen if os.environ.get("SOAK_NO_PREFETCH") else prefetch_to_device(gen, 2, state.trainer)
    for i, b in enumerate(src):
        state, aux, rng = fn(state, b, rng)
        out.append(aux["loss"])
        if i % 10 == 3 and not os.environ.get("SOAK_NO_VAL"):
            vaux, _ = vfn(state, batches[(i + 1) % 6], rng)
            out.append(vaux["loss"])
        if i % 7 == 5 and not os.environ.get("SOAK_NO_GREEDY"):
            out.append(float(state.trainer.greedy_action(*one)[0]))
    state.trainer.close()
    return np.asarray(out, np.float64)

t0 = time.time()
a, b = run(), run()
assert np.isfinite(a).all() and np.array_equal(a, b), f"trajectories differ at {np.flatnonzero(a != b)[:8]}: {a[np.flatnonzero(a != b)[:4]]} vs {b[np.flatnonzero(a != b)[:4]]}"
print(f"{steps} prefetched steps + validation + greedy actions, twice: {len(a)} values bit-identical, last loss {a[-1]:.4f} ({time.time() - t0:.0f} s)")
