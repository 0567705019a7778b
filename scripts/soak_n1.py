#!/usr/bin/env python3
"""Soak of the encoder-inside training flow (row N1, round 6): frames in, two device slots, an uploader thread that also enqueues the NEXT batch's frozen-encoder pass
(arp_dt_encode_ahead) beside the running step, validation steps on the synchronous slot in between, f16c encoder + f16 policy.  Three runs from the same state:
encode-ahead twice (must agree bit for bit: the extra thread and stream change WHEN the encoder runs, never what it computes) and once with ARP_DT_ENCODE_AHEAD=0
(the encoder at the head of its own step: the same numbers again).  Needs a GPU.   python scripts/soak_n1.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from arp_amd import m3ae, synth_policy as S
from arp_amd.train import PolicyConfig, TrainState, create_train_step, create_val_step, prefetch_to_device

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = int(os.environ.get("SOAK_B", "8"))
cfg, ecfg = PolicyConfig(lambda_ret=0.01), m3ae.EncoderConfig()
P, EP = S.policy_params(cfg, seed=0), S.m3ae_params(ecfg, seed=1)
batches = []
for i in range(5):
    rng = np.random.default_rng(40 + i)
    frames = S.normalized_frames(B * cfg.window, ecfg.img_res, seed=50 + i).reshape(B, cfg.window, ecfg.img_res, ecfg.img_res, 3)
    batches.append({"image": {"ob": frames}, "action": rng.integers(0, cfg.n_actions, (B, cfg.window)).astype(np.int32), "rtg": {"ob": rng.random((B, cfg.window, 1)).astype(np.float32)}})


def run(ahead):
    os.environ["ARP_DT_ENCODE_AHEAD"] = "1" if ahead else "0"
    enc = m3ae.M3AEEncoder(ecfg, EP, mode="f16c")
    state = TrainState.create(cfg, P, mode="f16")
    state.trainer.attach_encoder(enc)
    fn, vfn = create_train_step(cfg, lambda s: 5e-4, cfg.weight_decay), create_val_step(cfg)
    rng = np.array([0, 7], np.uint32)
    out = []
    for i, b in enumerate(prefetch_to_device((batches[i % 5] for i in range(steps)), 2, state.trainer)):
        state, aux, rng = fn(state, b, rng)
        out.append(aux["loss"])
        if i % 10 == 3:
            vaux, _ = vfn(state, batches[(i + 2) % 5], rng)
            out.append(vaux["loss"])
    state.trainer.close(); enc.close()
    return np.asarray(out, np.float64)


t0 = time.time()
a, b, c = run(True), run(True), run(False)
assert np.isfinite(a).all()
assert np.array_equal(a, b), f"two encode-ahead runs differ at {np.flatnonzero(a != b)[:8]}"
assert np.array_equal(a, c), f"encode-ahead differs from the encoder at the head of the step at {np.flatnonzero(a != c)[:8]}"
print(f"{steps} prefetched encoder-inside steps (B = {B}) + validation steps, three runs (encode-ahead x 2, at head x 1): {len(a)} values bit-identical, "
      f"loss {a[0]:.4f} -> {a[-1]:.4f} ({time.time() - t0:.0f} s)")
