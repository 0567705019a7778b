#!/usr/bin/env python3
"""The policy's logit / return error per adapter-correction plan (ARP_DT_ADAPTER_PLAN), on the GPU: 16 seeds on N(0,1) encodings (the seeds of
tests/test_policy_gpu.py::test_f16_full_geometry_logits_over_sixteen_seeds) and 8 seeds behind f32 ENCODER outputs (the seeds of scripts/n1_parity_probe.py; the
f32 encoder runs on the GPU, 5e-6 from the fp64 oracle), against oracle/arpdt_torch in fp64.  VERDICT r5 next #2's bars: <= 7.5e-4 and <= 1e-3.

    python scripts/adapter_plan_gpu.py [plan ...]      e.g. 22e 22h 12h
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arp_amd import m3ae, synth_policy as S  # noqa: E402
from arp_amd.train import PolicyConfig, PolicyTrainer  # noqa: E402
from oracle import arpdt_torch as O  # noqa: E402

plans = sys.argv[1:] or ["off", "22e", "22h", "12h", "21h", "11h", "12e"]
cfg, ocfg = PolicyConfig(lambda_ret=0.01), O.PolicyConfig(lambda_ret=0.01)


def oracle(P, enc, act, rtg):
    r = O.forward({k: torch.from_numpy(v).double() for k, v in P.items()}, ocfg, torch.from_numpy(np.asarray(enc, np.float64)), torch.from_numpy(act).long(), torch.from_numpy(rtg).double())
    return r["action_pred"].numpy(), r["return_pred"].numpy()


cases = {"N(0,1)": [], "encoder": []}
for seed in range(16):
    s = 100 + 7 * seed
    P = S.policy_params(cfg, seed=s)
    enc, act, rtg = S.policy_batch(cfg, 2, seed=s + 1)
    cases["N(0,1)"].append((P, enc, act, rtg) + oracle(P, enc, act, rtg))
ecfg = m3ae.EncoderConfig()
for seed in range(8):
    EP = S.m3ae_params(ecfg, seed=50 + seed)
    P = S.policy_params(cfg, seed=60 + seed)
    rng = np.random.default_rng(70 + seed)
    frames = S.normalized_frames(2 * cfg.window, 256, seed=80 + seed)
    act = rng.integers(0, cfg.n_actions, (2, cfg.window)).astype(np.int32)
    rtg = rng.random((2, cfg.window, 1)).astype(np.float32)
    e = m3ae.M3AEEncoder(ecfg, EP, mode="f32")
    enc = e.forward_representation(frames).reshape(2, cfg.window, ecfg.tokens, ecfg.width)
    e.close()
    cases["encoder"].append((P, enc, act, rtg) + oracle(P, enc, act, rtg))
print(f"# {len(cases['N(0,1)'])} + {len(cases['encoder'])} cases ready", flush=True)
for plan in plans:
    os.environ.pop("ARP_DT_ADAPTER_PLAN", None)
    if plan != "off":
        os.environ["ARP_DT_ADAPTER_PLAN"] = plan
    tr = PolicyTrainer(cfg, mode="f16", adapter_corrections=plan != "off")
    line = []
    for kind in ("N(0,1)", "encoder"):
        el, er = [], []
        for P, enc, act, rtg, rl, rr in cases[kind]:
            tr.set_params(P)
            tr.set_batch(enc, act, rtg)
            out = tr.forward()
            el.append(float(np.abs(out["action_pred"] - rl).max()))
            er.append(float(np.abs(out["return_pred"] - rr).max()))
        both = [max(a, b) for a, b in zip(el, er)]
        if os.environ.get("PER_SEED"):
            print(f"  plan {plan} {kind} logits per seed: " + " ".join(f"{v:.2e}" for v in el), flush=True)
            print(f"  plan {plan} {kind} return per seed: " + " ".join(f"{v:.2e}" for v in er), flush=True)
        line.append(f"{kind}: logits max {max(el):.2e} return max {max(er):.2e} | per-seed max {max(both):.2e} median {np.median(both):.2e} outside 1e-3: {sum(v >= 1e-3 for v in both)}/{len(both)}")
    tr.close()
    print(f"plan {plan:4s} " + "  ||  ".join(line), flush=True)
