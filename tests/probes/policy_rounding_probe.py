#!/usr/bin/env python3
"""Which operand rounding of the adapter path costs how much on the policy logits?  CPU emulation (fp64 arithmetic with IEEE
half rounding inserted at chosen tensors) at the real geometry, B = 2.  Test infrastructure: uses the oracle.

    python tests/probes/policy_rounding_probe.py [seed]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from arp_amd import synth_policy as S  # noqa: E402
from arp_amd.train import PolicyConfig  # noqa: E402
from oracle import arpdt_torch as O  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg, ocfg = PolicyConfig(lambda_ret=0.01), O.PolicyConfig(lambda_ret=0.01)
P = {k: torch.from_numpy(v).double() for k, v in S.policy_params(cfg, seed=seed).items()}
enc, act, rtg = S.policy_batch(cfg, 2, seed=seed + 1)
enc, act, rtg = torch.from_numpy(enc).double(), torch.from_numpy(act).long(), torch.from_numpy(rtg).double()
ref = O.forward(P, ocfg, enc, act, rtg)["action_pred"]


def h(t, on):
    return t.to(torch.float16).double() if on else t


def run(flags):
    B, T = act.shape
    x32 = enc.reshape(B * T * cfg.enc_tokens, cfg.enc_dim)
    x = h(x32, "X" in flags)
    a = torch.relu(x @ h(P["AdapterMLP_0/Dense_0/kernel"], "W1" in flags) + P["AdapterMLP_0/Dense_0/bias"])
    a = h(a, "H1" in flags)
    a = torch.relu(a @ h(P["AdapterMLP_0/Dense_1/kernel"], "W2" in flags) + P["AdapterMLP_0/Dense_1/bias"])
    a = h(a, "A" in flags)
    res = torch.sigmoid(P["residual_weight"])
    y = res * a + (1 - res) * (x if "Xskip" in flags else x32)
    y = h(y, "Y" in flags)
    # feed y back through the oracle: an identity adapter (res = 0 via a huge negative residual_weight) on enc := y
    P2 = dict(P)
    P2["residual_weight"] = torch.tensor([-1e4], dtype=torch.float64)
    P2["image_text_input/kernel"] = h(P["image_text_input/kernel"], "Wi" in flags)
    out = O.forward(P2, ocfg, y.reshape(enc.shape), act, rtg)["action_pred"]
    return float((out - ref).abs().max())


print("res =", float(torch.sigmoid(P["residual_weight"])))
print("none          ", run(set()))
for f in ("X", "W1", "H1", "W2", "A", "Y", "Wi", "Xskip"):
    print(f"{f:14s}", run({f}))
print("all           ", run({"X", "W1", "H1", "W2", "A", "Y", "Wi", "Xskip"}))
print("all - Xskip   ", run({"X", "W1", "H1", "W2", "A", "Y", "Wi"}))
print("all - Y/Wi/Xsk", run({"X", "W1", "H1", "W2", "A"}))
print("X W1 only     ", run({"X", "W1"}))
