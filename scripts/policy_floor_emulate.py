#!/usr/bin/env python3
"""Per-seed logit / return error of the f16 policy with full adapter corrections, fp64 emulation, under hypotheses about what ELSE is rounded -- to be matched
against the GPU's per-seed numbers (gpurun_out/r6_floor_per_seed.txt): the GPU reads 3.9e-4 / 6.6e-4 where the ideal corrected adapter leaves 1-2e-4."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arp_amd import synth_policy as S  # noqa: E402
from arp_amd.train import PolicyConfig  # noqa: E402
from oracle import arpdt_torch as O  # noqa: E402
import adapter_plan_emulate as E  # noqa: E402

torch.set_num_threads(8)
pcfg, ocfg = PolicyConfig(lambda_ret=0.01), O.PolicyConfig(lambda_ret=0.01)
h = E.h


def run(P, enc, act, rtg, ref, hyp):
    D = pcfg.enc_dim
    x = enc.reshape(-1, D)
    p1 = (1, 1)
    h1 = torch.relu(E.prod(x, P["AdapterMLP_0/Dense_0/kernel"], *p1) + P["AdapterMLP_0/Dense_0/bias"])
    a = torch.relu(E.prod(h1, P["AdapterMLP_0/Dense_1/kernel"], 1, 1) + P["AdapterMLP_0/Dense_1/bias"])
    if "A16" in hyp:
        a = h(a)
    res = torch.sigmoid(P["residual_weight"])
    if "res32" in hyp:
        res = res.float().double()
    y = res * a + (1 - res) * (h(x) if "x16" in hyp else x)
    if "Y16" in hyp:
        y = h(y)
    if "Y32" in hyp:
        y = y.float().double()
    P2 = dict(P)
    P2["residual_weight"] = torch.tensor([-1e4], dtype=torch.float64)
    if "Wi16" in hyp:
        P2["image_text_input/kernel"] = h(P["image_text_input/kernel"])
    if "small16" in hyp:  # every other (small) parameter rounded to binary16
        for k in P2:
            if k not in ("image_text_input/kernel", "residual_weight") and not k.startswith("AdapterMLP"):
                P2[k] = h(P2[k])
    out = O.forward(P2, ocfg, y.reshape(enc.shape), act, rtg)
    return float((out["action_pred"] - ref["action_pred"]).abs().max()), float((out["return_pred"] - ref["return_pred"]).abs().max())


HYPS = [(), ("A16",), ("Y16",), ("Wi16",), ("x16",), ("small16",), ("Y32",)]
rows = {hy: ([], []) for hy in HYPS}
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 16):
    s = 100 + 7 * seed
    P = {k: torch.from_numpy(v).double() for k, v in S.policy_params(pcfg, seed=s).items()}
    enc, act, rtg = S.policy_batch(pcfg, 2, seed=s + 1)
    e, a, r = torch.from_numpy(enc).double(), torch.from_numpy(act).long(), torch.from_numpy(rtg).double()
    ref = O.forward(P, ocfg, e, a, r)
    for hy in HYPS:
        l, rt = run(P, e, a, r, ref, hy)
        rows[hy][0].append(l); rows[hy][1].append(rt)
    print(f"# seed {seed} done", flush=True)
for hy in HYPS:
    print(f"22e + {'+'.join(hy) or 'nothing'}: logits " + " ".join(f"{v:.2e}" for v in rows[hy][0]))
    print(f"22e + {'+'.join(hy) or 'nothing'}: return " + " ".join(f"{v:.2e}" for v in rows[hy][1]))
