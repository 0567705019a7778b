#!/usr/bin/env python3
"""Where does the f16 policy's logit / return error come from once the adapter's operand roundings are corrected?  CPU emulation of the adapter
(scripts/adapter_plan_emulate.py) says full corrections leave 1-2e-4; the GPU reads 3.9e-4 (logits) / 6.6e-4 (return) over 16 seeds.  Bisect on the GPU:
the same seeds with the fused transformer on the f32 MFMA (ARP_PF_X3=0), image_text_input on the f32-MFMA GEMM (ARP_DT_ITI_X3=0), both.  Needs a GPU.

    python scripts/policy_floor_probe.py [n_seeds]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arp_amd import synth_policy as S  # noqa: E402
from arp_amd.train import PolicyConfig, PolicyTrainer  # noqa: E402
from oracle import arpdt_torch as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg, ocfg = PolicyConfig(lambda_ret=0.01), O.PolicyConfig(lambda_ret=0.01)
cases = []
for seed in range(n):
    s = 100 + 7 * seed
    P = S.policy_params(cfg, seed=s)
    enc, act, rtg = S.policy_batch(cfg, 2, seed=s + 1)
    ref = O.forward({k: torch.from_numpy(v).double() for k, v in P.items()}, ocfg, torch.from_numpy(enc).double(), torch.from_numpy(act).long(), torch.from_numpy(rtg).double())
    cases.append((P, enc, act, rtg, ref["action_pred"].numpy(), ref["return_pred"].numpy()))
    print(f"# oracle seed {seed}: |return_pred| max {np.abs(cases[-1][5]).max():.3f}, |logits| max {np.abs(cases[-1][4]).max():.3f}", flush=True)
CONFIGS = [("f16 default", "f16", False, {}), ("f16 + corrections", "f16", True, {}), ("f16 + corrections, transformer on f32 MFMA", "f16", True, {"ARP_PF_X3": "0"}),
           ("f16 + corrections, iti on f32 MFMA", "f16", True, {"ARP_DT_ITI_X3": "0"}), ("f16 + corrections, both", "f16", True, {"ARP_PF_X3": "0", "ARP_DT_ITI_X3": "0"}),
           ("f16 + corrections, per-op transformer", "f16", True, {"ARP_DT_FUSED": "0"}), ("f32", "f32", False, {})]
for name, mode, corr, env in CONFIGS:
    for k in ("ARP_PF_X3", "ARP_DT_ITI_X3", "ARP_DT_FUSED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    tr = PolicyTrainer(cfg, mode=mode, adapter_corrections=corr)
    el, er = [], []
    for P, enc, act, rtg, rl, rr in cases:
        tr.set_params(P)
        tr.set_batch(enc, act, rtg)
        out = tr.forward()
        el.append(float(np.abs(out["action_pred"] - rl).max()))
        er.append(float(np.abs(out["return_pred"] - rr).max()))
    tr.close()
    print(f"{name:48s} logits max {max(el):.2e} median {np.median(el):.2e} | return max {max(er):.2e} median {np.median(er):.2e}", flush=True)
