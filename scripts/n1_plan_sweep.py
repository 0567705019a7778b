"""Row N1, f16c encoder: the encoder-inside logit / return error per correction plan (ARP_F16C_PLAN = in_proj, out_proj, fc1, fc2: 0 plain, 1 weight rounding
corrected, 2 + activation rounding), 8 seeds at the real geometry (B = 2) against oracle/m3ae_np -> oracle/arpdt_torch in fp64, behind the default f16 policy
(adapter corrections on) and behind the f32 policy (the encoder's own share).  Test infrastructure: uses the oracle; needs a GPU.
usage: python scripts/n1_plan_sweep.py [n_seeds] [plan,plan,...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from arp_amd import m3ae, synth_policy as S
from arp_amd.train import PolicyConfig, PolicyTrainer
from oracle import arpdt_torch as O, m3ae_np as M

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
plans = sys.argv[2].split(",") if len(sys.argv) > 2 else ["1221", "1211", "1121", "1111", "1210", "0221"]
ecfg, eocfg = m3ae.EncoderConfig(), M.EncConfig()
pcfg, pocfg = PolicyConfig(lambda_ret=0.01), O.PolicyConfig(lambda_ret=0.01)
B, T = 2, pcfg.window
rows = {(p, pm): [] for p in plans for pm in ("f16", "f32")}
for seed in range(n_seeds):
    EP = S.m3ae_params(eocfg, seed=50 + seed)
    P = S.policy_params(pcfg, seed=60 + seed)
    rng = np.random.default_rng(70 + seed)
    frames = S.normalized_frames(B * T, 256, seed=80 + seed).reshape(B, T, 256, 256, 3)
    act = rng.integers(0, pcfg.n_actions, (B, T)).astype(np.int32)
    rtg = rng.random((B, T, 1)).astype(np.float32)
    t0 = time.perf_counter()
    codes = M.forward_representation(EP, eocfg, frames.reshape(-1, 256, 256, 3)).reshape(B, T, ecfg.tokens, ecfg.width)
    ref = O.forward({k: torch.from_numpy(v).double() for k, v in P.items()}, pocfg, torch.from_numpy(np.asarray(codes, np.float64)),
                    torch.from_numpy(act).long(), torch.from_numpy(rtg).double())
    t_or = time.perf_counter() - t0
    for plan in plans:
        os.environ["ARP_F16C_PLAN"] = plan  # read by arp_enc_create
        for pm in ("f16", "f32"):
            enc = m3ae.M3AEEncoder(ecfg, EP, mode="f16c")
            tr = PolicyTrainer(pcfg, mode=pm)
            tr.set_params(P)
            tr.attach_encoder(enc)
            tr.set_batch_images(frames, act, rtg)
            out = tr.forward()
            e = max(float(np.abs(out["action_pred"] - ref["action_pred"].numpy()).max()), float(np.abs(out["return_pred"] - ref["return_pred"].numpy()).max()))
            rows[(plan, pm)].append(e)
            tr.close(); enc.close()
    print(f"seed {seed} (oracle {t_or:.1f} s): " + "  ".join(f"{p}: {rows[(p, 'f16')][-1]:.2e} / {rows[(p, 'f32')][-1]:.2e}" for p in plans), flush=True)
for (p, pm), v in rows.items():
    print(f"plan {p} / policy {pm}: max {max(v):.2e}  median {float(np.median(v)):.2e}  over {len(v)} seeds; outside 1e-3: {sum(e >= 1e-3 for e in v)}   " + " ".join(f"{e:.2e}" for e in v))
