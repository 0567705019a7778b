"""Row N1 in 16-bit (VERDICT r3 next #1a): where does the error of the policy logits come from when the frozen encoder runs in f16?
For each seed: frames -> oracle/m3ae_np -> oracle/arpdt_torch in fp64 = reference; then every (encoder mode, policy mode) pair at the real geometry, B = 2.
usage: python scripts/n1_parity_probe.py [n_seeds] [extra env switches are read by the library as usual]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from arp_amd import m3ae, synth_policy as S
from arp_amd.train import PolicyConfig, PolicyTrainer
from oracle import arpdt_torch as O, m3ae_np as M

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pairs = [p.split(":") for p in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["f16:f16", "f16:f32", "f32:f16"])]
ecfg, eocfg = m3ae.EncoderConfig(), M.EncConfig()
pcfg, pocfg = PolicyConfig(lambda_ret=0.01), O.PolicyConfig(lambda_ret=0.01)
B, T = 2, pcfg.window
rows = {tuple(p): [] for p in pairs}
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
    for em, pm in pairs:
        enc = m3ae.M3AEEncoder(ecfg, EP, mode=em)
        tr = PolicyTrainer(pcfg, mode=pm.replace("+c", ""), adapter_corrections=pm.endswith("+c"))  # "f16+c": adapter products corrected on the fp4 MFMA
        tr.set_params(P)
        tr.attach_encoder(enc)
        tr.set_batch_images(frames, act, rtg)
        out = tr.forward()
        e = max(float(np.abs(out["action_pred"] - ref["action_pred"].numpy()).max()), float(np.abs(out["return_pred"] - ref["return_pred"].numpy()).max()))
        rows[(em, pm)].append(e)
        tr.close(); enc.close()
    print(f"seed {seed} (oracle {t_or:.1f} s): " + "  ".join(f"enc {em} / policy {pm}: {rows[(em, pm)][-1]:.2e}" for em, pm in pairs), flush=True)
for (em, pm), v in rows.items():
    print(f"enc {em} / policy {pm}: max {max(v):.2e}  median {float(np.median(v)):.2e}  over {len(v)} seeds; outside 1e-3: {sum(e >= 1e-3 for e in v)}")
