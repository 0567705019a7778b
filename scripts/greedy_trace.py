import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from arp_amd import synth_policy as S
from arp_amd.train import PolicyConfig, PolicyTrainer
cfg = PolicyConfig(lambda_ret=0.01)
tr = PolicyTrainer(cfg, mode="f16"); tr.set_params(S.policy_params(cfg, seed=0))
enc, act, rtg = S.policy_batch(cfg, 1, seed=5)
tr.set_batch(enc, act, rtg)
for _ in range(30): tr.forward()
tr.close()
