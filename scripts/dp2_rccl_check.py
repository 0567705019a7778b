#!/usr/bin/env python3
"""2-rank data-parallel policy step over RCCL, one GPU per rank (needs a node with >= 2 GPUs: RCCL refuses two ranks
on one device with "Duplicate GPU detected", which is what a 1-GPU box reports through arp_last_error()).
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scripts/dp2_rccl_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from arp_amd import synth_policy as S
from arp_amd.train import PolicyConfig, PolicyTrainer
from oracle import arpdt_torch as O

kw = dict(emb=64, depth=2, heads=4, window=3, enc_tokens=5, enc_dim=64, lambda_ret=0.5)
cfg, ocfg = PolicyConfig(**kw), O.PolicyConfig(**kw)
P = S.policy_params(cfg, seed=1)
enc, act, rtg = S.policy_batch(cfg, 4, seed=2)
tr = PolicyTrainer(cfg, mode="f32", device=rank)
# rank 1 starts from DIFFERENT params: broadcast_state (sync_state_fn) must overwrite them with rank 0's
tr.set_params(P if rank == 0 else {k: v + 1.0 for k, v in P.items()})
ids = [PolicyTrainer.new_unique_id() if rank == 0 else None]
dist.broadcast_object_list(ids, src=0)
tr.comm_init(ids[0], world, rank)
tr.broadcast_state()
sl = slice(rank * 2, rank * 2 + 2)
tr.set_batch(enc[sl], act[sl], rtg[sl])
aux = tr.train_step(1e-3)
Pt = {k: torch.from_numpy(v).double() for k, v in P.items()}
sh = [(torch.from_numpy(enc[s]).double(), torch.from_numpy(act[s]).long(), torch.from_numpy(rtg[s]).double()) for s in (slice(0, 2), slice(2, 4))]
st, oaux = O.train_step(O.init_state(Pt), ocfg, sh, lambda t: 1e-3)
got = tr.get_params()
perr = float(np.mean([np.abs(got[k] - st["params"][k].numpy()).mean() for k in P]))
print(f"rank {rank}: loss {aux['loss']:.6f} (oracle pmean {oaux['loss']:.6f}) grad_norm {aux['grad_norm']:.6f} (oracle {oaux['grad_norm']:.6f}) mean param err {perr:.2e}", flush=True)
assert abs(aux["loss"] - oaux["loss"]) < 1e-4 and abs(aux["grad_norm"] - oaux["grad_norm"]) < 1e-4 and perr < 1e-5
tr.close()
dist.barrier()
dist.destroy_process_group()
