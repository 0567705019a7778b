"""Real two-rank RCCL runs of the two data-parallel train steps (one process per GPU, gloo control plane).  SKIPPED on a box with fewer
than two GPUs -- which is every box this repository has been built on so far: the single-GPU evidence for the collective path is
tests/test_policy_gpu.py::test_bucketed_overlapped_allreduce_equals_serial (world = 1 through RCCL).  What these tests assert, the day two
GPUs are visible: after N data-parallel steps on the two halves of a batch, both ranks hold the same parameters, and they equal
a single-rank run on the whole batch to float tolerance (the mean of the shard gradients is the full-batch gradient for the policy's
per-sample losses)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

# every test that launches RCCL carries the gpu mark itself: the worker functions below are ALSO driven on CPU, with the trainer replaced by
# _DryTrainer, from tests/test_dist_gloo.py (so that import errors, argument drift and the queue plumbing of these never-yet-run tests are
# caught on a box without two GPUs)


class _DryTrainer:
    """Stands where PolicyTrainer / FinetuneTrainer stand when a worker is driven with dry=True (no GPU): keeps the parameter dict it is
    given, implements the library's collective contract on gloo -- broadcast_state = rank 0's parameters everywhere, a step = all-reduce(sum)
    of a 'gradient' that is a fixed function of the shard it was handed, 1/world folded into the update -- and records every call."""

    def __init__(self, cfg, mode="f32", device=0):
        self.cfg, self.mode, self.device = cfg, mode, device
        self.calls, self.world, self.rank, self.P, self.g = [("create", mode, device)], 1, 0, {}, 0.0

    def set_params(self, P):
        self.P = {k: np.array(v, np.float64) for k, v in P.items()}

    def get_params(self):
        return {k: v.astype(np.float32) for k, v in self.P.items()}

    def get_grads(self):
        return {k: np.full(v.shape, self.g, np.float32) for k, v in self.P.items()}

    def new_unique_id(self):
        self.calls.append("new_unique_id")
        return bytes([9]) * 128

    def comm_init(self, uid, world, rank):
        self.calls.append(("comm_init", bytes(uid), world, rank))
        self.world, self.rank = world, rank

    def broadcast_state(self):
        import torch
        import torch.distributed as dist
        self.calls.append("broadcast_state")
        for k in sorted(self.P):
            t = torch.from_numpy(self.P[k])
            dist.broadcast(t, src=0)

    def comm_info(self):  # what ncclCommCount / ncclCommUserRank would say: here, what gloo says
        import torch.distributed as dist
        return {"nranks": dist.get_world_size(), "rank": dist.get_rank(), "device": self.device, "rccl_version": 0, "has_comm": self.world > 1}

    def comm_selfcheck(self):
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(self.rank + 1)], dtype=torch.float64)
        dist.all_reduce(t)
        return float(t)

    def set_batch(self, *arrays):
        self.calls.append(("set_batch",) + tuple(np.asarray(a).shape for a in arrays))
        self.batch = arrays

    def train_step(self, lr):
        import torch
        import torch.distributed as dist
        g = torch.tensor([sum(float(np.asarray(a, np.float64).sum()) for a in self.batch)], dtype=torch.float64)
        dist.all_reduce(g)
        self.g = float(g) / self.world
        for k in self.P:
            self.P[k] -= lr * self.g * 1e-3
        return {"loss": self.g}

    def close(self):
        self.calls.append("close")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _need_two_gpus():
    from arp_amd import _ffi
    if _ffi.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")


def _policy_worker(rank, world, port, q, overlap, mode="f32", dry=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ARP_DT_OVERLAP=str(overlap), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch  # noqa: F401  -- before arp_amd: one HIP runtime per process
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arp_amd import synth_policy as S, train
    from arp_amd.train import PolicyConfig, PolicyTrainer
    # emb 128 / enc_dim 128 with the adapter on: in the 16-bit modes the TN weight-gradient kernels and the fused dY pass run, i.e. stage 2 of
    # the backward really computes beside bucket 1's in-place reduce on the communication stream (the f32 case leaves stage 2 empty)
    cfg = PolicyConfig(emb=128, depth=2, heads=8, window=4, enc_tokens=9, enc_dim=128, lambda_ret=0.01)
    tr = (_DryTrainer if dry else PolicyTrainer)(cfg, mode=mode, device=rank)
    tr.set_params(S.policy_params(cfg, seed=1 + rank))  # rank 1 starts elsewhere: sync_state_fn must overwrite it
    dp = train.DataParallel(tr, rank, world, train.torch_object_broadcast(dist))
    cert = dp.certify()  # ncclCommCount / ncclCommUserRank + an all-reduce of rank + 1: the communicator really spans `world` ranks
    assert cert["ok"] and cert["nranks"] == world and cert["allreduce_selfcheck"] == world * (world + 1) / 2, cert
    enc, act, rtg = S.policy_batch(cfg, 8, seed=5)
    batch = {"image": {"ob": enc}, "action": act, "rtg": {"ob": rtg}}
    auxs = [dp.train_step(batch, 1e-3) for _ in range(4)]
    q.put((rank, tr.get_params(), auxs, tr.get_grads()) + ((tr.calls,) if dry else ()))
    dist.barrier()
    tr.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("overlap,mode", [(1, "f32"), (0, "f32"), (1, "f16"), (0, "f16")])
def test_policy_two_ranks_equal_single_rank(gpu_lib, overlap, mode):
    _need_two_gpus()
    from arp_amd import synth_policy as S
    from arp_amd.train import PolicyConfig, PolicyTrainer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_policy_worker, args=(r, 2, port, q, overlap, mode)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, p0, a0, g0), (_, p1, a1, g1) = res
    assert all(np.array_equal(p0[k], p1[k]) for k in p0), "the two ranks diverged"
    assert [a["loss"] for a in a0] == [a["loss"] for a in a1]
    assert all(np.array_equal(g0[k], g1[k]) for k in g0)  # the getter returns the rank MEAN on both
    cfg = PolicyConfig(emb=128, depth=2, heads=8, window=4, enc_tokens=9, enc_dim=128, lambda_ret=0.01)
    tr = PolicyTrainer(cfg, mode=mode, device=0)
    tr.set_params(S.policy_params(cfg, seed=1))
    enc, act, rtg = S.policy_batch(cfg, 8, seed=5)
    ref = []
    for _ in range(4):
        tr.set_batch(enc, act, rtg)
        ref.append(tr.train_step(1e-3))
    want = tr.get_params()
    tr.close()
    # f16: the two half-batch backward passes round their 16-bit operands differently from one full-batch pass
    assert max(float(np.abs(want[k] - p0[k]).max()) for k in want) < (2e-5 if mode == "f32" else 2e-3)
    assert max(abs(a["loss"] - b["loss"]) for a, b in zip(ref, a0)) < (1e-5 if mode == "f32" else 2e-3)


def _ft_worker(rank, world, port, q, mode, dry=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch  # noqa: F401
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arp_amd import finetune as FT, train
    cfg = FT.FinetuneConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64, n_actions=5)
    tr = (_DryTrainer if dry else FT.FinetuneTrainer)(cfg, mode=mode, device=rank)
    tr.set_params(FT.synth_params(cfg, seed=1 + rank))
    dp = FT.DataParallel(tr, rank, world, train.torch_object_broadcast(dist))
    cert = dp.certify()
    assert cert["ok"] and cert["nranks"] == world and cert["allreduce_selfcheck"] == world * (world + 1) / 2, cert
    b = FT.synth_batch(cfg, 6, seed=2)
    # identical SHARDS on both ranks (the VIP term couples a batch's samples, so shard-mean != full-batch; with equal shards the
    # data-parallel update must equal a single rank's update on that shard)
    dup = tuple(np.concatenate([x, x], axis=1) if x.ndim == 3 else np.concatenate([x, x], axis=0) for x in b)
    auxs = [dp.train_step(dup, 1e-3) for _ in range(3)]
    q.put((rank, tr.get_params(), auxs) + ((tr.calls,) if dry else ()))
    dist.barrier()
    tr.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_finetune_two_ranks_on_identical_shards_equal_single_rank(gpu_lib, mode):
    _need_two_gpus()
    from arp_amd import finetune as FT
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ft_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, p0, a0), (_, p1, a1) = res
    assert all(np.array_equal(p0[k], p1[k]) for k in p0)
    cfg = FT.FinetuneConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64, n_actions=5)
    tr = FT.FinetuneTrainer(cfg, mode=mode, device=0)
    tr.set_params(FT.synth_params(cfg, seed=1))
    tr.set_batch(*FT.synth_batch(cfg, 6, seed=2))
    for _ in range(3):
        tr.train_step(1e-3)
    want = tr.get_params()
    tr.close()
    tol = 2e-5 if mode == "f32" else 2e-3
    assert max(float(np.abs(want[k] - p0[k]).max()) for k in want) < tol
