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

pytestmark = pytest.mark.gpu


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


def _policy_worker(rank, world, port, q, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ARP_DT_OVERLAP=str(overlap), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch  # noqa: F401  -- before arp_amd: one HIP runtime per process
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arp_amd import synth_policy as S, train
    from arp_amd.train import PolicyConfig, PolicyTrainer
    cfg = PolicyConfig(emb=128, depth=2, heads=8, window=4, enc_tokens=9, enc_dim=128, lambda_ret=0.01)
    tr = PolicyTrainer(cfg, mode="f32", device=rank)
    tr.set_params(S.policy_params(cfg, seed=1 + rank))  # rank 1 starts elsewhere: sync_state_fn must overwrite it
    dp = train.DataParallel(tr, rank, world, train.torch_object_broadcast(dist))
    enc, act, rtg = S.policy_batch(cfg, 8, seed=5)
    batch = {"image": {"ob": enc}, "action": act, "rtg": {"ob": rtg}}
    auxs = [dp.train_step(batch, 1e-3) for _ in range(4)]
    q.put((rank, tr.get_params(), auxs, tr.get_grads()))
    dist.barrier()
    tr.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [1, 0])
def test_policy_two_ranks_equal_single_rank(gpu_lib, overlap):
    _need_two_gpus()
    from arp_amd import synth_policy as S
    from arp_amd.train import PolicyConfig, PolicyTrainer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_policy_worker, args=(r, 2, port, q, overlap)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, p0, a0, g0), (_, p1, a1, g1) = res
    assert all(np.array_equal(p0[k], p1[k]) for k in p0), "the two ranks diverged"
    assert [a["loss"] for a in a0] == [a["loss"] for a in a1]
    assert all(np.array_equal(g0[k], g1[k]) for k in g0)  # the getter returns the rank MEAN on both
    cfg = PolicyConfig(emb=128, depth=2, heads=8, window=4, enc_tokens=9, enc_dim=128, lambda_ret=0.01)
    tr = PolicyTrainer(cfg, mode="f32", device=0)
    tr.set_params(S.policy_params(cfg, seed=1))
    enc, act, rtg = S.policy_batch(cfg, 8, seed=5)
    ref = []
    for _ in range(4):
        tr.set_batch(enc, act, rtg)
        ref.append(tr.train_step(1e-3))
    want = tr.get_params()
    tr.close()
    assert max(float(np.abs(want[k] - p0[k]).max()) for k in want) < 2e-5
    assert max(abs(a["loss"] - b["loss"]) for a, b in zip(ref, a0)) < 1e-5


def _ft_worker(rank, world, port, q, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch  # noqa: F401
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arp_amd import finetune as FT, train
    cfg = FT.FinetuneConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64, n_actions=5)
    tr = FT.FinetuneTrainer(cfg, mode=mode, device=rank)
    tr.set_params(FT.synth_params(cfg, seed=1 + rank))
    dp = FT.DataParallel(tr, rank, world, train.torch_object_broadcast(dist))
    b = FT.synth_batch(cfg, 6, seed=2)
    # identical SHARDS on both ranks (the VIP term couples a batch's samples, so shard-mean != full-batch; with equal shards the
    # data-parallel update must equal a single rank's update on that shard)
    dup = tuple(np.concatenate([x, x], axis=1) if x.ndim == 3 else np.concatenate([x, x], axis=0) for x in b)
    auxs = [dp.train_step(dup, 1e-3) for _ in range(3)]
    q.put((rank, tr.get_params(), auxs))
    dist.barrier()
    tr.close()
    dist.destroy_process_group()


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
