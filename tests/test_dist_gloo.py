"""world_size-2 gloo tests on CPU for the N > 1 host logic: labelling shards the trajectory stream with no
data-path collective (rank 0 gathers + writes); the train step's data-parallel plumbing (arp_amd.train.DataParallel /
shard_batch: id exchange, state sync, device shards) on two ranks; the oracle's pmean algebra."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeClip:
    def set_text(self, t):
        return self

    def label(self, frames, use_crop=False):
        return frames.reshape(len(frames), -1)[:, :7].astype(np.float32).sum(1) * 0.01

    def close(self):
        pass


def _store(lens, nf=8, seed=0):
    rng = np.random.default_rng(seed)
    L = sum(lens)
    ob = rng.integers(0, 256, (L, nf, 4, 4, 3), dtype=np.uint8)
    done = np.zeros((L, nf), np.float32)
    done[np.cumsum(lens) - 1, -1] = 1
    return {"ob": ob, "done": done}


def _label_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arp_amd import label_reward as L
    st = _store([5, 9, 3, 14, 2, 8], seed=4)

    def gather(res):
        out = [None] * world
        dist.all_gather_object(out, res)
        return out

    L.label_reward("coinrun", "hard", 500, 0, "x", ".", store=st, clip_model=_FakeClip(), tokens=np.zeros((1, 77), np.int32),
                   rank=rank, world=world, gather=gather)
    if rank == 0:
        q.put({k: np.asarray(v) for k, v in st.items() if k.startswith("ob_")})
    dist.barrier()
    dist.destroy_process_group()


def _label_h5_worker(rank, world, port, path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arp_amd import label_reward as L

    def gather(res):
        out = [None] * world
        dist.all_gather_object(out, res)
        return out

    L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=path, clip_model=_FakeClip(), tokens=np.zeros((1, 77), np.int32),
                   rank=rank, world=world, gather=gather)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_labelling_of_an_hdf5_file(tmp_path):
    """Two ranks read one recorder-style HDF5 file through read-only handles, rank 0 alone reopens it to write (HDF5 is
    single-writer): the file ends up with the datasets a single process writes."""
    import pytest
    try:
        from arp_amd import h5store
        h5store.lib()
    except ImportError as e:
        pytest.skip(str(e))
    from arp_amd import label_reward as L
    lens, F = [5, 9, 3, 14, 2, 8], 8
    rng = np.random.default_rng(4)
    ob, done = [], []
    for n in lens:
        fr = rng.integers(0, 256, (n, 8, 8, 3), dtype=np.uint8)
        idx = np.clip(np.arange(n)[:, None] + np.arange(-F + 1, 1)[None, :], 0, None)
        d = np.zeros((n, F), np.float32); d[-1, -1] = 1
        ob.append(fr[idx]); done.append(d)
    ob, done = np.concatenate(ob), np.concatenate(done)
    ref = {"ob": ob, "done": done}
    L.label_reward("coinrun", "hard", 500, 0, "x", ".", store=ref, clip_model=_FakeClip(), tokens=np.zeros((1, 77), np.int32))
    p = str(tmp_path / "data.hdf5")
    with h5store.H5Store(p, "w") as f:
        f.create_dataset("ob", data=ob, compression="gzip", chunks=(1, F, 8, 8, 3), maxshape=(None, F, 8, 8, 3))
        f.create_dataset("done", data=done, compression="gzip", chunks=(1, F), maxshape=(None, F))
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_label_h5_worker, args=(r, 2, port, p)) for r in range(2)]
    [q.start() for q in procs]
    [q.join(120) for q in procs]
    assert all(q.exitcode == 0 for q in procs)
    with h5store.H5Store(p, "r") as f:
        for k in ("ob_clip_reward", "ob_clip_pos_rtg"):
            assert f[k].chunks == (1, F) and np.array_equal(f[k][...], np.asarray(ref[k])), k


def test_sharded_labelling_equals_single_process():
    from arp_amd import label_reward as L
    ref = _store([5, 9, 3, 14, 2, 8], seed=4)
    L.label_reward("coinrun", "hard", 500, 0, "x", ".", store=ref, clip_model=_FakeClip(), tokens=np.zeros((1, 77), np.int32))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_label_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert set(got) == {"ob_clip_reward", "ob_clip_pos_rtg"}
    for k in got:
        assert got[k].shape == np.asarray(ref[k]).shape and (got[k] == np.asarray(ref[k])).all(), k


class _StubTrainer:
    """Stands where PolicyTrainer stands (no GPU here): records what train.DataParallel / create_train_step ask of it, and
    implements the library's collective contract on gloo -- all-reduce(sum) of a 'gradient' computed from the shard it was
    given, 1/world folded into the update -- so the PRODUCT's sharding, id exchange and state-sync logic is what runs."""

    def __init__(self, rank, seed):
        self.rank, self.calls, self.world = rank, [], 1
        self.state = np.random.default_rng(seed).standard_normal(5)  # differs per rank until broadcast_state
        self.step = 3 * rank                                          # so does the step counter
        from arp_amd.train import PolicyConfig
        self.cfg = PolicyConfig()

    def new_unique_id(self):
        self.calls.append("new_unique_id")
        return bytes([7 + self.rank]) * 128

    def comm_init(self, uid, world, rank):
        self.calls.append(("comm_init", uid, world, rank))
        self.world = world

    def broadcast_state(self):
        self.calls.append("broadcast_state")
        t = torch.from_numpy(np.concatenate([self.state, [float(self.step)]]))
        dist.broadcast(t, src=0)
        self.state, self.step = t.numpy()[:5].copy(), int(t[5])

    def set_batch(self, enc, action, rtg):
        self.calls.append(("set_batch", enc.shape, action.shape, rtg.shape))
        self.batch = (enc, action, rtg)

    def train_step(self, lr):
        enc, action, rtg = self.batch
        g = torch.tensor([float(enc.sum()), float(action.sum()), float(rtg.sum()), float(len(action)), 1.0], dtype=torch.float64)
        dist.all_reduce(g)  # the one data-path collective
        self.state = self.state - lr * (g.numpy() / self.world)
        self.step += 1
        return {"loss": float(g[0]) / self.world, "train_state_step": self.step - 1}


def _dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arp_amd import train
    rng = np.random.default_rng(0)  # the same GLOBAL batch on every rank
    B, T = 6, 4
    batch = {"image": {"ob": rng.standard_normal((B, T, 3, 8)).astype(np.float32)}, "action": rng.integers(0, 15, (B, T)),
             "rtg": {"ob": rng.standard_normal((B, T, 1)).astype(np.float32), "ob2": rng.standard_normal((B, T, 1)).astype(np.float32)},
             "instruct": None, "text_padding_mask": None}
    tr = _StubTrainer(rank, seed=10 + rank)
    dp = train.DataParallel(tr, rank, world, train.torch_object_broadcast(dist))
    aux = dp.train_step(batch, 0.5)
    # the reference hands its pmapped fn a batch with a leading [n_devices] axis: same shards through device_axis=True
    pm = train.shard_batch({k: (None if v is None else ({kk: vv.reshape(world, -1, *vv.shape[1:]) for kk, vv in v.items()} if isinstance(v, dict)
                                                       else v.reshape(world, -1, *v.shape[1:]))) for k, v in batch.items()}, rank, world, device_axis=True)
    mine = train.shard_batch(batch, rank, world)
    same = all(np.array_equal(pm[k][kk], mine[k][kk]) for k in ("image", "rtg") for kk in mine[k]) and np.array_equal(pm["action"], mine["action"])
    q.put((rank, tr.calls, tr.state.tolist(), tr.step, aux, same, mine["action"].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_plumbing_two_ranks():
    """P10 / P12 host logic of arp_amd/train.py on two gloo ranks (the reference: pmean main_procgen.py:128-139, sync_state_fn
    :94-101, generate_batch's device reshape :645-683): rank 0's RCCL id reaches both ranks, comm_init comes before the state
    broadcast, every rank ends up with rank 0's state AND step, the shards are the reference's contiguous device slices
    (disjoint, covering the global batch, multi-view rtg averaged as ARPDT.py:285-290), and after one step both ranks hold the
    same state = the full-batch update."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in range(2)])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, c0, s0, st0, a0, same0, act0), (r1, c1, s1, st1, a1, same1, act1) = res
    assert c0[0] == "new_unique_id" and "new_unique_id" not in c1          # only rank 0 makes the id
    ci0, ci1 = c0[1], c1[0]
    assert ci0[0] == ci1[0] == "comm_init" and ci0[1] == ci1[1] == bytes([7]) * 128 and (ci0[2], ci0[3]) == (2, 0) and (ci1[2], ci1[3]) == (2, 1)
    assert c0[2] == "broadcast_state" and c1[1] == "broadcast_state"         # state sync right after the communicator exists
    assert c0[3][0] == "set_batch" and c0[3][1] == (3, 4, 3, 8) and c0[3][3] == (3, 4, 1)  # 6 samples -> 3 per rank; rtg views averaged
    assert same0 and same1
    rng = np.random.default_rng(0)
    rng.standard_normal((6, 4, 3, 8)); act = rng.integers(0, 15, (6, 4))
    assert act0 == act[:3].tolist() and act1 == act[3:].tolist()            # contiguous device slices, in order
    assert s0 == s1 and st0 == st1 == 1 and a0 == a1                         # rank 0's step (0) won the sync, then +1; same aux
    # full-batch reference of the stub's update
    rng = np.random.default_rng(0)
    enc = rng.standard_normal((6, 4, 3, 8)).astype(np.float32); act = rng.integers(0, 15, (6, 4))
    r1_ = rng.standard_normal((6, 4, 1)).astype(np.float32); r2_ = rng.standard_normal((6, 4, 1)).astype(np.float32)
    rtg = np.mean(np.stack([r1_, r2_]), axis=0)
    g = np.array([sum(float(enc[i * 3:(i + 1) * 3].sum()) for i in range(2)), float(act.sum()), sum(float(rtg[i * 3:(i + 1) * 3].sum()) for i in range(2)), 6.0, 2.0]) / 2
    want = np.random.default_rng(10).standard_normal(5) - 0.5 * g
    assert np.allclose(s0, want, rtol=0, atol=1e-9)


def _ft_dp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arp_amd import finetune as FT, train

    class Stub:
        def __init__(self):
            self.calls = []

        def new_unique_id(self):
            self.calls.append("id")
            return bytes([3]) * 128

        def comm_init(self, uid, world, rank):
            self.calls.append(("comm_init", uid, world, rank))

        def broadcast_state(self):
            self.calls.append("broadcast_state")

        def set_batch(self, *b):
            self.calls.append(("set_batch",) + tuple(np.asarray(x).shape for x in b))
            self.b = b

        def train_step(self, lr):
            t = torch.tensor([float(np.asarray(self.b[5]).sum()), float(len(self.b[5]))], dtype=torch.float64)
            dist.all_reduce(t)
            return {"loss": float(t[0] / t[1])}

    cfg = FT.FinetuneConfig(layers=2, width_v=8, width_t=8, embed=8, hidden=8)
    batch = FT.synth_batch(cfg, 6, seed=1)  # the same global batch on both ranks
    st = Stub()
    dp = FT.DataParallel(st, rank, world, train.torch_object_broadcast(dist))
    aux = dp.train_step(batch, 1e-3)
    q.put((rank, st.calls, aux))
    dist.barrier()
    dist.destroy_process_group()


def test_finetune_data_parallel_plumbing_two_ranks():
    """configs[4]'s DP wrapper (arp_amd.finetune.DataParallel / shard_batch) on two gloo ranks: id from rank 0, communicator,
    state broadcast, contiguous sample shards of all six batch arrays, identical aux on both ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ft_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in range(2)])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, c0, a0), (_, c1, a1) = res
    assert c0[0] == "id" and "id" not in c1 and c0[1][0] == "comm_init" and c0[1][1] == c1[0][1] == bytes([3]) * 128
    assert (c0[1][2], c0[1][3]) == (2, 0) and (c1[0][2], c1[0][3]) == (2, 1) and c0[2] == c1[1] == "broadcast_state"
    sb = c0[3]
    assert sb[0] == "set_batch" and sb[1] == (3, 3, 16) and sb[2] == (3, 3, 8) and sb[3] == (3, 16) and sb[6] == (3,)
    assert a0 == a1


def test_shard_batch_errors_and_identity():
    from arp_amd import train
    b = {"action": np.zeros((5, 4), np.int32), "image": {"ob": np.zeros((5, 4, 2, 2), np.float32)}, "instruct": None}
    assert train.shard_batch(b, 0, 1) is b
    import pytest
    with pytest.raises(ValueError, match="does not divide"):
        train.shard_batch(b, 0, 2)
    with pytest.raises(ValueError, match="leading device axis"):
        train.shard_batch(b, 0, 2, device_axis=True)


def _pmean_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arp_amd import synth_policy as S
    from oracle import arpdt_torch as O
    cfg = O.PolicyConfig(emb=32, depth=1, heads=2, window=2, enc_tokens=2, enc_dim=32)
    P = {k: torch.from_numpy(v) for k, v in S.policy_params(cfg, seed=1, dtype=np.float64).items()}
    enc, act, rtg = S.policy_batch(cfg, 4, seed=2, dtype=np.float64)
    sl = slice(rank * 2, rank * 2 + 2)
    g, aux, _ = O.grads(P, cfg, torch.from_numpy(enc[sl]), torch.from_numpy(act[sl]).long(), torch.from_numpy(rtg[sl]))
    flat = torch.cat([g[k].reshape(-1) for k in sorted(g)])
    dist.all_reduce(flat)  # the one collective of the train step: sum, then 1/world (folded into the update)
    flat /= world
    if rank == 0:
        gf, _, _ = O.grads(P, cfg, torch.from_numpy(enc), torch.from_numpy(act).long(), torch.from_numpy(rtg))
        q.put(float((flat - torch.cat([gf[k].reshape(-1) for k in sorted(gf)])).abs().max()))
    dist.barrier()
    dist.destroy_process_group()


def test_oracle_shard_mean_equals_full_batch_gradient_two_ranks():
    """Pins the ORACLE's pmean algebra (mean of per-shard gradients = full-batch gradient); the product's own world-2 logic
    is test_data_parallel_plumbing_two_ranks above."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pmean_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    err = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs) and err < 1e-12


def test_the_two_rank_rccl_test_workers_run_dry_on_gloo():
    """tests/test_multi_gpu.py has never executed (no box with two GPUs).  Its worker functions are driven here as they are --
    same spawn, same rendezvous, same arp_amd imports, same DataParallel wrappers, same queue protocol -- with only the trainer class replaced
    by a gloo-backed stand-in, so everything except the RCCL calls themselves is known to work: rank 0's id reaches both ranks, comm_init comes
    before the state broadcast, rank 1's differently-seeded parameters are overwritten, the shards differ, and both ranks end equal."""
    import test_multi_gpu as M
    ctx = mp.get_context("spawn")
    for worker, args in ((M._policy_worker, (1, "f16", True)), (M._ft_worker, ("f16", True))):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=worker, args=(r, 2, port, q) + args) for r in range(2)]
        [p.start() for p in procs]
        res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
        [p.join(60) for p in procs]
        assert all(p.exitcode == 0 for p in procs), worker.__name__
        p0, p1, c0, c1 = res[0][1], res[1][1], res[0][-1], res[1][-1]
        assert p0.keys() == p1.keys() and all(np.array_equal(p0[k], p1[k]) for k in p0), "ranks diverged"
        assert [a["loss"] for a in res[0][2]] == [a["loss"] for a in res[1][2]]
        assert c0[0] == ("create", "f16", 0) and c1[0] == ("create", "f16", 1)         # one device per rank
        assert c0[1] == "new_unique_id" and "new_unique_id" not in c1
        assert c0[2][0] == c1[1][0] == "comm_init" and c0[2][1] == c1[1][1] and (c0[2][2], c0[2][3]) == (2, 0) and (c1[1][2], c1[1][3]) == (2, 1)
        assert c0[3] == c1[2] == "broadcast_state"
        sb0, sb1 = [c for c in c0 if c[0] == "set_batch"], [c for c in c1 if c[0] == "set_batch"]
        assert len(sb0) == len(sb1) == len(res[0][2]) and sb0[0][1:] == sb1[0][1:]      # equal-sized shards, one per step
