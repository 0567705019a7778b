"""world_size-2 gloo tests on CPU for the N > 1 host logic: labelling shards the trajectory stream with no
data-path collective (rank 0 gathers + writes); the train step's pmean algebra over two ranks."""
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


def test_gradient_allreduce_algebra_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pmean_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    err = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs) and err < 1e-12
