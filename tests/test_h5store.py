"""HDF5 demonstration files through libhdf5 (ctypes), the schema of data/PPG/trajectory_recorder.py:148-176, the
access pattern of arp_dt/label_reward.py:69-87,268,273-289.  Skipped where no libhdf5 can be found."""
import os

import numpy as np
import pytest

try:
    from arp_amd import h5store
    h5store.lib()
except ImportError as e:  # pragma: no cover
    pytest.skip(f"libhdf5 not available: {e}", allow_module_level=True)

from test_host import _FakeClip

F = 8


def _recorder_file(path, lens, hw=16, seed=0, trailing=0, bool_done=False):
    """A file as the recorder writes it: per trajectory, stack_frames (deque of the last F items, first one left-padded,
    trajectory_recorder.py:103-115), datasets grown by resize (:175), gzip chunks of one row."""
    rng = np.random.default_rng(seed)
    frames, ob, done = [], [], []
    for L in lens:
        fr = rng.integers(0, 256, (L, hw, hw, 3), dtype=np.uint8)
        idx = np.clip(np.arange(L)[:, None] + np.arange(-F + 1, 1)[None, :], 0, None)
        d = np.zeros(L, np.float32)
        d[-1] = 1
        frames.append(fr); ob.append(fr[idx]); done.append(d[idx])
    if trailing:  # rows after the last done flag (an unfinished trajectory)
        fr = rng.integers(0, 256, (trailing, hw, hw, 3), dtype=np.uint8)
        idx = np.clip(np.arange(trailing)[:, None] + np.arange(-F + 1, 1)[None, :], 0, None)
        frames.append(fr); ob.append(fr[idx]); done.append(np.zeros((trailing, F), np.float32))
    with h5store.H5Store(path, "w") as f:
        f.attrs["env_name"] = "coinrun"
        for i, (o, d) in enumerate(zip(ob, done)):
            d = d.astype(bool) if bool_done else d
            if i == 0:
                f.create_dataset("ob", data=o, compression="gzip", chunks=(1, F, hw, hw, 3), maxshape=(None, F, hw, hw, 3))
                f.create_dataset("done", data=d, compression="gzip", chunks=(1, F), maxshape=(None, F))
            else:
                for k, v in (("ob", o), ("done", d)):
                    ds = f[k]
                    n0 = ds.shape[0]
                    ds.resize(n0 + len(v), axis=0)
                    ds[n0:] = v
    return np.concatenate(frames), np.concatenate(ob), np.concatenate(done)


def test_file_is_hdf5_and_schema_roundtrips(tmp_path):
    p = str(tmp_path / "data.hdf5")
    frames, ob, done = _recorder_file(p, [5, 17, 9, 1])
    assert open(p, "rb").read(8) == b"\x89HDF\r\n\x1a\n"  # the format signature: written by libhdf5 itself
    assert h5store.lib_version() >= (1, 10, 3)
    with h5store.H5Store(p, "r") as f:
        assert sorted(f.keys()) == ["done", "ob"] and "ob" in f and "nope" not in f and f.get("nope") is None
        assert f.attrs["env_name"] == "coinrun" and f.attrs.get("missing", 3) == 3
        d = f["ob"]
        assert d.shape == ob.shape and d.dtype == np.uint8 and d.chunks == (1, F, 16, 16, 3) and d.maxshape == (None, F, 16, 16, 3)
        assert d.compression == "gzip" and d.compression_opts == 4 and d.fast_path_ok()
        assert f["done"].shape == (32, F) and f["done"].dtype == np.float32 and f["done"].chunks == (1, F)
        assert np.array_equal(d[...], ob) and np.array_equal(np.asarray(d), ob)
        assert np.array_equal(d[3:9, -1], ob[3:9, -1]) and np.array_equal(d[7], ob[7]) and np.array_equal(d[-1, 0], ob[-1, 0])
        assert np.array_equal(d[[3, 4, 5], -1], ob[[3, 4, 5], -1])      # g[img_key][traj, -1] with a list (label_reward.py:268)
        assert np.array_equal(d[[9, 2, 30], -1], ob[[9, 2, 30], -1])    # non-consecutive rows
        assert np.array_equal(f["done"][:, -1], done[:, -1])
        with pytest.raises(IndexError):
            d[40]
        with pytest.raises(KeyError):
            f["nope"]
        with pytest.raises(h5store.H5Error):
            f.create_dataset("x", data=np.zeros(3, np.float32))          # read-only file
    with pytest.raises(h5store.H5Error):
        h5store.H5Store(str(tmp_path / "absent.hdf5"), "r")


def test_read_last_frames_equals_per_row_reads(tmp_path):
    p = str(tmp_path / "data.hdf5")
    lens = [5, 17, 9, 1, 8, 16]
    frames, ob, _ = _recorder_file(p, lens, seed=1)
    with h5store.H5Store(p, "r") as f:
        d = f["ob"]
        s = 0
        for L in lens:
            for threads in (1, 4):
                got = d.read_last_frames(s, s + L, threads=threads)
                assert got.dtype == np.uint8 and np.array_equal(got, frames[s : s + L]) and np.array_equal(got, ob[s : s + L, -1])
            assert np.array_equal(d.read_last_frames(s, s + L, stacked=False), frames[s : s + L])
            for native in (False, True):  # Python pool vs the C++ threads of arp_h5_inflate_last_frames: same bytes
                assert np.array_equal(d.read_last_frames(s, s + L, native=native, native_threads=3), frames[s : s + L])
                assert np.array_equal(d.read_last_frames(s, s + L, native=native, stacked=False), frames[s : s + L])
            s += L
        assert d._stack_ok is True
        assert d.read_last_frames(4, 4).shape == (0, 16, 16, 3)
        bounds = np.concatenate([[0], np.cumsum(lens)])
        spans = [(int(bounds[i]), int(bounds[i + 1])) for i in (1, 2, 4)] + [(3, 3)]
        want = np.concatenate([frames[a:b] for a, b in spans])
        for native in (False, True):
            assert np.array_equal(d.read_last_frames_spans(spans, native=native), want)
            assert np.array_equal(d.read_last_frames_spans(spans, native=native, stacked=False), want)
        # a partial range inside a trajectory (a rank's shard never splits one, but the reader does not care)
        assert np.array_equal(d.read_last_frames(7, 19), frames[7:19])


def test_unstacked_file_falls_back_to_per_row_reads(tmp_path):
    """A file whose rows are NOT the recorder's sliding window must still give g[key][rows, -1]."""
    p = str(tmp_path / "odd.hdf5")
    rng = np.random.default_rng(2)
    ob = rng.integers(0, 256, (20, F, 8, 8, 3), dtype=np.uint8)  # independent frames in every slot
    with h5store.H5Store(p, "w") as f:
        f.create_dataset("ob", data=ob, compression="gzip", chunks=(1, F, 8, 8, 3), maxshape=(None, F, 8, 8, 3))
    with h5store.H5Store(p, "r") as f:
        d = f["ob"]
        assert np.array_equal(d.read_last_frames(0, 20), ob[:, -1]) and d._stack_ok is False
        assert np.array_equal(d.read_last_frames(3, 11), ob[3:11, -1])
    # uncompressed / differently chunked datasets take the plain path
    with h5store.H5Store(p, "a") as f:
        f.create_dataset("raw", data=ob)
        assert not f["raw"].fast_path_ok() and np.array_equal(f["raw"].read_last_frames(2, 9), ob[2:9, -1])


def test_unstacked_file_with_a_length_one_first_trajectory(tmp_path):
    """the stacked fast path was trusted after checking ONE group, and a one-row group (first trajectory of length 1)
    passes that check trivially.  Only a group of >= 2 rows may latch the verdict; a non-stacked file must still read as
    g[key][rows, -1], whatever the first trajectory's length."""
    p = str(tmp_path / "odd1.hdf5")
    rng = np.random.default_rng(4)
    ob = rng.integers(0, 256, (30, F, 8, 8, 3), dtype=np.uint8)  # independent frames in every slot: NOT a sliding window
    with h5store.H5Store(p, "w") as f:
        f.create_dataset("ob", data=ob, compression="gzip", chunks=(1, F, 8, 8, 3), maxshape=(None, F, 8, 8, 3))
    spans = [(0, 1), (1, 2), (2, 13), (13, 30)]  # trajectories of length 1, 1, 11, 17
    with h5store.H5Store(p, "r") as f:
        d = f["ob"]
        one = d.read_last_frames_spans(spans[:2])  # one-row groups only: correct, and nothing may be latched
        assert np.array_equal(one, ob[0:2, -1]) and not getattr(d, "_stack_checked", False)
        got = d.read_last_frames_spans(spans)
        assert np.array_equal(got, ob[:, -1]) and d._stack_checked and d._stack_ok is False
    with h5store.H5Store(p, "r") as f:  # fresh handle, whole file in one call (label_store's pattern)
        d = f["ob"]
        assert np.array_equal(d.read_last_frames_spans(spans), ob[:, -1]) and d._stack_ok is False


@pytest.mark.parametrize("bool_done", [False, True])
def test_label_reward_on_an_hdf5_file_matches_the_mapping_path(tmp_path, bool_done):
    """label_reward(data_path=...) end to end (fake model): same datasets as the store= path and as the oracle loop; created as
    gzip chunks (1, num_frames) float32 with an unlimited first axis (label_reward.py:277-283); a second run overwrites in place."""
    from arp_amd import label_reward as L
    from oracle import rtg
    p = str(tmp_path / "data.hdf5")
    lens = [1, 3, 9, 17]
    frames, ob, done = _recorder_file(p, lens, hw=8, seed=3, trailing=3, bool_done=bool_done)
    fake = _FakeClip()
    ref = rtg.label_file({"ob": ob, "done": done}, lambda im: fake.label(im))
    tok = np.zeros((1, 77), np.int32)
    L.label_reward("coinrun", "hard", 500, 0, "the goal is to collect the coin.", ".", data_path=p, clip_model=fake, tokens=tok)
    with h5store.H5Store(p, "r") as f:
        assert set(ref) <= set(f.keys())
        for k, v in ref.items():
            d = f[k]
            assert d.dtype == np.float32 and d.chunks == (1, F) and d.compression == "gzip" and d.maxshape == (None, F)
            assert np.array_equal(d[...], v), k
    mtime_keys = None
    with h5store.H5Store(p, "r") as f:
        mtime_keys = sorted(f.keys())

    class Shifted(_FakeClip):
        def label(self, frames, use_crop=False):
            return super().label(frames) + 1.0

    L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=p, clip_model=Shifted(), tokens=tok)
    ref2 = rtg.label_file({"ob": ob, "done": done}, lambda im: Shifted().label(im))
    with h5store.H5Store(p, "r") as f:
        assert sorted(f.keys()) == mtime_keys
        for k, v in ref2.items():
            assert np.array_equal(f[k][...], v), k


def test_streamed_and_collected_writes_give_the_same_file_content(tmp_path, monkeypatch):
    """RowSink (rows written by a writer thread while the next batch is labelled) against ARP_LABEL_STREAM_WRITE=0 (the reference's
    order: label everything, then write): same datasets, shapes, filters and values -- also when the file has rows after the last
    `done` (not labelled, not part of the datasets) and when the datasets already exist and are longer or shorter."""
    from arp_amd import label_reward as L
    tok = np.zeros((1, 77), np.int32)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ARP_LABEL_STREAM_WRITE", mode)
        p = str(tmp_path / f"data{mode}.hdf5")
        _recorder_file(p, [5, 40, 2, 130, 9], hw=8, seed=11, trailing=4)
        L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=p, clip_model=_FakeClip(), tokens=tok)
        with h5store.H5Store(p, "a") as f:  # a too-short and a too-long existing dataset, then a second run
            f["ob_clip_reward"].resize(100, axis=0)
            f["ob_clip_pos_rtg"].resize(400, axis=0)
        L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=p, clip_model=_FakeClip(), tokens=tok)
        with h5store.H5Store(p, "r") as f:
            out[mode] = {k: (f[k][...], f[k].chunks, f[k].compression, f[k].maxshape, f[k].dtype) for k in sorted(f.keys()) if k.startswith("ob_clip")}
    assert set(out["0"]) == set(out["1"]) == {"ob_clip_reward", "ob_clip_pos_rtg"}
    for k in out["0"]:
        a, b = out["0"][k], out["1"][k]
        assert a[1:] == b[1:] and a[0].shape == b[0].shape and np.array_equal(a[0], b[0]), k
    assert out["1"]["ob_clip_reward"][0].shape == (186, F) and out["1"]["ob_clip_pos_rtg"][0].shape == (400, F)


def test_a_failing_writer_thread_fails_the_call(tmp_path, monkeypatch):
    """RowSink's writer thread must not swallow an error: label_reward raises it, and the file handle is still closed (a second open works)."""
    from arp_amd import label_reward as L
    p = str(tmp_path / "data.hdf5")
    _recorder_file(p, [30, 70], hw=8, seed=5)

    def boom(self, *a, **k):
        raise h5store.H5Error("disk full")

    monkeypatch.setattr(h5store.H5Store, "create_dataset", boom)
    with pytest.raises(h5store.H5Error, match="disk full"):
        L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=p, clip_model=_FakeClip(), tokens=np.zeros((1, 77), np.int32))
    monkeypatch.undo()
    with h5store.H5Store(p, "a") as f:
        assert "ob_clip_reward" not in f.keys()


def test_a_failure_in_the_middle_of_a_pass_leaves_no_half_labelled_datasets(tmp_path, monkeypatch):
    """with the streamed writes a GPU error (here: a labeller that fails on its third batch) used to leave reward datasets created at
    full length with a prefix of real rows and fill values behind it.  The reference writes only after everything is labelled
    (label_reward.py:273-289): a failed pass deletes the datasets it created and cuts the ones it grew back to their former length."""
    from arp_amd import label_reward as L
    tok = np.zeros((1, 77), np.int32)

    class Flaky(_FakeClip):
        def __init__(self, fail_after):
            self.calls, self.fail_after = 0, fail_after

        def label(self, frames, use_crop=False):
            self.calls += 1
            if self.calls > self.fail_after:
                raise RuntimeError("GPU fell over")
            return super().label(frames, use_crop)

    monkeypatch.setattr(L, "_BATCH_FRAMES_DEFAULT", 64, raising=False)
    # (a) fresh file: nothing may remain
    p = str(tmp_path / "fresh.hdf5")
    _recorder_file(p, [700, 650, 900, 800], hw=8, seed=3)
    with pytest.raises(RuntimeError, match="GPU fell over"):
        L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=p, clip_model=Flaky(2), tokens=tok)
    with h5store.H5Store(p, "r") as f:
        assert not [k for k in f.keys() if k.startswith("ob_clip")], "a failed pass left datasets behind"
    # (b) datasets exist but are shorter (a file that was extended after an earlier labelling): back to their former length, former rows intact or relabelled
    p2 = str(tmp_path / "grown.hdf5")
    _recorder_file(p2, [700, 650], hw=8, seed=3)
    L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=p2, clip_model=_FakeClip(), tokens=tok)
    with h5store.H5Store(p2, "r") as f:
        before = {k: f[k][...] for k in f.keys() if k.startswith("ob_clip")}
    with h5store.H5Store(p2, "a") as f:  # the recorder appends two more trajectories
        rng = np.random.default_rng(9)
        for L_ in (900, 800):
            fr = rng.integers(0, 256, (L_, 8, 8, 3), dtype=np.uint8)
            idx = np.clip(np.arange(L_)[:, None] + np.arange(-F + 1, 1)[None, :], 0, None)
            d = np.zeros((L_, F), np.float32); d[-1, -1] = 1
            for k, v in (("ob", fr[idx]), ("done", d)):
                ds = f[k]; n0 = ds.shape[0]; ds.resize(n0 + L_, axis=0); ds[n0:] = v
    with pytest.raises(RuntimeError, match="GPU fell over"):
        L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=p2, clip_model=Flaky(2), tokens=tok)
    with h5store.H5Store(p2, "r") as f:
        for k, v in before.items():
            assert f[k].shape == v.shape and np.array_equal(f[k][...], v), k


def test_default_path_layout(tmp_path):
    """data_path=None builds <base>/<env>_<mode>_level<start>to<num>_num<demos>_frame<frames>[_<env_type>]/data.hdf5 (label_reward.py:62-68)."""
    from arp_amd import label_reward as L
    d = tmp_path / "coinrun_hard_level0to500_num500_frame8_et"
    os.makedirs(d)
    _recorder_file(str(d / "data.hdf5"), [4, 6], hw=8)
    L.label_reward("coinrun", "hard", 500, 0, "x", str(tmp_path), env_type="et", clip_model=_FakeClip(), tokens=np.zeros((1, 77), np.int32))
    with h5store.H5Store(str(d / "data.hdf5"), "r") as f:
        assert f["ob_clip_reward"].shape == (10, F)


def test_native_inflater_reports_corrupt_chunks(tmp_path):
    """arp_h5_inflate_last_frames (csrc/arp_io.cpp) fails loudly -- error text through arp_last_error -- on a truncated stream, a
    wrong size and bad arguments; an unwritten chunk reads as zeros."""
    import ctypes as C
    import zlib
    from arp_amd import _ffi
    raw = np.arange(8 * 48, dtype=np.uint8).reshape(8, 48)
    z = zlib.compress(raw.tobytes(), 4)
    p = tmp_path / "blob.bin"
    p.write_bytes(b"\0" * 7 + z + z[: len(z) // 2])
    fd = os.open(str(p), os.O_RDONLY)
    u64, u8 = C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)

    def run(addr, size, cnt, chunk_bytes=8 * 48, rawf=None):
        addr, size = np.asarray(addr, np.uint64), np.asarray(size, np.uint64)
        cnt = np.asarray(cnt, np.uint32)
        offs = (np.concatenate([[0], np.cumsum(cnt)[:-1]]) * 48).astype(np.uint64)
        out = np.full((int(cnt.sum()), 48), 255, np.uint8)
        rf = None if rawf is None else np.asarray(rawf, np.uint8).ctypes.data_as(u8)
        rc = _ffi.lib.arp_h5_inflate_last_frames(fd, len(addr), addr.ctypes.data_as(u64), size.ctypes.data_as(u64), rf, chunk_bytes, 48,
                                                 offs.ctypes.data_as(u64), cnt.ctypes.data_as(C.POINTER(C.c_uint32)), out.ctypes.data_as(u8), 2)
        return rc, out

    rc, out = run([7, 7, 0], [len(z), len(z), 0], [8, 3, 2])
    assert rc == 0 and np.array_equal(out[:8], raw) and np.array_equal(out[8:11], raw[5:]) and (out[11:] == 0).all()
    rc, _ = run([7 + len(z)], [len(z) // 2], [1])                       # truncated deflate stream
    assert rc < 0 and "inflate" in _ffi.last_error()
    rc, _ = run([7], [len(z)], [1], chunk_bytes=16 * 48)                # inflates to fewer bytes than the chunk should hold
    assert rc < 0
    rc, _ = run([7], [len(z)], [9])                                     # more frames than a chunk has
    assert rc < 0 and "cnt" in _ffi.last_error()
    rc, _ = run([0], [10_000], [1])                                     # reads past the end of the file
    assert rc < 0 and "pread" in _ffi.last_error()
    os.close(fd)
