"""CPU tests of the host side: the C ABI loads and exports every declared symbol, host-only entry
points, the label_reward mirror's plumbing against the oracle and the reference-generated goldens,
sharding.  No compute call needs a GPU here."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def _declared():
    names = []
    for hdr in sorted(os.listdir(os.path.join(ROOT, "include"))):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names += re.findall(r"\b(arp_[a-z0-9_]+)\s*\(", txt)
    return sorted(set(names))


def test_abi_exports_every_declared_symbol():
    from arp_amd import _ffi
    decl = _declared()
    assert len(decl) >= 25
    for n in decl:
        assert hasattr(_ffi.lib, n), f"libarp_hip.so does not export {n}"
    assert sorted(_ffi.SIGNATURES) == [n for n in decl if n in _ffi.SIGNATURES]
    missing = [n for n in decl if n not in _ffi.SIGNATURES]
    assert not missing, f"ctypes binding lacks {missing}"
    assert _ffi.lib.arp_version() >= 100


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, never compute on the CPU."""
    from arp_amd import _ffi, clip
    if _ffi.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_ffi.ArpError):
        clip.preprocess(np.zeros((1, 256, 256, 3), np.uint8))
    with pytest.raises(_ffi.ArpError):
        clip.ClipLabeller(clip.VIT_B32, {}, mode="bf16")
    from arp_amd import finetune, train
    with pytest.raises(_ffi.ArpError):
        finetune.FinetuneTrainer(finetune.FinetuneConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64), mode="f32")
    with pytest.raises(_ffi.ArpError):
        train.PolicyTrainer(train.PolicyConfig(), mode="f32")
    out = np.zeros((4, 4), np.float32)
    import ctypes as C
    p = out.ctypes.data_as(C.POINTER(C.c_float))
    assert _ffi.lib.arp_op_gemm_nt(0, 0, p, p, None, None, p, 4, 4, 32) < 0
    assert _ffi.last_error()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "arp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f"{f} imports the oracle"


@pytest.mark.parametrize("io", [(256, 224), (128, 224), (64, 224), (300, 224), (224, 224), (512, 224), (48, 224)])
def test_bicubic_tables_match_oracle(io):
    from arp_amd import clip
    from oracle import preprocess as P
    xm, ct, W = clip.bicubic_coeffs(*io)
    oxm, oct_, oW = P.bicubic_coeffs(*io)
    k = min(oW.shape[1], W.shape[1])
    assert (xm == oxm).all() and (ct == oct_).all() and (W[:, :k] == oW[:, :k]).all()
    assert (W.sum(1) >= (1 << 22) - 8).all() and (W.sum(1) <= (1 << 22) + 8).all()


def test_discount_cumsum_and_stack_match_oracle():
    from arp_amd import label_reward as L
    from oracle import rtg
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 9, 17, 300):
        x = (rng.standard_normal(n) * 5).astype(np.float32)
        assert (L.discount_cumsum(x) == rtg.discount_cumsum(x)).all()  # bit-identical f32 summation order
        assert np.allclose(L.discount_cumsum(x, 0.9), rtg.discount_cumsum(x, 0.9))
        for nf in (1, 4, 8):
            assert (L.stack_outputs(x, nf) == rtg.stack_outputs(x, nf)).all()
    assert L.discount_cumsum(np.float32(3.0)).shape == (1,)
    assert L.stack_outputs(np.float32(3.0), 8).shape == (1, 8)


class _FakeClip:
    """Stands in for ClipLabeller in host-logic tests: reward = first byte of each frame."""

    def set_text(self, tokens):
        self.tokens = np.asarray(tokens)
        return self

    def label(self, frames, use_crop=False):
        return frames.reshape(len(frames), -1)[:, 0].astype(np.float32) * 0.25 - 3

    def encode_image(self, frames, use_crop=False, normalize=False):
        return frames.reshape(len(frames), -1)[:, :4].astype(np.float32)

    def close(self):
        pass


def _store(lens, nf=8, seed=0, trailing=0):
    rng = np.random.default_rng(seed)
    L = sum(lens) + trailing
    ob = rng.integers(0, 256, (L, nf, 4, 4, 3), dtype=np.uint8)
    done = np.zeros((L, nf), np.float32)
    done[np.cumsum(lens) - 1, -1] = 1
    return {"ob": ob, "done": done}


def test_label_reward_mirror_matches_oracle_loop():
    from arp_amd import label_reward as L
    from oracle import rtg
    for lens, trailing in (([1, 3, 9, 17], 0), ([8, 8], 0), ([5], 0), ([4, 6], 3)):
        st = _store(lens, trailing=trailing)
        fake = _FakeClip()
        ref = rtg.label_file(st, lambda im: fake.label(im))
        L.label_reward("coinrun", "hard", 500, 0, "the goal is to collect the coin.", ".", store=st, clip_model=fake,
                       tokens=np.zeros((1, 77), np.int32), num_frames=3)  # num_frames is overwritten from the file (quirk Q2)
        for k, v in ref.items():
            assert k in st and (np.asarray(st[k]) == v).all(), k
        assert set(ref) == {"ob_clip_reward", "ob_clip_pos_rtg"}  # writer key names (quirk Q3)


def test_label_reward_mirror_matches_reference_goldens():
    from arp_amd import label_reward as L
    g = np.load(os.path.join(G, "rtg.npz"))

    class Fake(_FakeClip):
        def label(self, frames, use_crop=False):
            return frames[:, 0, 0, 0].astype(np.float32)

    for case in "abct":
        rewards, done = g[f"{case}_rewards"], g[f"{case}_done"]
        Ln, nf = done.shape
        ob = np.zeros((Ln, nf, 1, 1, 3), np.float32)
        ob[:, -1, 0, 0, 0] = rewards
        st = {"ob": ob, "done": done}
        if case == "t":  # boundaries from `time` when the done-key path raises (label_reward.py:84-87)
            st = {"ob": ob, "done": done[:, -1].copy(), "time": g["t_time"]}
        L.label_reward("coinrun", "hard", 500, 0, "x", ".", store=st, clip_model=Fake(), tokens=np.zeros((1, 77), np.int32))
        for k in g[f"{case}_keys"]:
            assert (np.asarray(st[str(k)]) == g[f"{case}__{k}"]).all(), (case, k)


def test_label_reward_variants_and_errors():
    from arp_amd import label_reward as L
    st = _store([3, 4])
    L.label_reward("coinrun", "hard", 500, 0, "x", ".", store=st, clip_model=_FakeClip(), model_type="clip_goal_conditioned",
                   inst_type="random1")
    assert "ob_clip_goal_conditioned_reward_random1" in st and "ob_clip_goal_conditioned_pos_rtg_random1" in st
    assert st["ob_clip_goal_conditioned_reward_random1"][2, -1] == 0  # last frame of a trajectory is its own goal
    with pytest.raises(ValueError):
        L.label_reward("coinrun", "hard", 500, 0, "x", ".", store={"ob": st["ob"]}, clip_model=_FakeClip(),
                       tokens=np.zeros((1, 77), np.int32))  # no done / rewards / is_terminal key (label_reward.py:71-78)
    with pytest.raises(NotImplementedError):
        L.label_reward("coinrun", "hard", 500, 0, "x", ".", store=st, clip_model=_FakeClip(), model_type="r3m")
    # the fine-tuned-model branch (label_reward.py:165-230) writes its own dataset names
    L.label_reward("coinrun", "hard", 500, 0, "x", ".", store=st, clip_model=_FakeClip(), model_type="clip_ft", tokens=np.zeros((1, 77), np.int32))
    assert "ob_clip_ft_reward" in st and "ob_clip_ft_pos_rtg" in st
    with pytest.raises(ValueError):
        L.label_reward("coinrun", "hard", 500, 0, "x", ".", store=st, clip_model=_FakeClip())  # neither tokens nor tokenizer
    from arp_amd import data
    assert data.get_clip_instruct("coinrun") == "the goal is to collect the coin."
    # data_procgen.py:296-317: every branch, and the ValueError when none returns
    assert data.get_clip_special_instruct("coinrun_aisc", "misinfo") == "The agent must go to the far right of the level."
    assert data.get_clip_special_instruct("maze_aisc", "misinfo") == "navigate a maze to reacth to the top right corner."
    assert data.get_clip_special_instruct("maze_yellowline", "misinfo") == "navigate a maze to collect yellow gem."
    assert data.get_clip_special_instruct("coinrun", "misinfo2") == "The goal is to collect the red strawberry."
    assert data.get_clip_special_instruct("coinrun", "misinfo3") == "The goal is to reach the saw."
    assert data.get_clip_special_instruct("coinrun", "misinfo4") == "The goal is to jump as high as you can."
    for env, inst in (("maze", "misinfo"), ("maze", "misinfo2"), ("coinrun", "none"), ("coinrun", "bogus")):
        with pytest.raises(ValueError, match="You must pass any condition"):
            data.get_clip_special_instruct(env, inst)
    # sharded call without a gather: refuse instead of writing a short file
    with pytest.raises(ValueError, match="gather"):
        L.label_reward("coinrun", "hard", 500, 0, "x", ".", store=_store([3, 4]), clip_model=_FakeClip(), tokens=np.zeros((1, 77), np.int32),
                       rank=0, world=2)
    # goal-conditioned rewards keep the float64 the reference's compute_reward returns (label_reward.py:163)
    assert st["ob_clip_goal_conditioned_reward_random1"].dtype == np.float64


def test_shard_trajectories_balanced_and_complete():
    from arp_amd import label_reward as L
    rng = np.random.default_rng(1)
    for world in (1, 2, 3, 8):
        for ntraj in (1, 2, 7, 40):
            lens = rng.integers(1, 50, ntraj)
            bounds = [0] + list(np.cumsum(lens))
            sh = L.shard_trajectories(bounds, world)
            assert len(sh) == world and sh[0][0] == 0 and sh[-1][1] == ntraj
            assert all(sh[i][1] == sh[i + 1][0] for i in range(world - 1))
            if ntraj >= 4 * world:
                per = [bounds[b] - bounds[a] for a, b in sh]
                assert max(per) <= sum(lens) / world + 50


def test_trajectory_batching_changes_nothing():
    """label_store labels several short trajectories per compute_reward call; values, stacking and rtg are per trajectory."""
    from arp_amd import label_reward as L
    st = _store([3, 5, 1, 4, 2])
    calls = []

    class Counting(_FakeClip):
        def label(self, frames, use_crop=False):
            calls.append(len(frames))
            return super().label(frames, use_crop)

    cr = L.make_compute_reward("clip")
    one = L.label_store(st, Counting(), cr, batch_frames=0)
    n_one = len(calls)
    calls.clear()
    many = L.label_store(st, Counting(), cr, batch_frames=8)
    assert n_one == 5 and calls == [8, 7]  # [3+5], [1+4+2]
    assert one.keys() == many.keys()
    for k in one:
        assert one[k][0] == many[k][0] and np.array_equal(one[k][1], many[k][1])
    calls.clear()
    L.label_store(st, Counting(), L.make_compute_reward("clip_goal_conditioned"), model_type="clip_goal_conditioned", batch_frames=8)
    assert calls == []  # goal-conditioned goes through encode_image, one trajectory at a time


def test_policy_batch_rtg_views_mean_and_symlog():
    """batch["rtg"] holds one array per image view; the policy consumes their mean, each view through symlog first when
    config.use_symlog (arp_dt/ARPDT.py:251-258,281-293; symlog = arp_dt/utils.py:445-446)."""
    from arp_amd.train import _batch_arrays, symlog
    rng = np.random.default_rng(5)
    enc = rng.standard_normal((2, 4, 3, 8)).astype(np.float32)
    act = rng.integers(0, 15, (2, 4))
    views = {"ob": rng.standard_normal((2, 4, 1)).astype(np.float32) * 30, "ob2": rng.standard_normal((2, 4, 1)).astype(np.float32) * 30}
    batch = {"image": {"ob": enc}, "action": act, "rtg": views}
    e, a, r = _batch_arrays(batch)
    assert e is not None and np.array_equal(a, act)
    np.testing.assert_allclose(r, (views["ob"] + views["ob2"]) / 2, rtol=1e-6)
    _, _, rs = _batch_arrays(batch, use_symlog=True)
    want = np.mean([np.sign(v) * np.log(1 + np.abs(v)) for v in views.values()], axis=0)
    np.testing.assert_allclose(rs, want, rtol=1e-6)
    assert not np.allclose(rs, r)
    np.testing.assert_allclose(symlog(np.array([-np.e + 1, 0.0, np.e - 1])), [-1.0, 0.0, 1.0], atol=1e-6)
    _, _, r1 = _batch_arrays({"image": enc, "action": act, "rtg": views["ob"]}, use_symlog=True)  # a bare array = one view
    np.testing.assert_allclose(r1, symlog(views["ob"]), rtol=1e-6)


def test_allreduce_bucket_plan_tiles_the_flat_gradient_exactly_once():
    """the data-parallel step all-reduces the flat gradient in two buckets of two ranges; every float must be in
    exactly one range, bucket 1 must hold image_text_input's kernel (the bulk, produced before the adapter's backward)."""
    from arp_amd.train import PolicyConfig, bucket_plan
    for cfg in (PolicyConfig(), PolicyConfig(use_adapter=False), PolicyConfig(emb=64, depth=3, heads=4, window=3, enc_tokens=5, enc_dim=64)):
        ranges, total = bucket_plan(cfg)
        assert len(ranges) == 4 and all(0 <= lo <= hi <= total for lo, hi in ranges)
        cover = np.zeros(total, np.int32)
        for lo, hi in ranges:
            cover[lo:hi] += 1
        assert (cover == 1).all(), "the bucket ranges must tile [0, P) exactly once"
        wi = cfg.enc_tokens * cfg.enc_dim * cfg.emb
        assert ranges[0][1] - ranges[0][0] >= wi                       # bucket 1, range a: image_text_input/kernel + the transformer's matrices
        b2 = sum(hi - lo for lo, hi in ranges[2:])
        if cfg.use_adapter:
            assert b2 == 2 * cfg.enc_dim * cfg.enc_dim + 2 * cfg.enc_dim + 4  # two Dense kernels, two biases, residual_weight (padded to 4)
        else:
            assert b2 == 0
    ranges, total = bucket_plan(PolicyConfig())
    assert sum(hi - lo for lo, hi in ranges[:2]) / total > 0.94            # what overlaps the adapter's backward GEMMs


def test_split_rng_is_jax_random_split():
    """train_step_fn / val_step_fn return jax.random.split(rng)[0] (main_procgen.py:130,163).  Known answer: the documented
    jax.random.split(jax.random.PRNGKey(0)) = [[4146024105, 967050713], [2718843009, 1272950319]] (threefry2x32, original split)."""
    from arp_amd.train import split_rng
    nxt, sub = split_rng(np.array([0, 0], np.uint32))
    assert nxt.tolist() == [4146024105, 967050713] and sub.tolist() == [2718843009, 1272950319]
    # the pmapped "sharded_rng": one key per device, split independently
    keys = np.array([[0, 0], [0, 1]], np.uint32)
    n2, s2 = split_rng(keys)
    assert n2.shape == (2, 2) and n2[0].tolist() == [4146024105, 967050713] and not np.array_equal(n2[0], n2[1])
    # anything that is not a raw threefry key is carried through untouched
    assert split_rng(None) == (None, None) and split_rng(7) == (7, 7)


def test_label_reward_rejects_a_misconfigured_shard_before_doing_any_work():
    """rank / world / gather are validated BEFORE the store or the model is touched."""
    from arp_amd.label_reward import label_reward, reward_dtype

    class Boom(dict):
        def __getitem__(self, k):
            raise AssertionError("the store was touched before validation")

    kw = dict(env_name="coinrun", distribution_mode="hard", num_levels=500, start_level=0, text="x", base_path="/nonexistent")
    with pytest.raises(ValueError, match="gather"):
        label_reward(**kw, store=Boom(), clip_model=object(), world=2, rank=0)
    with pytest.raises(ValueError, match="rank"):
        label_reward(**kw, store=Boom(), clip_model=object(), world=2, rank=2, gather=lambda r: [r])
    assert reward_dtype("clip") == np.float32 and reward_dtype("clip_ft") == np.float32 and reward_dtype("clip_goal_conditioned") == np.float64


def test_empty_shard_keeps_the_goal_conditioned_dtype():
    """a rank without trajectories must hand over float64 (0, F) rows for the goal-conditioned model, so that whichever rank's
    result reaches create_dataset first fixes the reference's dtype."""
    from arp_amd.label_reward import label_store
    store = _store([5])  # one trajectory: rank 1 of 2 gets nothing
    dist = lambda m, im, text=None, use_crop=False: np.arange(len(im), dtype=np.float64)
    for rank in (0, 1):
        res = label_store(store, None, dist, model_type="clip_goal_conditioned", rank=rank, world=2)
        assert res, "both ranks report both datasets"
        for k, (first, rows) in res.items():
            assert rows.dtype == np.float64, (rank, k, rows.dtype)
        assert sum(rows.shape[0] for _, rows in res.values()) == (0 if rank == 1 else 10) or rank == 0


def test_finetune_bucket_plan_tiles_the_flat_gradient_exactly_once():
    from arp_amd.finetune import FinetuneConfig, bucket_plan
    for cfg in (FinetuneConfig(), FinetuneConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64, n_actions=5)):
        buckets, total = bucket_plan(cfg)
        cover = np.zeros(total, np.int8)
        for b in buckets:
            for lo, hi in b:
                assert 0 <= lo <= hi <= total
                cover[lo:hi] += 1
        assert (cover == 1).all() and len(buckets) == 7
    # the buckets that overlap the backward carry all but the last intermediate-linear weight and the scalars
    buckets, total = bucket_plan(FinetuneConfig())
    last = sum(hi - lo for lo, hi in buckets[-1])
    assert last / total < 0.1


def test_two_prefetchers_on_one_trainer_never_share_device_slots():
    """the reference opens train_iter AND val_iter with prefetch_to_device(..., 2) on the same state (main_procgen.py:703-708).
    The trainer's two device slots have ONE owner at a time: the second live prefetcher yields host batches (staged synchronously by the step
    functions), so no upload can land in a slot that holds an uploaded-but-unconsumed batch of the other iterator.  Stub trainer, no GPU."""
    from arp_amd.train import DeviceBatch, PolicyConfig, prefetch_to_device
    import threading

    class Stub:
        cfg = PolicyConfig(emb=8, depth=1, heads=2, window=2, enc_tokens=2, enc_dim=4)

        def __init__(self):
            self.holds = {}  # slot -> tag of the uploaded-but-unconsumed batch in it
            self.lock = threading.Lock()
            self.clobbered = []

        def upload_async(self, slot, enc, act, rtg, images=False):
            with self.lock:
                if slot in self.holds:
                    self.clobbered.append((slot, self.holds[slot], float(enc.flat[0])))
                self.holds[slot] = float(enc.flat[0])

    def batches(tag, n):
        for i in range(n):
            yield {"image": {"ob": np.full((2, 2, 2, 4), tag + i, np.float32)}, "action": np.zeros((2, 2), np.int32), "rtg": {"ob": np.zeros((2, 2, 1), np.float32)}}

    tr = Stub()
    train_it = prefetch_to_device(batches(100, 6), 2, tr)
    with pytest.warns(RuntimeWarning, match="another prefetcher owns"):  # ownership is decided at CREATION, and the demotion is not silent
        val_it = prefetch_to_device(batches(900, 3), 2, tr)
    assert next(val_it)["image"]["ob"].flat[0] == 900.0   # pulling the second one FIRST does not make it the owner
    seen_train, seen_val = [], [900.0]
    for step in range(6):
        b = next(train_it)
        assert isinstance(b, DeviceBatch)
        with tr.lock:
            seen_train.append(tr.holds.pop(b.slot))  # the step consumes what the slot holds
        b.done()
        if step in (1, 3):
            v = next(val_it)
            assert isinstance(v, dict), "the second live prefetcher must not use the trainer's device slots"
            seen_val.append(float(v["image"]["ob"].flat[0]))
    assert seen_train == [100.0 + i for i in range(6)] and seen_val == [900.0, 901.0, 902.0]
    assert tr.clobbered == []
    train_it.close()
    val_it.close()
    # once the owner is gone the slots can be claimed again
    again = prefetch_to_device(batches(500, 1), 2, tr)
    assert isinstance(next(again), DeviceBatch)
    again.close()
    # ... and a prefetcher that is created but never pulled gives them back when it is dropped
    unused = prefetch_to_device(batches(600, 1), 2, tr)
    del unused
    import gc
    gc.collect()
    last = prefetch_to_device(batches(700, 1), 2, tr)
    assert isinstance(next(last), DeviceBatch)
    last.close()


def test_prefetcher_enqueues_the_next_batch_encoder_pass_behind_its_upload(monkeypatch):
    """Round 6 (row N1): with a frozen encoder attached and FRAMES in the batches, the prefetch worker enqueues the slot's encoder pass (arp_dt_encode_ahead) right
    behind that slot's upload -- never before it, never for encodings-in batches, and not at all with ARP_DT_ENCODE_AHEAD=0.  Stub trainer, no GPU."""
    from arp_amd.train import DeviceBatch, PolicyConfig, prefetch_to_device

    class Stub:
        cfg = PolicyConfig(emb=8, depth=1, heads=2, window=2, enc_tokens=2, enc_dim=4)

        def __init__(self, encoder):
            self._encoder = encoder
            self.log = []

        def upload_async(self, slot, enc, act, rtg, images=False):
            self.log.append(("upload", slot, bool(images)))

        def encode_ahead(self, slot):
            self.log.append(("encode", slot))

    def frames(n):
        for i in range(n):
            yield {"image": {"ob": np.full((2, 2, 8, 8, 3), i, np.float32)}, "action": np.zeros((2, 2), np.int32), "rtg": {"ob": np.zeros((2, 2, 1), np.float32)}}

    def encodings(n):
        for i in range(n):
            yield {"image": {"ob": np.full((2, 2, 2, 4), i, np.float32)}, "action": np.zeros((2, 2), np.int32), "rtg": {"ob": np.zeros((2, 2, 1), np.float32)}}

    def drain(tr, gen):
        it = prefetch_to_device(gen, 2, tr)
        for b in it:
            assert isinstance(b, DeviceBatch)
            b.done()
        it.close()
        return tr.log

    log = drain(Stub(encoder=object()), frames(4))
    ups = [e for e in log if e[0] == "upload"]
    assert len(ups) == 4 and all(e[2] for e in ups)
    for i, e in enumerate(log):  # every upload of frames is followed IMMEDIATELY by the encoder pass of the same slot
        if e[0] == "upload":
            assert log[i + 1] == ("encode", e[1]), log
    assert [e[1] for e in ups] == [0, 1, 0, 1]
    assert not any(e[0] == "encode" for e in drain(Stub(encoder=object()), encodings(3)))  # encodings in: nothing to encode
    assert not any(e[0] == "encode" for e in drain(Stub(encoder=None), frames(3)))        # no encoder attached
    monkeypatch.setenv("ARP_DT_ENCODE_AHEAD", "0")
    assert not any(e[0] == "encode" for e in drain(Stub(encoder=object()), frames(3)))     # switched off: the encoder runs at the head of its own step


def test_alibi_slopes_known_answers():
    """_get_attention_slopes (arp_dt/layers.py:97-110) in the oracle: the published ALiBi slopes -- 1/2 ... 1/256 for 8 heads, 2^(-8 i / n) in general for a
    power of two, and for 12 heads the 8-head slopes followed by every other 16-head slope."""
    from oracle import arpdt_torch as O
    assert np.allclose(O.alibi_slopes(8), [2.0 ** -(i + 1) for i in range(8)], rtol=1e-12)
    assert np.allclose(O.alibi_slopes(4), [2.0 ** -(2 * (i + 1)) for i in range(4)], rtol=1e-12)
    s16 = [2.0 ** -(0.5 * (i + 1)) for i in range(16)]
    assert np.allclose(O.alibi_slopes(16), s16, rtol=1e-12)
    assert np.allclose(O.alibi_slopes(12), [2.0 ** -(i + 1) for i in range(8)] + s16[0::2][:4], rtol=1e-12)
