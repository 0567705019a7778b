"""GPU parity of the whole labelling pass (preprocess -> ViT -> text tower -> reward) against the
numpy oracle, through the C ABI.  Tolerances: north_star = cosine within 1e-4 in f32; the stored
reward is exp(logit_scale) * cos = 100 * cos, so 1e-2 on the reward."""
import numpy as np
import pytest

from conftest import TINY

pytestmark = pytest.mark.gpu

COS_TOL_F32 = 1e-4   # north_star tolerance (parity mode)
COS_TOL_F16 = 1e-4   # default labelling mode (IEEE-half MFMA operands): meets north_star's tolerance at the bf16 rate
COS_TOL_BF16 = 3e-3  # bf16 GEMM operands (8-bit significands); measured error is reported by bench.py


def _setup(cfg_kw, n, seed, use_crop=False, H=256, W=256):
    from arp_amd import synth
    from oracle import clip_np as C
    ocfg = C.ClipConfig(**cfg_kw)
    Wt = synth.clip_weights(ocfg, seed=seed)
    fr = synth.procgen_like_frames(n, H, W, seed=seed + 1)
    tok = synth.prompt_tokens(2, [7, 3], ctx=ocfg.ctx, vocab=ocfg.vocab, seed=seed + 2)
    ref = C.compute_reward(Wt, ocfg, fr, tok, use_crop=use_crop)
    return ocfg, Wt, fr, tok, ref


@pytest.mark.parametrize("mode,tol", [("f32", COS_TOL_F32), ("f16", 2 * COS_TOL_F16), ("bf16", COS_TOL_BF16)])  # tiny: K = 64 rows, noisier
@pytest.mark.parametrize("attn_impl", [0, 1])
def test_tiny_end_to_end(gpu_lib, mode, tol, attn_impl):
    from arp_amd import clip
    ocfg, Wt, fr, tok, ref = _setup(TINY, 5, seed=11)
    m = clip.ClipLabeller(clip.ClipConfig(**TINY), Wt, mode=mode, attn_impl=attn_impl).set_text(tok)
    got = m.label(fr)
    scale = float(np.exp(Wt["logit_scale"]))
    err = np.abs(got - ref).max() / scale
    assert err < tol, f"tiny {mode}: cosine err {err}; got {got}, ref {ref}"
    m.close()


def test_tiny_features_and_text(gpu_lib):
    from arp_amd import clip
    from oracle import clip_np as C, preprocess as P
    ocfg, Wt, fr, tok, _ = _setup(TINY, 3, seed=5)
    Wd = C.cast_weights(Wt, np.float64)
    f_ref = C.encode_image(Wd, ocfg, P.preprocess(fr).astype(np.float64))
    t_ref = C.l2n(C.encode_text(Wd, ocfg, tok))
    m = clip.ClipLabeller(clip.ClipConfig(**TINY), Wt, mode="f32").set_text(tok)
    f = m.encode_image(fr)
    assert np.abs(f - f_ref).max() < 1e-4 * np.abs(f_ref).max()
    fn = m.encode_image(fr, normalize=True)
    assert np.abs(fn - C.l2n(f_ref)).max() < 1e-5
    assert np.abs(m.text_features() - t_ref).max() < 1e-5
    # goal-conditioned variant (label_reward.py:148-163): -||f_i - f_last||
    goal_ref = -np.linalg.norm(f_ref - f_ref[-1], axis=1)
    goal = -np.linalg.norm(f - f[-1], axis=1)
    assert np.abs(goal - goal_ref).max() < 1e-3 * max(1.0, np.abs(goal_ref).max())
    m.close()


def test_chunking_and_crop(gpu_lib):
    """max_batch smaller than n, ragged last chunk, use_crop transform, n = 1."""
    from arp_amd import clip
    ocfg, Wt, fr, tok, ref = _setup(TINY, 7, seed=21, use_crop=True)
    m = clip.ClipLabeller(clip.ClipConfig(**TINY), Wt, mode="f32", max_batch=3).set_text(tok)
    got = m.label(fr, use_crop=True)
    scale = float(np.exp(Wt["logit_scale"]))
    assert np.abs(got - ref).max() / scale < COS_TOL_F32
    one = m.label(fr[:1], use_crop=True)
    assert abs(one[0] - ref[0]) / scale < COS_TOL_F32
    assert m.label(fr[:0]).shape == (0,)
    m.close()


@pytest.mark.parametrize("name,n", [("ViT-B/32", 4), ("ViT-B/16", 2)])
def test_full_size_parity(gpu_lib, name, n):
    """Full ViT-B geometry, seeded random-init weights: f32 AND f16 (the default, timed by bench.py) modes within 1e-4
    cosine of the oracle; bf16 mode within its stated tolerance."""
    from arp_amd import clip, synth
    from oracle import clip_np as C
    cfg = clip.MODELS[name]
    ocfg = C.ClipConfig(patch=cfg.patch)
    Wt = synth.clip_weights(ocfg, seed=0)
    fr = synth.procgen_like_frames(n, seed=1)
    tok = synth.prompt_tokens(1, 8, seed=2)
    ref = C.compute_reward(Wt, ocfg, fr, tok)
    scale = 100.0
    for mode, tol in (("f32", COS_TOL_F32), ("f16", COS_TOL_F16), ("bf16", COS_TOL_BF16)):
        m = clip.ClipLabeller(cfg, Wt, mode=mode).set_text(tok)
        got = m.label(fr)
        err = np.abs(got - ref).max() / scale
        print(f"{name} {mode}: cosine err {err:.3e}  got {got} ref {ref}")
        assert err < tol, f"{name} {mode}: cosine err {err}"
        m.close()


def test_error_paths(gpu_lib):
    from arp_amd import clip, synth, _ffi
    from oracle import clip_np as C
    ocfg = C.ClipConfig(**TINY)
    Wt = synth.clip_weights(ocfg, seed=0)
    bad = dict(Wt)
    del bad["visual.proj"]
    with pytest.raises(_ffi.ArpError, match="missing weight"):
        clip.ClipLabeller(clip.ClipConfig(**TINY), bad, mode="f32")
    m = clip.ClipLabeller(clip.ClipConfig(**TINY), Wt, mode="f32")
    with pytest.raises(_ffi.ArpError, match="no prompt"):
        m.label(np.zeros((1, 256, 256, 3), np.uint8))
    with pytest.raises(ValueError):
        m.label(np.zeros((1, 256, 256, 4), np.uint8))
    m.close()


def test_online_single_frame_reward(gpu_lib):
    """get_torch_clip_reward (envs/vl_reward.py:11-23): the N = 1 path equals the batched one."""
    from arp_amd import clip, label_reward as L
    ocfg, Wt, fr, tok, ref = _setup(TINY, 3, seed=31)
    m = clip.ClipLabeller(clip.ClipConfig(**TINY), Wt, mode="f32").set_text(tok)
    scale = float(np.exp(Wt["logit_scale"]))
    for i in range(3):
        r = L.get_torch_clip_reward(m, fr[i])
        assert r.shape == (1,) and abs(r[0] - ref[i]) / scale < COS_TOL_F32
    m.close()


@pytest.mark.parametrize("name", ["ViT-B/32", "ViT-B/16"])
def test_latency_path_full_size(gpu_lib, name, monkeypatch):
    """Row N4 at the real geometry: single-frame calls (skinny GEMMs with LayerNorm folded into them, small preprocess tiles, the
    pass replayed as a hipGraph, pinned staging) within north_star's 1e-4 cosine of the fp32 oracle in f16 mode; the graph replays the
    same bits as launch-by-launch; a call of more token rows than the path's limit (1024) leaves it; a new prompt drops the
    captured passes."""
    from arp_amd import clip, synth, label_reward as L
    from oracle import clip_np as C
    cfg = clip.MODELS[name]
    ocfg = C.ClipConfig(patch=cfg.patch)
    Wt = synth.clip_weights(ocfg, seed=0)
    fr = synth.procgen_like_frames(6, seed=1)
    tok = synth.prompt_tokens(1, 8, seed=2)
    tok2 = synth.prompt_tokens(1, 6, seed=9)
    ref, ref2 = C.compute_reward(Wt, ocfg, fr, tok), C.compute_reward(Wt, ocfg, fr[:2], tok2)
    m = clip.ClipLabeller(cfg, Wt, mode="f16", n_streams=1).set_text(tok)
    one = np.concatenate([L.get_torch_clip_reward(m, fr[i]) for i in range(6)])
    again = np.concatenate([L.get_torch_clip_reward(m, fr[i]) for i in range(6)])  # replays
    assert (one == again).all()
    err = np.abs(one - ref).max() / 100.0
    print(f"{name} single-frame latency path: cosine err {err:.2e}")
    assert err < COS_TOL_F16
    m.profile(True)  # profiling: launch by launch, one event pair per site
    prof = np.concatenate([m.label(fr[i:i + 1]) for i in range(6)])
    sites = m.profile_read(); m.profile(False)
    # five launches per block: LayerNorm folded into the consumer GEMMs (no reduce + LayerNorm kernels, no fused QKV + attention kernel)
    assert (prof == one).all() and "vit.proj_reduce_ln_1" not in sites and "vit.qkv_attn" not in sites and "vit.c_proj" in sites
    assert sites["vit.qkv"]["calls"] == 6 * cfg.layers and "vit.ln_1" not in sites  # the last block's in_proj is a folded consumer too
    big = m.label(np.concatenate([fr] * 4))  # 24 frames: 1200 / 4728 token rows, past the path's row limit: the throughput kernels
    assert np.abs(big - np.concatenate([ref] * 4)).max() / 100.0 < COS_TOL_F16
    assert np.abs(m.label(fr[:3]) - ref[:3]).max() / 100.0 < COS_TOL_F16  # 150 / 591 rows: several frames per call on the latency path
    m.set_text(tok2)
    assert np.abs(np.concatenate([m.label(fr[i:i + 1]) for i in range(2)]) - ref2).max() / 100.0 < COS_TOL_F16
    m.close()
    # the unfolded latency path (split-K slabs + reduce / residual / LayerNorm kernels): seven launches per block, same tolerance
    monkeypatch.setenv("ARP_LAT_FOLD", "0")
    m = clip.ClipLabeller(cfg, Wt, mode="f16", n_streams=1).set_text(tok)
    m.profile(True)
    unf = np.concatenate([m.label(fr[i:i + 1]) for i in range(6)])
    assert "vit.proj_reduce_ln_1" in m.profile_read() and np.abs(unf - ref).max() / 100.0 < COS_TOL_F16
    m.close()
    monkeypatch.setenv("ARP_SKINNY", "0"); monkeypatch.setenv("ARP_CLIP_GRAPH", "0"); monkeypatch.setenv("ARP_CLIP_PINNED", "0")
    m = clip.ClipLabeller(cfg, Wt, mode="f16", n_streams=1).set_text(tok)
    old = np.concatenate([m.label(fr[i:i + 1]) for i in range(6)])
    m.close()
    assert np.abs(old - ref).max() / 100.0 < COS_TOL_F16 and np.abs(old - one).max() / 100.0 < COS_TOL_F16


def test_latency_path_tiny_bf16_and_crop(gpu_lib):
    """The same path on the tiny geometry (K = 64: one wave per workgroup, split 1 and 2), bf16, with the use_crop transform and a few
    frames per call."""
    from arp_amd import clip
    ocfg, Wt, fr, tok, ref = _setup(TINY, 6, seed=41, use_crop=True)
    scale = float(np.exp(Wt["logit_scale"]))
    for mode, tol in (("bf16", COS_TOL_BF16), ("f16", 2 * COS_TOL_F16)):
        m = clip.ClipLabeller(clip.ClipConfig(**TINY), Wt, mode=mode, n_streams=1).set_text(tok)
        got = np.concatenate([m.label(fr[i:i + 2], use_crop=True) for i in (0, 2, 4)])
        assert np.abs(got - ref).max() / scale < tol
        m.close()
    # other frame geometries through the small-tile preprocess plan of a latency-path call: native Procgen size, non-square
    for (H, W_) in ((64, 64), (96, 128), (300, 200)):
        ocfg, Wt, fr, tok, ref = _setup(TINY, 3, seed=43, H=H, W=W_)
        m = clip.ClipLabeller(clip.ClipConfig(**TINY), Wt, mode="f16", n_streams=1).set_text(tok)
        got = np.concatenate([m.label(fr[i:i + 1]) for i in range(3)])
        assert np.abs(got - ref).max() / float(np.exp(Wt["logit_scale"])) < 2 * COS_TOL_F16, (H, W_)
        assert (np.concatenate([m.label(fr[i:i + 1]) for i in range(3)]) == got).all()  # replayed
        m.close()


def test_interleaved_calls_on_one_handle_repeat_bit_for_bit(gpu_lib):
    """A short version of scripts/soak_label.py: single frames (captured graphs, pinned staging), a few frames, batches on two streams,
    the asynchronous pair with other calls while both slots are in flight, prompt changes, crops and feature calls, interleaved at random
    on ONE handle; every repeat of a call must give the bits of its first occurrence (stale captures, workspace growth under live
    graphs, slot reuse would show here)."""
    from arp_amd import clip, synth
    cfg = clip.ClipConfig(**MID)
    m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=5), mode="f16", max_batch=256, n_streams=2)
    toks = [synth.prompt_tokens(1, 5, ctx=cfg.ctx, vocab=cfg.vocab, seed=4), synth.prompt_tokens(1, 7, ctx=cfg.ctx, vocab=cfg.vocab, seed=8)]
    base = synth.procgen_like_frames(32, seed=3)
    big = np.ascontiguousarray(np.tile(base, (8, 1, 1, 1)))
    ref, rng, prompt = {}, np.random.default_rng(1), 0

    def check(key, val):
        if key not in ref:
            ref[key] = val.copy()
        assert np.array_equal(ref[key], val), key

    m.set_text(toks[prompt])
    for _ in range(400):
        kind = int(rng.integers(0, 7))
        if kind == 0:
            i = int(rng.integers(0, 32)); check((prompt, "one", i), m.label(base[i:i + 1]))
        elif kind == 1:
            n = int(rng.integers(2, 21)); check((prompt, "few", n), m.label(base[:n]))
        elif kind == 2:
            n = int(rng.choice([21, 64, 130])); check((prompt, "mid", n), m.label(big[:n]))
        elif kind == 3:
            check((prompt, "big"), m.label(big))
        elif kind == 4:
            m.label_submit(0, big); m.label_submit(1, big[:128])
            if rng.integers(0, 2):
                check((prompt, "one", 5), m.label(base[5:6])); check((prompt, "mid", 64), m.label(big[:64]))
            check((prompt, "big"), m.label_collect(0)); check((prompt, "half"), m.label_collect(1))
        elif kind == 5:
            prompt ^= 1; m.set_text(toks[prompt])
        else:
            check((prompt, "crop", 1), m.label(base[:1], use_crop=True)); check(("enc", 3), m.encode_image(base[:3]))
    assert len(ref) > 40
    m.close()


def test_two_stream_split_matches_single_stream(gpu_lib):
    """n_streams = 2 labels the two halves of a batch on two HIP streams; results are bit-identical to one stream."""
    from arp_amd import clip, synth
    from oracle import clip_np as C
    ocfg = C.ClipConfig(**TINY)
    Wt = synth.clip_weights(ocfg, seed=3)
    tok = synth.prompt_tokens(1, 5, ctx=ocfg.ctx, vocab=ocfg.vocab, seed=4)
    fr = synth.procgen_like_frames(301, 64, 64, seed=5)  # odd count: halves of 150 and 151
    a = clip.ClipLabeller(clip.ClipConfig(**TINY), Wt, mode="bf16", n_streams=1).set_text(tok)
    b = clip.ClipLabeller(clip.ClipConfig(**TINY), Wt, mode="bf16", n_streams=2).set_text(tok)
    ra, rb = a.label(fr), b.label(fr)
    assert (ra == rb).all()
    b.profile(True)
    rb2 = b.label(fr)
    assert (rb2 == ra).all() and b.profile_read()["vit.qkv"]["calls"] == 2 * TINY["layers"]
    ref = C.compute_reward(Wt, ocfg, fr[:6], tok)
    assert np.abs(ra[:6] - ref).max() / float(np.exp(Wt["logit_scale"])) < COS_TOL_BF16
    a.close(); b.close()


MID = dict(patch=32, width=128, layers=3, heads=2, embed=64, img_res=224, txt_width=64, txt_layers=2, txt_heads=2, ctx=77, vocab=512)


def test_layernorm_fold_matches_unfused(gpu_lib, monkeypatch):
    """bf16 mode folds LayerNorm into the consumer GEMMs (gamma into W, mean/rstd from the residual GEMM's epilogue
    partial sums).  Folded and unfused paths must agree with each other and with the oracle; LN scale/bias are far
    from identity and the rows have a large mean so the mean-subtraction term is exercised.  (ARP_SKINNY=0: the 9-frame call would
    otherwise take the latency path, which has no separate ln_1 launches to look for.)"""
    from arp_amd import clip, synth
    from oracle import clip_np as C
    monkeypatch.setenv("ARP_SKINNY", "0")
    ocfg = C.ClipConfig(**MID)
    Wt = synth.clip_weights(ocfg, seed=17)
    rng = np.random.default_rng(3)
    for k in list(Wt):
        if k.startswith("visual") and (".ln_1." in k or ".ln_2." in k):
            Wt[k] = (Wt[k] + (0.5 * rng.standard_normal(Wt[k].shape) if k.endswith("weight") else 0.3 * rng.standard_normal(Wt[k].shape))).astype(np.float32)
    Wt["visual.class_embedding"] = (Wt["visual.class_embedding"] + 0.5).astype(np.float32)   # rows with a large mean
    Wt["visual.ln_pre.bias"] = (Wt["visual.ln_pre.bias"] + 0.7).astype(np.float32)
    tok = synth.prompt_tokens(1, 5, ctx=ocfg.ctx, vocab=ocfg.vocab, seed=4)
    fr = synth.procgen_like_frames(9, seed=5)
    ref = C.compute_reward(Wt, ocfg, fr, tok)
    scale = float(np.exp(Wt["logit_scale"]))
    out = {}
    for fold in ("0", "1"):
        monkeypatch.setenv("ARP_LN_FOLD", fold)
        m = clip.ClipLabeller(clip.ClipConfig(**MID), Wt, mode="bf16").set_text(tok)
        m.profile(True)
        out[fold] = m.label(fr)
        sites = m.profile_read()
        assert ("vit.ln_1" in sites) == (fold == "0") and ("vit.ln_stats" in sites) == (fold == "1")
        m.close()
    e0, e1 = np.abs(out["0"] - ref).max() / scale, np.abs(out["1"] - ref).max() / scale
    print(f"cosine err unfused {e0:.2e}, folded {e1:.2e}, folded vs unfused {np.abs(out['0'] - out['1']).max() / scale:.2e}")
    assert e0 < COS_TOL_BF16 and e1 < COS_TOL_BF16


@pytest.mark.parametrize("mode,tol", [("f16", COS_TOL_F16), ("bf16", COS_TOL_BF16)])
def test_full_batch_properties(gpu_lib, mode, tol):
    """BASELINE configs[1] at full size (1024 frames, ViT-B/32, the default f16 mode and bf16), through properties that need no oracle at that size:
    a frame's reward does not depend on what else is in the batch or where it sits (permutation / duplication), on the number
    of streams, or on the run (bit-identical repeats), and a 16-frame sample agrees with the oracle."""
    from arp_amd import clip, synth
    from oracle import clip_np as C
    cfg = clip.MODELS["ViT-B/32"]
    W = synth.clip_weights(C.ClipConfig(patch=cfg.patch), seed=0)
    tok = synth.prompt_tokens(1, 8, seed=2)
    fr = synth.procgen_like_frames(1024, seed=5)
    fr[777] = fr[3]  # a duplicate far away in the batch
    m = clip.ClipLabeller(cfg, W, mode=mode, max_batch=1024, n_streams=2).set_text(tok)
    r = m.label(fr)
    assert r.shape == (1024,) and np.isfinite(r).all()
    assert np.array_equal(r, m.label(fr))                      # deterministic
    assert r[777] == r[3]                                       # position-independent
    perm = np.random.default_rng(0).permutation(1024)
    assert np.array_equal(m.label(fr[perm]), r[perm])           # batch-composition independent
    for ns in (1, 3):
        m.set_streams(ns)
        assert np.array_equal(m.label(fr), r)                   # stream-count independent
    assert np.array_equal(m.label(fr[:100]), r[:100])           # sub-batch (different GEMM grid, same per-row arithmetic)
    idx = np.arange(0, 1024, 64)
    ref = C.compute_reward(W, C.ClipConfig(patch=cfg.patch), fr[idx], tok)
    assert np.abs(r[idx] - ref).max() / 100.0 < tol
    m.close()


@pytest.mark.parametrize("mode", ["f32", "f16", "bf16"])
def test_last_block_class_token_only_is_exact(gpu_lib, monkeypatch, mode):
    """The vision tower's last block computes out_proj / ln_2 / MLP (and the attention's query side) for the class-token
    rows only -- the rows ln_post reads (arp_dt/models/openai/layers.py:330).  Rewards, features and the multi-scale
    class-token export must be bit-identical to running the block on every row (ARP_CLS_ONLY=0)."""
    from arp_amd import clip, synth
    from oracle import clip_np as C
    ocfg = C.ClipConfig(**MID)
    Wt = synth.clip_weights(ocfg, seed=23)
    tok = synth.prompt_tokens(1, 5, ctx=ocfg.ctx, vocab=ocfg.vocab, seed=4)
    fr = synth.procgen_like_frames(37, seed=6)
    out = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("ARP_CLS_ONLY", flag)
        m = clip.ClipLabeller(clip.ClipConfig(**MID), Wt, mode=mode, n_streams=1).set_text(tok)
        m.profile(True)
        out[flag] = (m.label(fr), m.encode_image(fr)) + tuple(m.encode_image_multiscale(fr))
        sites = m.profile_read()
        assert ("vit.c_fc_cls" in sites) == (flag == "1")
        assert sites["vit.c_fc"]["calls"] == 3 * (MID["layers"] - (flag == "1"))  # label + encode_image + multiscale
        m.close()
    for a, b in zip(out["0"], out["1"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("mode", ["f16", "bf16"])
def test_fused_qkv_attention_is_exact(gpu_lib, monkeypatch, mode):
    """16-bit modes at head_dim 64 / N <= 64 run the QKV projection and the attention as ONE kernel (csrc/qkvattn.h: a
    workgroup owns 5 whole frames x one head, q|k|v never leave the CU).  Every q/k/v value is the same MFMA chain rounded at
    the same point and the attention arithmetic is attn_mfma_kernel's, so rewards, features and the multi-scale export must be
    bit-identical to the two-kernel path (ARP_QKV_FUSED=0) -- on full tiles, a ragged last tile (37 = 7 x 5 + 2 frames), fewer
    frames than one tile, and with the class-token-only last block on and off.  (ARP_SKINNY=0: a 3-frame call would otherwise take the
    latency path, which never uses the fused kernel -- tests/test_clip_gpu.py::test_latency_path_*.)"""
    from arp_amd import clip, synth
    from oracle import clip_np as C
    monkeypatch.setenv("ARP_SKINNY", "0")
    ocfg = C.ClipConfig(**MID)
    Wt = synth.clip_weights(ocfg, seed=29)
    tok = synth.prompt_tokens(1, 5, ctx=ocfg.ctx, vocab=ocfg.vocab, seed=4)
    fr = synth.procgen_like_frames(37, seed=8)
    ref = C.compute_reward(Wt, ocfg, fr[:5], tok)
    for cls_only in ("1", "0"):
        monkeypatch.setenv("ARP_CLS_ONLY", cls_only)
        out = {}
        for flag in ("0", "1"):
            monkeypatch.setenv("ARP_QKV_FUSED", flag)
            m = clip.ClipLabeller(clip.ClipConfig(**MID), Wt, mode=mode, n_streams=1).set_text(tok)
            m.profile(True)
            out[flag] = (m.label(fr), m.label(fr[:3]), m.encode_image(fr)) + tuple(m.encode_image_multiscale(fr))
            sites = m.profile_read()
            assert ("vit.qkv_attn" in sites) == (flag == "1") and ("vit.attn" in sites) == (flag == "0"), sorted(sites)
            if cls_only == "1":
                assert ("vit.qkv_attn_cls" in sites) == (flag == "1")
            m.close()
        for a, b in zip(out["0"], out["1"]):
            assert np.array_equal(a, b)
        assert np.abs(out["1"][0][:5] - ref).max() / float(np.exp(Wt["logit_scale"])) < (2 * COS_TOL_F16 if mode == "f16" else COS_TOL_BF16)


def test_heavy_tailed_weights_f16_stays_finite_and_in_tolerance(gpu_lib):
    """VERDICT r1 item 3c.  Pretrained CLIP is not random-init: a handful of LayerNorm gains are ~30x the rest and some c_fc biases
    push QuickGELU inputs into the tens, so residual-stream rows reach the hundreds.  binary16 has 5 exponent bits: every 16-bit
    tensor of the f16 mode (LayerNorm outputs, q/k/v, softmax weights, QuickGELU outputs) must stay finite and the rewards must
    stay within tolerance of the fp64 oracle on such weights, at full ViT-B/32 size."""
    from arp_amd import clip, synth
    from oracle import clip_np as C
    cfg = clip.MODELS["ViT-B/32"]
    ocfg = C.ClipConfig(patch=cfg.patch)
    W = synth.clip_weights(ocfg, seed=0)
    rng = np.random.default_rng(7)
    for i in range(12):
        p = f"visual.transformer.resblocks.{i}."
        for ln in ("ln_1", "ln_2"):
            ch = rng.choice(768, 6, replace=False)
            W[p + ln + ".weight"][ch] *= 30.0                      # outlier channels
        b = W[p + "mlp.c_fc.bias"]
        ch = rng.choice(3072, 24, replace=False)
        b[ch] = rng.choice([-20.0, 20.0], 24).astype(np.float32)    # QuickGELU inputs in the tens, both signs
    W["visual.ln_pre.weight"][rng.choice(768, 4, replace=False)] *= 30.0
    och, osign = rng.choice(768, 3, replace=False), np.array([1.0, -1.0, 1.0], np.float32)
    for i in range(12):  # "massive activation" channels: the same three residual channels pushed by every block
        W[f"visual.transformer.resblocks.{i}.mlp.c_proj.bias"][och] += 15.0 * osign
    fr = synth.procgen_like_frames(6, seed=3)
    tok = synth.prompt_tokens(1, 8, seed=2)
    ref = C.compute_reward(W, ocfg, fr, tok)
    m32 = clip.ClipLabeller(cfg, W, mode="f32").set_text(tok)
    m16 = clip.ClipLabeller(cfg, W, mode="f16").set_text(tok)
    r32, r16 = m32.label(fr), m16.label(fr)
    f16 = m16.encode_image(fr)
    inter, final = m16.encode_image_multiscale(fr)   # the class-token row of the residual stream after every block
    assert np.isfinite(r16).all() and np.isfinite(f16).all() and np.isfinite(inter).all() and np.isfinite(final).all()
    print(f"heavy-tailed: |residual stream| up to {np.abs(inter).max():.1f}; cosine err f32 {np.abs(r32 - ref).max() / 100:.2e}, f16 {np.abs(r16 - ref).max() / 100:.2e}")
    assert np.abs(inter).max() > 100.0, "the stress case must actually stress the range"
    assert np.abs(r32 - ref).max() / 100.0 < COS_TOL_F32
    # north_star's 1e-4 itself (VERDICT r2 next #2c): pretrained CLIP IS the heavy-tailed case, so this is where "f16 meets 1e-4" counts
    assert np.abs(r16 - ref).max() / 100.0 < COS_TOL_F16
    m32.close(); m16.close()


def test_fp8_mlp_mode(gpu_lib):
    """BASELINE configs[4] names "fp8 MFMA GEMMs": the vision tower's c_fc / c_proj on e4m3 operands (v_mfma_scale_f32_16x16x128_f8f6f4).
    Two checks: (1) KERNEL correctness -- against the oracle with the same e4m3 roundings inserted (activations x 32 / x 16, weights x
    a per-tensor power of two) the features agree to 16-bit-mode accuracy; (2) the PRICE of three significand bits -- the feature
    error against the plain fp64 oracle is reported and bounded (a throughput mode for the frozen towers of the fine-tune step)."""
    from arp_amd import clip, synth
    from oracle import clip_np as C, preprocess as P
    kw = dict(MID, width=256, heads=4, layers=3)  # width % 128 == 0; K = 256 (c_fc) and 1024 (c_proj): 2 and 8 fp8 K-tiles
    ocfg = C.ClipConfig(**kw)
    Wt = synth.clip_weights(ocfg, seed=41)
    fr = synth.procgen_like_frames(40, seed=42)   # 2000 rows: several 256-row tiles, the last one ragged
    Wd = C.cast_weights(Wt, np.float64)
    x = P.preprocess(fr).astype(np.float64)
    f_ref = C.encode_image(Wd, ocfg, x)
    f_emul = C.encode_image(Wd, ocfg, x, mlp_fp8=True)
    m16 = clip.ClipLabeller(clip.ClipConfig(**kw), Wt, mode="f16", n_streams=1)
    m8 = clip.ClipLabeller(clip.ClipConfig(**kw), Wt, mode="f16", n_streams=1, fp8_mlp=True)
    m8.profile(True)
    f16, f8 = m16.encode_image(fr), m8.encode_image(fr)
    assert "vit.c_fc_fp8" in m8.profile_read() and np.isfinite(f8).all()
    cos = lambda a, b: float(np.min(np.sum(a * b, 1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))))
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    print(f"fp8 MLP: vs e4m3-emulating oracle rel {rel(f8, f_emul):.2e} (f16 path vs plain oracle {rel(f16, f_ref):.2e}); "
          f"vs plain oracle rel {rel(f8, f_ref):.2e}, min cosine {cos(f8, f_ref):.5f}")
    # The kernels themselves are checked EXACTLY in test_ops_gpu.py::test_gemm_fp8_exact_on_representable_operands.  End to end an
    # f16-level difference in a LayerNorm / QuickGELU output that crosses an e4m3 rounding boundary moves that value by a whole
    # step (6 %), so the product agrees with the emulating oracle only to a fraction of the quantisation error itself.
    assert rel(f8, f_emul) < 0.6 * rel(f_emul, f_ref) and rel(f8, f_emul) < 2e-2
    assert cos(f8, f_ref) > 0.995 and cos(f8, f_emul) > 0.9995
    m16.close(); m8.close()


def test_fp8_attention_projections(gpu_lib):
    """fp8 level 2 (arp_clip_set_fp8_mlp(h, 2)): in_proj and out_proj on e4m3 operands as well, in a geometry whose attention is NOT the
    fused kernel (patch 16: 197 tokens, the fine-tune step's ViT-B/16 towers).  Checked like the MLP mode: close to the oracle with the
    same e4m3 roundings inserted, and the price against the plain oracle reported and bounded."""
    from arp_amd import clip, synth
    from oracle import clip_np as C, preprocess as P
    kw = dict(MID, patch=16, width=256, heads=4, layers=3)
    ocfg = C.ClipConfig(**kw)
    Wt = synth.clip_weights(ocfg, seed=43)
    fr = synth.procgen_like_frames(10, seed=44)   # 1970 rows
    Wd = C.cast_weights(Wt, np.float64)
    x = P.preprocess(fr).astype(np.float64)
    f_ref = C.encode_image(Wd, ocfg, x)
    f_emul = C.encode_image(Wd, ocfg, x, mlp_fp8=True, attn_fp8=True)
    m8 = clip.ClipLabeller(clip.ClipConfig(**kw), Wt, mode="f16", n_streams=1, fp8_mlp=2)
    m8.profile(True)
    f8 = m8.encode_image(fr)
    sites = m8.profile_read()
    assert "vit.qkv_fp8" in sites and "vit.out_proj_fp8" in sites and "vit.c_fc_fp8" in sites and np.isfinite(f8).all()
    cos = lambda a, b: float(np.min(np.sum(a * b, 1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1))))
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    print(f"fp8 MLP + attention projections: vs e4m3-emulating oracle rel {rel(f8, f_emul):.2e}; vs plain oracle rel {rel(f8, f_ref):.2e}, "
          f"min cosine {cos(f8, f_ref):.5f} (emulation vs plain: rel {rel(f_emul, f_ref):.2e})")
    # (five e4m3 roundings per block instead of two: more values sit on a rounding boundary that 16-bit noise can cross, so the product
    #  and the emulation agree a little less closely than in the MLP-only mode; the kernels themselves are checked exactly in test_ops_gpu.py)
    assert rel(f8, f_emul) < 0.8 * rel(f_emul, f_ref) and rel(f8, f_emul) < 3e-2
    assert cos(f8, f_ref) > 0.99 and cos(f8, f_emul) > 0.999
    m8.close()


def test_label_reward_from_hdf5_file(gpu_lib, tmp_path):
    """SURVEY row N3 end to end on the GPU: recorder-style HDF5 file in, reward / rtg datasets out (gzip, chunks (1, num_frames)),
    equal to labelling the same frames from memory."""
    h5store = pytest.importorskip("arp_amd.h5store")
    try:
        h5store.lib()
    except ImportError as e:
        pytest.skip(str(e))
    from arp_amd import clip, label_reward as L, synth
    from oracle import clip_np as C, rtg
    ocfg = C.ClipConfig(**TINY)
    Wt = synth.clip_weights(ocfg, seed=3)
    tok = synth.prompt_tokens(1, 5, ctx=ocfg.ctx, vocab=ocfg.vocab, seed=4)
    lens, F = [3, 11, 8, 1], 8
    frames = synth.procgen_like_frames(sum(lens), 64, 64, seed=9)
    ob, done, s = [], [], 0
    for n in lens:
        idx = np.clip(np.arange(n)[:, None] + np.arange(-F + 1, 1)[None, :], 0, None)
        d = np.zeros((n, F), np.float32); d[-1, -1] = 1
        ob.append(frames[s : s + n][idx]); done.append(d); s += n
    ob, done = np.concatenate(ob), np.concatenate(done)
    p = str(tmp_path / "data.hdf5")
    with h5store.H5Store(p, "w") as f:
        f.create_dataset("ob", data=ob, compression="gzip", chunks=(1, F, 64, 64, 3), maxshape=(None, F, 64, 64, 3))
        f.create_dataset("done", data=done, compression="gzip", chunks=(1, F), maxshape=(None, F))
    m = clip.ClipLabeller(clip.ClipConfig(**TINY), Wt, mode="f32").set_text(tok)
    L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=p, clip_model=m, tokens=tok)
    ref = rtg.label_file({"ob": ob, "done": done}, lambda im: m.label(im))
    with h5store.H5Store(p, "r") as f:
        for k, v in ref.items():
            assert f[k].chunks == (1, F) and f[k].compression == "gzip" and np.array_equal(f[k][...], v), k
    orc = C.compute_reward(Wt, ocfg, frames[:4], tok)
    with h5store.H5Store(p, "r") as f:
        got = np.concatenate([f["ob_clip_reward"][0:3, -1], f["ob_clip_reward"][3:4, -1]])
    assert np.abs(got - orc).max() / float(np.exp(Wt["logit_scale"])) < COS_TOL_F32
    m.close()


def test_pipelined_submit_collect_equals_synchronous_label(gpu_lib):
    """arp_clip_label_submit / _collect (two host-fed calls in flight): rewards bit-identical to arp_clip_label, any interleaving of the
    two slots, ragged sizes, pinned and pageable sources; misuse (busy slot, empty slot, oversize) is an error, not a hang."""
    from arp_amd import clip, synth
    from arp_amd._ffi import ArpError
    from oracle import clip_np as C
    kw = dict(MID)
    W = synth.clip_weights(C.ClipConfig(**kw), seed=5)
    m = clip.ClipLabeller(clip.ClipConfig(**kw), W, mode="f16", max_batch=300, n_streams=2).set_text(synth.prompt_tokens(1, 6, vocab=kw["vocab"], seed=6))
    sets = [synth.procgen_like_frames(n, seed=10 + i) for i, n in enumerate((300, 17, 256, 129, 300))]
    ref = [m.label(f) for f in sets]
    m.pin_host(sets[2])
    got = [None] * len(sets)
    m.label_submit(0, sets[0])
    for i in range(1, len(sets)):
        m.label_submit(i & 1, sets[i])
        got[i - 1] = m.label_collect((i - 1) & 1)
    got[-1] = m.label_collect((len(sets) - 1) & 1)
    m.unpin_host(sets[2])
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    m.label_submit(0, sets[1])
    with pytest.raises(ArpError):
        m.label_submit(0, sets[1])          # the slot is busy
    m.label_collect(0)
    with pytest.raises(ArpError):
        m.label_collect(1)                  # nothing submitted
    with pytest.raises(ArpError):
        m.label_submit(1, synth.procgen_like_frames(301, seed=3))  # more than max_batch
    m.close()


def test_online_reward_family_full_vit_b16(gpu_lib):
    """Row N4, the rest of the rollout loop's dispatch (envs/rollout_procgen.py:133-151) at the model the reference loads (ViT-B/16, f16 operands,
    latency path): get_torch_clip_reward with a LIST of prompts = the mean over prompts of the logits (vl_reward.py:19-22) and with one prompt =
    prompt 0; get_torch_clip_goal_conditioned_reward = -||f(obs) - f(goal)|| on un-normalised features, two frames per call (:26-41), with the
    reference's crop-of-the-cropped goal under use_crop.  Oracle: oracle/clip_np in fp64."""
    from arp_amd import clip, synth, label_reward as L
    from oracle import clip_np as C, preprocess as PP
    cfg = clip.MODELS["ViT-B/16"]
    ocfg = C.ClipConfig(patch=cfg.patch)
    W = synth.clip_weights(ocfg, seed=0)
    fr = synth.procgen_like_frames(4, seed=11)
    tok3 = synth.prompt_tokens(3, [8, 5, 11], seed=12)
    Wd = C.cast_weights(W, np.float64)
    feat = C.encode_image(Wd, ocfg, PP.preprocess(fr).astype(np.float64))
    txt = C.encode_text(Wd, ocfg, tok3)
    logits = np.exp(float(W["logit_scale"])) * (C.l2n(txt) @ C.l2n(feat).T)  # [3 prompts, 4 frames]
    m = clip.ClipLabeller(cfg, W, mode="f16", n_streams=1).set_text(tok3)
    scale = 100.0
    for i in range(4):
        r_list = L.get_torch_clip_reward(m, fr[i], ["a", "b", "c"])
        r_one = L.get_torch_clip_reward(m, fr[i], "a")
        assert r_list.shape == (1,) and r_one.shape == (1,)
        assert abs(r_list[0] - logits[:, i].mean()) / scale < COS_TOL_F16, (i, r_list, logits[:, i].mean())
        assert abs(r_one[0] - logits[0, i]) / scale < COS_TOL_F16
    assert np.array_equal(m.label(fr), np.concatenate([L.get_torch_clip_reward(m, fr[i]) for i in range(4)]))  # back on prompt 0
    with pytest.raises(ValueError):
        L.get_torch_clip_reward(m, fr[0], ["only", "two"])
    # a token array as pos_text re-encodes the prompts and averages over them
    r_tok = L.get_torch_clip_reward(m, fr[0], tok3[:2])
    assert abs(r_tok[0] - logits[:2, 0].mean()) / scale < COS_TOL_F16
    m.set_text(tok3)
    # goal-conditioned: un-normalised features, python float
    fn = np.linalg.norm(feat, axis=1).mean()
    for i in (0, 1, 2):
        got = L.get_torch_clip_goal_conditioned_reward(m, fr[i], fr[3])
        ref = -np.linalg.norm(feat[i] - feat[3])
        assert isinstance(got, float) and abs(got - ref) < 2e-3 * fn, (got, ref, fn)
    assert L.get_torch_clip_goal_conditioned_reward(m, fr[3], fr[3]) == 0.0
    # use_crop: obs -> centre 128 x 128, goal -> centre 64 x 64 (the reference sizes the second crop from the cropped obs)
    o, g = fr[0][64:192, 64:192], fr[3][96:160, 96:160]
    fo = C.encode_image(Wd, ocfg, PP.preprocess(o[None]).astype(np.float64))[0]
    fg = C.encode_image(Wd, ocfg, PP.preprocess(g[None]).astype(np.float64))[0]
    got = L.get_torch_clip_goal_conditioned_reward(m, fr[0], fr[3], use_crop=True)
    assert abs(got + np.linalg.norm(fo - fg)) < 2e-3 * fn, (got, np.linalg.norm(fo - fg))
    assert set(L.VL_REWARD_FNS) == {"clip", "clip_goal_conditioned", "clip_ft", "clip_ft_goal_conditioned"}
    m.close()
