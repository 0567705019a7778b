"""GPU parity of SURVEY row N2 (the CLIP multi-scale adapter fine-tune head) through the C ABI: against the fixture produced
by the reference class itself (tests/golden/finetune_tiny.npz) and against the torch oracle on seeded cases."""
import numpy as np
import pytest

from test_finetune_oracle import load_golden

pytestmark = pytest.mark.gpu

MID = dict(layers=3, width_v=128, width_t=64, embed=64, hidden=64, n_actions=15)


def _trainer(cfg_o, mode):
    from arp_amd.finetune import FinetuneConfig, FinetuneTrainer
    cfg = FinetuneConfig(layers=cfg_o.layers, width_v=cfg_o.width_v, width_t=cfg_o.width_t, embed=cfg_o.embed, hidden=cfg_o.hidden,
                         n_actions=cfg_o.n_actions, gamma=cfg_o.gamma, logit_scale=cfg_o.logit_scale, use_vip=cfg_o.use_vip, use_id=cfg_o.use_id)
    return FinetuneTrainer(cfg, mode=mode)


def _rel_errs(got, ref):
    return {k: float(np.abs(got[k] - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-6)) for k in ref}


def test_matches_reference_class_fixture(gpu_lib):
    cfg, P, G, batch, g = load_golden()
    tr = _trainer(cfg, "f32")
    tr.set_params(P)
    assert tr.shapes == {k: tuple(v.shape) for k, v in P.items()}
    tr.set_batch(*batch)
    out = tr.forward()
    assert abs(out["loss"] - float(g["loss"])) < 5e-5
    assert abs(out["vip_loss"] - float(g["vip_loss"])) < 5e-5
    assert abs(out["lambda_id"] * out["id_loss"] - float(g["lambda_id_times_id_loss"])) < 5e-5
    tr.backward()
    errs = _rel_errs(tr.get_grads(), G)
    bad = {k: v for k, v in errs.items() if not v < 5e-4}
    assert not bad, bad
    tr.close()


@pytest.mark.parametrize("use_vip,use_id", [(True, True), (True, False), (False, True)])
def test_gradients_match_oracle(gpu_lib, use_vip, use_id):
    from arp_amd import finetune as FT
    from oracle import finetune_torch as O
    cfg = O.HeadConfig(**MID, use_vip=use_vip, use_id=use_id)
    P = O.init_params(cfg, seed=3)
    P["image_residual_weight"] = np.float32(0.4) * np.ones((), np.float32)
    P["text_residual_weight"] = np.float32(-0.6) * np.ones((), np.float32)
    batch = FT.synth_batch(FT.FinetuneConfig(**MID), 7, seed=4)
    g_ref, aux = O.grads(P, cfg, batch)
    tr = _trainer(cfg, "f32")
    tr.set_params(P)
    tr.set_batch(*batch)
    out = tr.forward()
    for k in ("loss", "vip_loss", "id_loss"):
        assert abs(out[k] - aux[k]) < 5e-5, (k, out[k], aux[k])
    tr.backward()
    errs = _rel_errs(tr.get_grads(), g_ref)
    bad = {k: v for k, v in errs.items() if not v < 3e-4}
    assert not bad, bad
    tr.close()


@pytest.mark.parametrize("mode,ltol,cmin", [("bf16", 5e-2, 0.99), ("f16", 5e-3, 0.9999)])
def test_16bit_modes_track_oracle(gpu_lib, mode, ltol, cmin):
    """f16 (round 2: IEEE-half operands, gradients seeded with a power-of-two scale that AdamW / get_tensor take out again) is
    ~10x closer to the fp64 oracle than bf16 at the same MFMA rate."""
    from arp_amd import finetune as FT
    from oracle import finetune_torch as O
    cfg = O.HeadConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64)
    P = O.init_params(cfg, seed=5)
    batch = FT.synth_batch(FT.FinetuneConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64), 6, seed=6)
    g_ref, aux = O.grads(P, cfg, batch)
    tr = _trainer(cfg, mode)
    tr.set_params(P)
    tr.set_batch(*batch)
    out = tr.forward()
    print(f"{mode}: loss err {abs(out['loss'] - aux['loss']):.2e}")
    assert abs(out["loss"] - aux["loss"]) < ltol * max(1.0, abs(aux["loss"]))
    tr.backward()
    g = tr.get_grads()
    num = sum(float(g[k].ravel().astype(np.float64) @ g_ref[k].ravel()) for k in g_ref)
    den = np.sqrt(sum(float((g[k].astype(np.float64) ** 2).sum()) for k in g_ref) * sum(float((g_ref[k] ** 2).sum()) for k in g_ref))
    nr = np.sqrt(sum(float((g[k].astype(np.float64) ** 2).sum()) for k in g_ref) / sum(float((g_ref[k] ** 2).sum()) for k in g_ref))
    print(f"{mode}: gradient cosine {num / den:.6f}, norm ratio {nr:.5f}")
    assert num / den > cmin and abs(nr - 1) < (5e-3 if mode == "f16" else 5e-2), (num / den, nr)
    tr.close()


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("f16", 1e-3), ("bf16", 5e-3)])
def test_train_steps_match_torch_adamw(gpu_lib, mode, tol):
    from arp_amd import finetune as FT
    from oracle import finetune_torch as O
    cfg = O.HeadConfig(**MID)
    P = O.init_params(cfg, seed=7)
    fcfg = FT.FinetuneConfig(**MID, weight_decay=0.01)
    batches = [FT.synth_batch(fcfg, 5, seed=8 + i) for i in range(2)]
    P_ref, aux_ref = O.train_steps(P, cfg, batches, 1e-3, 0.01, 3)
    tr = FT.FinetuneTrainer(FT.FinetuneConfig(**MID, weight_decay=0.01, logit_scale=cfg.logit_scale), mode=mode)
    tr.set_params(P)
    for i in range(3):
        tr.set_batch(*batches[i % 2])
        aux = tr.train_step(1e-3)
        assert abs(aux["loss"] - aux_ref[i]["loss"]) < (1e-4 if mode == "f32" else 0.1), (i, aux, aux_ref[i])
    assert tr.step == 3
    got = tr.get_params()
    err = float(np.mean([np.abs(got[k] - P_ref[k]).mean() for k in P]))
    print(f"{mode}: mean abs parameter error after 3 AdamW steps {err:.2e}")
    assert err < tol  # Adam's first steps move every weight by ~lr regardless of gradient size: compare on average
    tr.close()


def test_error_paths(gpu_lib):
    from arp_amd import finetune as FT
    from arp_amd._ffi import ArpError
    fcfg = FT.FinetuneConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64)
    tr = FT.FinetuneTrainer(fcfg, mode="f32")
    with pytest.raises(ArpError, match="no batch staged"):
        tr.train_step(1e-3)
    b = list(FT.synth_batch(fcfg, 4, seed=1))
    b[5] = b[5].copy()
    b[5][0] = 99
    with pytest.raises(ArpError, match="action id out of range"):
        tr.set_batch(*b)
    with pytest.raises(ArpError, match="multiples of"):
        FT.FinetuneTrainer(FT.FinetuneConfig(layers=2, width_v=48, width_t=64, embed=64, hidden=64), mode="bf16")
    tr.close()


# ---- the frozen towers' side: per-block CLS / EOT features through the C ABI -----------------------------------------
TOWER = dict(patch=32, width=64, layers=2, heads=2, embed=64, img_res=224, txt_width=64, txt_layers=2, txt_heads=2, ctx=77, vocab=512)


@pytest.mark.parametrize("hw", [(256, 256), (224, 224), (64, 96)])
def test_multiscale_tower_features_match_oracle(gpu_lib, hw):
    import torch
    from arp_amd import clip, synth
    from oracle import clip_np as C, clip_torch as CT
    ocfg = C.ClipConfig(**TOWER)
    W = synth.clip_weights(ocfg, seed=41)
    fr = synth.procgen_like_frames(3, hw[0], hw[1], seed=42)
    tok = synth.prompt_tokens(3, [7, 3, 5], ctx=ocfg.ctx, vocab=ocfg.vocab, seed=43)
    Wt = CT.to_torch(W)
    ref_ii, ref_if = CT.encode_image_multiscale(Wt, ocfg, CT.finetune_transform(fr))
    ref_ti, ref_tf = CT.encode_text_multiscale(Wt, ocfg, tok)
    m = clip.ClipLabeller(clip.ClipConfig(**TOWER), W, mode="f32")
    ii, fi = m.encode_image_multiscale(fr)
    ti, tf = m.encode_text_multiscale(tok)
    for got, ref, name in ((ii, ref_ii, "image inter"), (fi, ref_if, "image final"), (ti, ref_ti, "text inter"), (tf, ref_tf, "text final")):
        err = np.abs(got - ref.numpy()).max() / max(np.abs(ref.numpy()).max(), 1e-6)
        assert err < 2e-4, (name, err)
    # the multi-scale text call must leave the cached prompt set of set_text alone
    m.set_text(tok[:1])
    before = m.text_features().copy()
    m.encode_text_multiscale(tok)
    assert np.array_equal(before, m.text_features())
    m.close()


def test_one_side_at_224_is_rejected(gpu_lib):
    from arp_amd import clip, synth
    from arp_amd._ffi import ArpError
    from oracle import clip_np as C
    m = clip.ClipLabeller(clip.ClipConfig(**TOWER), synth.clip_weights(C.ClipConfig(**TOWER), seed=1), mode="f32")
    with pytest.raises(ArpError, match="BOTH sides"):
        m.encode_image_multiscale(synth.procgen_like_frames(1, 224, 256, seed=2))
    m.close()


def test_frames_to_loss_end_to_end(gpu_lib):
    """uint8 frames + tokens -> towers (multi-scale export) -> head loss and gradients, against the oracle run on the oracle's
    own tower features: the composition CLIPMultiscaleAdapter.forward performs (clip_multiscale_adapter.py:177-250)."""
    import torch
    from arp_amd import clip, synth
    from arp_amd import finetune as FT
    from oracle import clip_np as C, clip_torch as CT, finetune_torch as O
    ocfg = C.ClipConfig(**TOWER)
    W = synth.clip_weights(ocfg, seed=51)
    B = 4
    frames = [synth.procgen_like_frames(B, 256, 256, seed=52 + k) for k in range(3)]
    tok = synth.prompt_tokens(B, [7, 3, 5, 4], ctx=ocfg.ctx, vocab=ocfg.vocab, seed=60)
    hcfg = O.HeadConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64)
    P = O.init_params(hcfg, seed=61)
    rng = np.random.Generator(np.random.PCG64(62))
    r, action = rng.integers(0, 2, B).astype(np.float32), rng.integers(0, 15, B).astype(np.int32)
    Wt = CT.to_torch(W)
    oi = [CT.encode_image_multiscale(Wt, ocfg, CT.finetune_transform(f)) for f in frames]
    ot = CT.encode_text_multiscale(Wt, ocfg, tok)
    obatch = (np.stack([x[0].numpy() for x in oi]), np.stack([x[1].numpy() for x in oi]), ot[0].numpy(), ot[1].numpy(), r, action)
    g_ref, aux = O.grads(P, hcfg, obatch)

    m = clip.ClipLabeller(clip.ClipConfig(**TOWER), W, mode="f32")
    gi = [m.encode_image_multiscale(f) for f in frames]
    gt = m.encode_text_multiscale(tok)
    tr = FT.FinetuneTrainer(FT.FinetuneConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64, logit_scale=hcfg.logit_scale), mode="f32")
    tr.set_params(P)
    tr.set_batch(np.stack([x[0] for x in gi]), np.stack([x[1] for x in gi]), gt[0], gt[1], r, action)
    out = tr.forward()
    assert abs(out["loss"] - aux["loss"]) < 2e-3, (out["loss"], aux["loss"])
    tr.backward()
    g = tr.get_grads()
    bad = {k: float(np.abs(g[k] - g_ref[k]).max() / max(np.abs(g_ref[k]).max(), 1e-6)) for k in g_ref}
    bad = {k: v for k, v in bad.items() if not v < 5e-3}
    assert not bad, bad
    tr.close()
    m.close()


@pytest.mark.parametrize("use_crop", [False, True])
def test_clip_ft_labelling_branch(gpu_lib, use_crop):
    """label_reward(model_type="clip_ft") (arp_dt/label_reward.py:165-230): towers + fine-tuned head -> reward, rtg datasets."""
    import torch
    from arp_amd import label_reward as LR, synth
    from arp_amd import finetune as FT
    from oracle import clip_np as C, clip_torch as CT, finetune_torch as O, rtg as R
    ocfg = C.ClipConfig(**TOWER)
    W = synth.clip_weights(ocfg, seed=71)
    hcfg = O.HeadConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64, logit_scale=float(W["logit_scale"]))
    P = O.init_params(hcfg, seed=72)
    P["image_residual_weight"] = np.float32(0.5) * np.ones((), np.float32)
    P["text_residual_weight"] = np.float32(-0.2) * np.ones((), np.float32)
    lens = [5, 3]
    frames = synth.procgen_like_frames(sum(lens), 256, 256, seed=73)
    tok = synth.prompt_tokens(1, 6, ctx=ocfg.ctx, vocab=ocfg.vocab, seed=74)
    # oracle reward per frame
    Wt = CT.to_torch(W)
    fr = frames[:, 64:192, 64:192] if use_crop else frames
    ii, fi = CT.encode_image_multiscale(Wt, ocfg, CT.finetune_transform(fr))
    ti, tf = CT.encode_text_multiscale(Wt, ocfg, tok)
    Pt = O.to_torch(P)
    a = O._encode(Pt, "image", ii.double(), fi.double())
    t = O._encode(Pt, "text", ti.double(), tf.double())
    ref = (np.exp(hcfg.logit_scale) * (a @ t[0])).numpy()

    nf = 4
    done = np.zeros((sum(lens), nf), bool)
    done[lens[0] - 1, -1] = done[-1, -1] = True
    store = {"ob": np.repeat(frames[:, None], nf, axis=1), "done": done}
    ckpt = {**{"clip_model." + k: v for k, v in W.items()}, **P}
    LR.label_reward("coinrun", "hard", 500, 0, "collect the coin", "/nonexistent", image_keys="ob", num_frames=nf, env_type="none",
                    model_type="clip_ft", use_crop=use_crop, store=store, weights=ckpt, tokens=tok, model_name=FT_TOWER_CFG(), mode="f32")
    got = store["ob_clip_ft_reward"][:, -1]
    assert np.abs(got - ref).max() < 2e-3 * max(1.0, np.abs(ref).max()), (got, ref)
    exp_rtg = np.concatenate([R.discount_cumsum(ref[:5]), R.discount_cumsum(ref[5:])])
    assert np.abs(store["ob_clip_ft_pos_rtg"][:, -1] - exp_rtg).max() < 1e-2


def FT_TOWER_CFG():
    from arp_amd import clip
    return clip.ClipConfig(**TOWER)


def test_device_resident_handoff_equals_host_path(gpu_lib):
    """arp_clip_encode_*_multiscale_dev -> arp_ft_set_batch_dev gives the same loss and gradients as the host-staged batch."""
    from arp_amd import clip, synth
    from arp_amd import finetune as FT
    from oracle import clip_np as C, finetune_torch as O
    ocfg = C.ClipConfig(**TOWER)
    W = synth.clip_weights(ocfg, seed=81)
    B = 3
    frames = np.concatenate([synth.procgen_like_frames(B, 256, 256, seed=82 + k) for k in range(3)])
    tok = synth.prompt_tokens(B, [7, 3, 5], ctx=ocfg.ctx, vocab=ocfg.vocab, seed=90)
    P = O.init_params(O.HeadConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64), seed=91)
    r, action = np.array([0, 1, 1], np.float32), np.array([3, 0, 14], np.int32)
    m = clip.ClipLabeller(clip.ClipConfig(**TOWER), W, mode="f32")
    tr = FT.FinetuneTrainer(FT.FinetuneConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64), mode="f32")
    tr.set_params(P)
    ii, fi = m.encode_image_multiscale(frames)
    ti, tf = m.encode_text_multiscale(tok)
    tr.set_batch(ii.reshape(3, B, -1), fi.reshape(3, B, -1), ti, tf, r, action)
    out_h = tr.forward()
    tr.backward()
    g_h = tr.get_grads()
    bufs = tr.feature_buffers(B)
    m.encode_multiscale_to(frames, tok, bufs)
    tr.set_batch_device(bufs, r, action)
    out_d = tr.forward()
    tr.backward()
    g_d = tr.get_grads()
    assert out_h["loss"] == out_d["loss"]
    assert all(np.array_equal(g_h[k], g_d[k]) for k in g_h)
    for b in bufs:
        b.free()
    tr.close()
    m.close()


def test_full_size_head_step_is_bitwise_reproducible(gpu_lib):
    """BASELINE configs[4] geometry (B = 64, 476 M trainable parameters, bf16): two runs of two AdamW steps from the same state
    are bit-identical (fixed-order split-K reductions, no float atomics) and the loss on the fixed batch goes down."""
    from arp_amd import finetune as FT
    cfg = FT.FinetuneConfig()
    P = FT.synth_params(cfg, seed=0)
    batch = FT.synth_batch(cfg, 64, seed=1)
    probe = ("image_intermediate_linear.weight", "text_adapter.layers.3.weight", "inverse_layer.layers.0.weight", "lambda_id", "image_residual_weight")
    runs = []
    for _ in range(2):
        tr = FT.FinetuneTrainer(cfg, mode="bf16")
        tr.set_params(P)
        tr.set_batch(*batch)
        aux = [tr.train_step(1e-4) for _ in range(2)]
        got = {}
        import ctypes as C
        from arp_amd import _ffi
        for k in probe:
            a = np.empty(tr.shapes[k], np.float32)
            _ffi.check(_ffi.lib.arp_ft_get_tensor(tr._h, k.encode(), 0, _ffi.as_ptr(a, C.c_float)))
            got[k] = a
        runs.append((aux, got))
        tr.close()
    (a0, p0), (a1, p1) = runs
    assert [a["loss"] for a in a0] == [a["loss"] for a in a1]
    assert all(np.array_equal(p0[k], p1[k]) for k in probe)
    assert a0[1]["loss"] < a0[0]["loss"]


def test_full_geometry_head_matches_oracle(gpu_lib):
    """The REAL head: 12 layers, widths 768 / 512, hidden 1024 -> 476 M parameters, B = 4, f32 mode,
    against the fp64 torch oracle: losses, every gradient tensor (9216- and 6656-wide contractions, the split-K weight-streaming
    GEMMs, the K-padded transposed operand copies of the weight-gradient GEMMs) and one AdamW step.  B = 4 keeps the oracle to
    seconds; the shapes that could hide an indexing bug do not depend on B."""
    from arp_amd import finetune as FT
    from oracle import finetune_torch as O
    fcfg = FT.FinetuneConfig(weight_decay=0.01)
    cfg = O.HeadConfig(logit_scale=fcfg.logit_scale)
    assert (cfg.layers, cfg.width_v, cfg.width_t, cfg.embed, cfg.hidden) == (12, 768, 512, 512, 1024)
    P = FT.synth_params(fcfg, seed=11)
    assert set(P) == set(O.param_shapes(cfg)) and sum(int(np.prod(v.shape)) for v in P.values()) > 470e6
    batch = FT.synth_batch(fcfg, 4, seed=12)
    g_ref, aux = O.grads(P, cfg, batch)
    tr = FT.FinetuneTrainer(fcfg, mode="f32")
    tr.set_params(P)
    tr.set_batch(*batch)
    out = tr.forward()
    for k in ("loss", "vip_loss", "id_loss"):
        assert abs(out[k] - aux[k]) < 1e-4 * max(1.0, abs(aux[k])), (k, out[k], aux[k])
    tr.backward()
    import ctypes as C
    from arp_amd import _ffi
    bad = []
    for k in P:  # tensor by tensor: the whole gradient dict would be another 1.9 GB of host memory
        a = np.empty(tr.shapes[k], np.float32)
        _ffi.check(_ffi.lib.arp_ft_get_tensor(tr._h, k.encode(), 1, _ffi.as_ptr(a, C.c_float)))
        r = g_ref[k]
        err = float(np.abs(a - r).max() / max(np.abs(r).max(), 1e-9))
        rel_l2 = float(np.linalg.norm(a.astype(np.float64) - r) / max(np.linalg.norm(r), 1e-30))
        print(f"  {k}: max err {err:.2e}, rel L2 {rel_l2:.2e}")
        # ReLU ties (a pre-activation within f32 noise of zero) move a few entries by one term; an indexing bug moves everything
        if rel_l2 > 1e-4 or err > 5e-3:
            bad.append((k, err, rel_l2))
    assert not bad, bad
    del g_ref
    lr = 1e-3
    P1, aux1 = O.train_steps(P, cfg, [batch], lr, 0.01, 1)
    st = tr.train_step(lr)
    assert abs(st["loss"] - aux1[0]["loss"]) < 1e-4 * max(1.0, abs(aux1[0]["loss"]))
    errs = []
    for k in P:
        a = np.empty(tr.shapes[k], np.float32)
        _ffi.check(_ffi.lib.arp_ft_get_tensor(tr._h, k.encode(), 0, _ffi.as_ptr(a, C.c_float)))
        errs.append(float(np.abs(a - P1[k]).mean()))
    print(f"mean abs parameter error after one AdamW step: {np.mean(errs):.2e}")
    assert np.mean(errs) < 2e-5  # first Adam step = lr * sign-like update: a ~0 gradient may flip sign, compare on average
    tr.close()


def test_adamw_leaves_gradientless_parameters_alone(gpu_lib):
    """use_id_loss off: inverse_layer.* and lambda_id get no gradient; torch.optim.AdamW (finetune.py:141) skips parameters whose
    .grad is None -- no decay, no moment update.  Three steps against the oracle (pinned on torch.optim.AdamW for this case)."""
    from arp_amd import finetune as FT
    from oracle import finetune_torch as O
    cfg = O.HeadConfig(**MID, use_id=False)
    P = O.init_params(cfg, seed=9)
    fcfg = FT.FinetuneConfig(**MID, weight_decay=0.05, use_id=False, logit_scale=cfg.logit_scale)
    batch = FT.synth_batch(fcfg, 5, seed=10)
    P_ref, _ = O.train_steps(P, cfg, [batch], 1e-3, 0.05, 3)
    tr = FT.FinetuneTrainer(fcfg, mode="f32")
    tr.set_params(P)
    for _ in range(3):
        tr.set_batch(*batch)
        tr.train_step(1e-3)
    got = tr.get_params()
    for k in P:
        if k.startswith("inverse_layer.") or k == "lambda_id":
            assert np.array_equal(got[k], P[k]), k  # bit for bit untouched
    assert float(np.mean([np.abs(got[k] - P_ref[k]).mean() for k in P])) < 2e-5
    tr.close()


def test_single_rank_comm_and_shards(gpu_lib):
    """RCCL communicator with world = 1 on the fine-tune head: all-reduce / broadcast are exercised and are identities (the
    DP = 8 of BASELINE configs[4] needs 8 GPUs; the two-rank host logic is tests/test_dist_gloo.py)."""
    from arp_amd import finetune as FT
    from oracle import finetune_torch as O
    cfg = O.HeadConfig(**MID)
    P = O.init_params(cfg, seed=13)
    fcfg = FT.FinetuneConfig(**MID, logit_scale=cfg.logit_scale)
    batch = FT.synth_batch(fcfg, 6, seed=14)
    a = FT.FinetuneTrainer(fcfg, mode="f32"); a.set_params(P); a.set_batch(*batch)
    b = FT.FinetuneTrainer(fcfg, mode="f32"); b.set_params(P)
    dp = FT.DataParallel(b, 0, 1, lambda obj, src=0: obj)
    x, y = a.train_step(1e-3), dp.train_step(batch, 1e-3)
    assert x == y
    pa, pb = a.get_params(), b.get_params()
    assert all(np.array_equal(pa[k], pb[k]) for k in pa)
    s0, s1 = FT.shard_batch(batch, 0, 2), FT.shard_batch(batch, 1, 2)
    assert s0[0].shape[1] == 3 and np.array_equal(np.concatenate([s0[0], s1[0]], 1), batch[0]) and np.array_equal(np.concatenate([s0[5], s1[5]]), batch[5])
    with pytest.raises(ValueError, match="does not divide"):
        FT.shard_batch(batch, 0, 4)
    a.close(); b.close()


def test_f16_head_training_tracks_f32_over_many_steps(gpu_lib):
    """The head's 16-bit mode (gradients seeded x1024 in the loss kernel, un-scaled in AdamW) against the f32 mode over 60 AdamW steps on
    two alternating batches: finite throughout, both learn, and the loss curves stay together."""
    from arp_amd import finetune as FT
    from oracle import finetune_torch as O
    cfg = O.HeadConfig(**MID)
    P = O.init_params(cfg, seed=11)
    fcfg = FT.FinetuneConfig(**MID, weight_decay=0.01, logit_scale=cfg.logit_scale)
    batches = [FT.synth_batch(fcfg, 6, seed=30 + i) for i in range(2)]
    curves = {}
    for mode in ("f32", "f16"):
        tr = FT.FinetuneTrainer(fcfg, mode=mode)
        tr.set_params(P)
        losses = []
        for i in range(60):
            tr.set_batch(*batches[i % 2])
            aux = tr.train_step(3e-4)
            assert np.isfinite(aux["loss"]), (mode, i, aux)
            losses.append(aux["loss"])
        curves[mode] = np.array(losses)
        tr.close()
    a, b = curves["f32"], curves["f16"]
    print("loss f32:", a[[0, 1, 19, 39, 59]], " f16:", b[[0, 1, 19, 39, 59]])
    assert a[-2:].mean() < a[:2].mean() and b[-2:].mean() < b[:2].mean(), "both modes must learn"
    rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-2)
    assert rel[:10].max() < 1e-2 and rel.max() < 0.1, (float(rel[:10].max()), float(rel.max()))


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_adamw_fused_into_the_weight_gradient_gemms(gpu_lib, monkeypatch, mode):
    """Single-process steps at the full geometry apply AdamW to the seven big weights from inside their dW GEMMs (gemm.h,
    GEMM_SITE_ADAMW: 26 bytes per parameter instead of 34).  Against ARP_FT_FUSE_ADAM=0 -- the same steps with the gradients stored and
    one AdamW pass -- parameters, moments and losses agree to f32 round-off of the gradient (two GEMM kernels, same operands), the
    getter refuses the gradients that were never stored, and arp_ft_backward still materialises all of them."""
    import ctypes as C
    from arp_amd import _ffi, finetune as FT
    cfg = FT.FinetuneConfig()
    P = FT.synth_params(cfg, seed=0)
    batch = FT.synth_batch(cfg, 64, seed=1)
    probe = ("image_intermediate_linear.weight", "text_adapter.layers.3.weight", "image_adapter.layers.0.weight", "inverse_layer.layers.0.weight",
             "inverse_layer.layers.3.bias", "image_residual_weight")
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("ARP_FT_FUSE_ADAM", fuse)
        tr = FT.FinetuneTrainer(cfg, mode=mode)
        tr.set_params(P)
        tr.set_batch(*batch)
        aux = [tr.train_step(1e-4) for _ in range(3)]
        got = {}
        for k in probe:
            for which in (0, 2, 3):  # parameter, first moment, second moment
                a = np.empty(tr.shapes[k], np.float32)
                _ffi.check(_ffi.lib.arp_ft_get_tensor(tr._h, k.encode(), which, _ffi.as_ptr(a, C.c_float)))
                got[(k, which)] = a
        g = np.empty(tr.shapes[probe[0]], np.float32)
        rc = _ffi.lib.arp_ft_get_tensor(tr._h, probe[0].encode(), 1, _ffi.as_ptr(g, C.c_float))
        assert (rc != 0) == (fuse == "1")  # fused: that gradient never existed in memory
        tr.backward()
        _ffi.check(_ffi.lib.arp_ft_get_tensor(tr._h, probe[0].encode(), 1, _ffi.as_ptr(g, C.c_float)))
        assert np.isfinite(g).all() and np.abs(g).max() > 0
        out[fuse] = ([a["loss"] for a in aux], got)
        tr.close()
    la, lb = out["1"][0], out["0"][0]
    assert max(abs(x - y) for x, y in zip(la, lb)) < 1e-5 * max(1.0, abs(lb[0])), (la, lb)
    for key, a in out["1"][1].items():
        b = out["0"][1][key]
        assert np.abs(a - b).max() <= 1e-5 * max(np.abs(b).max(), 1e-12) + 1e-12, key


@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_finetune_bucketed_allreduce_equals_serial(gpu_lib, monkeypatch, mode):
    """seven gradient buckets leave in production order from inside the backward, on a communication
    stream.  With the all-reduce path forced on at world = 1 the overlapped step equals the serial step and the step without a
    communicator, bit for bit."""
    from arp_amd import finetune as FT
    cfg = FT.FinetuneConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64, n_actions=5)
    P = FT.synth_params(cfg, seed=1)
    batch = FT.synth_batch(cfg, 6, seed=2)
    res = {}
    for name, env in (("plain", None), ("serial", {"ARP_FT_FORCE_COMM": "1", "ARP_FT_OVERLAP": "0"}), ("overlap", {"ARP_FT_FORCE_COMM": "1", "ARP_FT_OVERLAP": "1"})):
        for k in ("ARP_FT_FORCE_COMM", "ARP_FT_OVERLAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        tr = FT.FinetuneTrainer(cfg, mode=mode)
        tr.set_params(P)
        if env:
            tr.comm_init(FT.FinetuneTrainer.new_unique_id(), 1, 0)
            tr.broadcast_state()
        if name == "overlap":
            tr.profile(True)
        tr.set_batch(*batch)
        auxs = [tr.train_step(1e-3) for _ in range(3)]
        if name == "overlap":
            sites = tr.profile_read()
            assert all(f"ft.allreduce_b{b}" in sites for b in range(7)), sorted(sites)
        res[name] = (tr.get_params(), auxs, tr.get_grads())
        tr.close()
    for other in ("serial", "overlap"):
        for k in P:
            assert np.array_equal(res["plain"][0][k], res[other][0][k]), (other, k)
            assert np.array_equal(res["plain"][2][k], res[other][2][k]), (other, "grad", k)
        assert [a["loss"] for a in res["plain"][1]] == [a["loss"] for a in res[other][1]]


@pytest.mark.parametrize("mode,tol_l,tol_g", [("f32", 5e-5, 5e-4), ("f16", 2e-3, None)])
def test_goal_conditioned_head_matches_the_reference_class_fixture(gpu_lib, mode, tol_l, tol_g):
    """clip_multiscale_adapter.py:208-212,224-230 (goal_conditioned=True): four image groups, scores = -||a3 - a_k||, inverse-model
    input [a1|a3|a2|a3], no text head.  Fixture = the reference class itself with the switch on (make_golden_finetune.py --goal).
    Loss, both components, every gradient; AdamW leaves the text head's parameters and moments exactly alone."""
    import os
    from arp_amd import finetune as FT
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "finetune_tiny_goal.npz"))
    L, wv, wt, embed, hid, na = [int(v) for v in g["cfg"]]
    cfg = FT.FinetuneConfig(layers=L, width_v=wv, width_t=wt, embed=embed, hidden=hid, n_actions=na, gamma=float(g["gamma"]),
                            logit_scale=float(g["logit_scale"]), goal_conditioned=True)
    P = {k[6:]: g[k] for k in g.files if k.startswith("param:")}
    G = {k[5:]: g[k] for k in g.files if k.startswith("grad:")}
    tr = FT.FinetuneTrainer(cfg, mode=mode)
    tr.set_params(P)
    tr.set_batch(g["img_inter"], g["img_final"], None, None, g["r"], g["action"])
    out = tr.forward()
    assert abs(out["loss"] - float(g["loss"])) < tol_l and abs(out["vip_loss"] - float(g["vip_loss"])) < tol_l, (out["loss"], float(g["loss"]))
    assert abs(out["lambda_id"] * out["id_loss"] - float(g["lambda_id_times_id_loss"])) < 10 * tol_l
    tr.backward()
    got = tr.get_grads()
    if tol_g is not None:
        for k in G:
            if k.startswith("text_"):
                continue
            scale = max(np.abs(G[k]).max(), 1e-6)
            assert np.abs(got[k] - G[k]).max() / scale < tol_g, (k, np.abs(got[k] - G[k]).max() / scale)
    else:  # f16 operands: direction of the whole gradient
        num = sum(float((got[k].astype(np.float64) * G[k]).sum()) for k in G if not k.startswith("text_"))
        den = np.sqrt(sum(float((got[k].astype(np.float64) ** 2).sum()) for k in G if not k.startswith("text_")) * sum(float((G[k].astype(np.float64) ** 2).sum()) for k in G))
        assert num / den > 0.9999, num / den
    tr.set_batch(g["img_inter"], g["img_final"], None, None, g["r"], g["action"])
    tr.train_step(1e-3)
    after = tr.get_params()
    for k in P:
        if k.startswith("text_"):
            assert np.array_equal(after[k], P[k]), k  # no gradient: no decay, no moments (torch.optim.AdamW skips .grad is None)
        elif k != "lambda_id" or True:
            assert not np.array_equal(after[k], P[k]) or P[k].size == 0, k
    mu = tr.get_tensors(2)
    assert all(not np.any(mu[k]) for k in P if k.startswith("text_"))
    tr.close()


@pytest.mark.parametrize("full", [False, True])
def test_online_adapter_rewards(gpu_lib, full):
    """Row N4 with the fine-tuned model (envs/vl_reward.py:44-79): get_torch_clip_adapter_reward = exp(logit_scale) <adapted image, adapted prompt>
    (prompt 0, or the mean over a list of prompts) and get_torch_clip_adapter_goal_conditioned_reward = -||a(obs) - a(goal)|| -- on ONE frame per
    call, through the LABEL transform (Pillow bicubic; `preprocess(Image.fromarray(obs))`), not the fine-tune one.  Oracle: oracle/clip_torch towers
    on oracle/preprocess frames + oracle/finetune_torch head in fp64.  full = the real ViT-B/16 + 476 M-parameter head geometry in f16."""
    import torch
    from arp_amd import clip, label_reward as LR, synth
    from arp_amd import finetune as FT
    from oracle import clip_np as C, clip_torch as CT, finetune_torch as O, preprocess as PP
    if full:
        ccfg = clip.MODELS["ViT-B/16"]
        ocfg = C.ClipConfig(patch=16)
        hcfg = O.HeadConfig(logit_scale=float(np.log(100.0)))
        mode, tol = "f16", 2.5e-3
    else:
        ccfg, ocfg = clip.ClipConfig(**TOWER), C.ClipConfig(**TOWER)
        hcfg = O.HeadConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64, logit_scale=float(np.log(100.0)))
        mode, tol = "f32", 2e-4
    W = synth.clip_weights(ocfg, seed=171)
    W["logit_scale"] = np.float32(hcfg.logit_scale) * np.ones((), np.float32)
    P = O.init_params(hcfg, seed=172)
    P["image_residual_weight"] = np.float32(0.5) * np.ones((), np.float32)
    P["text_residual_weight"] = np.float32(-0.2) * np.ones((), np.float32)
    fr = synth.procgen_like_frames(3, 256, 256, seed=173)
    tok = synth.prompt_tokens(2, [6, 9], ctx=ocfg.ctx, vocab=ocfg.vocab, seed=174)
    Wt, Pt = CT.to_torch(W), O.to_torch(P)

    def adapted(frames):
        x = torch.from_numpy(PP.preprocess(frames)).float()
        ii, fi = CT.encode_image_multiscale(Wt, ocfg, x)
        return O._encode(Pt, "image", ii.double(), fi.double()).numpy()

    ti, tf = CT.encode_text_multiscale(Wt, ocfg, tok)
    t = O._encode(Pt, "text", ti.double(), tf.double()).numpy()
    a = adapted(fr)
    logit = np.exp(hcfg.logit_scale) * (t @ a.T)  # [2 prompts, 3 frames]
    ckpt = {**{"clip_model." + k: v for k, v in W.items()}, **P}
    m = FT.FinetunedClip.from_state_dict(ckpt, mode=mode, model=ccfg).set_text(tok)
    for i in range(3):
        r0 = LR.get_torch_clip_adapter_reward(m, fr[i], "one prompt")
        rm = LR.get_torch_clip_adapter_reward(m, fr[i], ["one prompt", "another"])
        assert r0.shape == (1,) and r0.dtype == np.float32 and rm.shape == (1,)
        assert abs(r0[0] - logit[0, i]) < tol * 100 and abs(rm[0] - logit[:, i].mean()) < tol * 100, (r0, logit[0, i], rm, logit[:, i].mean())
    g = LR.get_torch_clip_adapter_goal_conditioned_reward(m, fr[0], fr[2])
    assert isinstance(g, float) and abs(g + np.linalg.norm(a[0] - a[2])) < tol * 4, (g, np.linalg.norm(a[0] - a[2]))
    # use_crop: obs centre half, goal centre quarter (the reference's crop of the cropped obs)
    ac = np.concatenate([adapted(fr[0:1, 64:192, 64:192]), adapted(fr[2:3, 96:160, 96:160])])
    gc = LR.get_torch_clip_adapter_goal_conditioned_reward(m, fr[0], fr[2], use_crop=True)
    assert abs(gc + np.linalg.norm(ac[0] - ac[1])) < tol * 4
    rc = LR.get_torch_clip_adapter_reward(m, fr[1], "one prompt", use_crop=True)
    assert abs(rc[0] - np.exp(hcfg.logit_scale) * float(t[0] @ adapted(fr[1:2, 64:192, 64:192])[0])) < tol * 100
    m.close()


@pytest.mark.parametrize("mode", ["f16", "bf16"])
def test_nn_dx_path_equals_the_transposed_shadow_path(gpu_lib, monkeypatch, mode):
    """Round 4: dX = dY . W on the NN kernel (gemm_tn.h::gemm_nn_kernel: the weight read as stored through the transposing LDS read, every dX ahead of
    its layer's fused weight-gradient + AdamW GEMM) against ARP_FT_NN=0 (round 3: NT products on transposed weight shadows rebuilt every step), full
    476 M-parameter geometry, three steps: same 16-bit operands and the same products, summed in a different order -- losses, parameters and moments
    agree to f32 round-off; the NN handle never runs ft.refresh_shadows after its first step and holds no shadow buffers."""
    import ctypes as C
    from arp_amd import _ffi, finetune as FT
    cfg = FT.FinetuneConfig()
    P = FT.synth_params(cfg, seed=3)
    batch = FT.synth_batch(cfg, 64, seed=4)
    probe = ("image_intermediate_linear.weight", "text_intermediate_linear.weight", "image_adapter.layers.0.weight", "text_adapter.layers.3.weight",
             "inverse_layer.layers.0.weight", "image_adapter.layers.0.bias", "text_residual_weight")
    out = {}
    for nn in ("1", "0"):
        monkeypatch.setenv("ARP_FT_NN", nn)
        tr = FT.FinetuneTrainer(cfg, mode=mode)
        tr.set_params(P)
        tr.set_batch(*batch)
        aux = [tr.train_step(1e-4) for _ in range(2)]
        tr.profile(True); tr.profile_reset()
        aux.append(tr.train_step(1e-4))
        sites = tr.profile_read(); tr.profile(False)
        got = {}
        for k in probe:
            for which in (0, 2, 3):
                a = np.empty(tr.shapes[k], np.float32)
                _ffi.check(_ffi.lib.arp_ft_get_tensor(tr._h, k.encode(), which, _ffi.as_ptr(a, C.c_float)))
                got[(k, which)] = a
        out[nn] = ([a["loss"] for a in aux], got, sites)
        tr.close()
    assert "ft.refresh_shadows" not in out["1"][2] and "ft.refresh_shadows" in out["0"][2]
    assert {"ft.image_fc2_dX", "ft.image_fc1_dX", "ft.text_fc2_dX", "ft.text_fc1_dX", "ft.inverse_fc1_dX"} <= set(out["1"][2])
    la, lb = out["1"][0], out["0"][0]
    assert max(abs(x - y) for x, y in zip(la, lb)) < 2e-5 * max(1.0, abs(lb[0])), (la, lb)
    for key, a in out["1"][1].items():
        b = out["0"][1][key]
        # (a first-moment entry is ~ the gradient: the two summation orders differ in the last f32 bits of an entry that is itself a sum of 16-bit products)
        assert np.abs(a - b).max() <= 2e-4 * max(np.abs(b).max(), 1e-12) + 1e-12, (key, float(np.abs(a - b).max()), float(np.abs(b).max()))


@pytest.mark.parametrize("mode", ["f16", "bf16"])
def test_fused_adamw_leaves_the_gradientless_inverse_model_alone_at_full_geometry(gpu_lib, mode):
    """use_id = 0 (the reference's use_id_loss=False) in a 16-bit mode at the real geometry -- where ft.inverse_fc1_dW runs on the GEMM
    whose epilogue applies AdamW -- must leave inverse_layer.* and lambda_id bit for bit untouched (torch.optim.AdamW skips .grad is None: no decay, no
    moments), as the f32 path and ARP_FT_FUSE_ADAM=0 do.  Before the fix the fused epilogue decayed inverse_layer.layers.0.weight by 1 - lr wd per step."""
    import ctypes as C
    from arp_amd import _ffi, finetune as FT
    cfg = FT.FinetuneConfig(use_id=False, weight_decay=0.05)
    P = FT.synth_params(cfg, seed=5)
    tr = FT.FinetuneTrainer(cfg, mode=mode)
    tr.set_params(P)
    tr.set_batch(*FT.synth_batch(cfg, 64, seed=6))
    for _ in range(3):
        tr.train_step(1e-3)
    for k in ("inverse_layer.layers.0.weight", "inverse_layer.layers.0.bias", "inverse_layer.layers.3.weight", "inverse_layer.layers.3.bias", "lambda_id"):
        for which in (0, 2, 3):
            a = np.empty(tr.shapes[k], np.float32)
            _ffi.check(_ffi.lib.arp_ft_get_tensor(tr._h, k.encode(), which, _ffi.as_ptr(a, C.c_float)))
            want = np.asarray(P[k], np.float32).reshape(tr.shapes[k]) if which == 0 else np.zeros(tr.shapes[k], np.float32)
            assert np.array_equal(a, want), (k, which)
    moved = np.empty(tr.shapes["image_adapter.layers.0.weight"], np.float32)
    _ffi.check(_ffi.lib.arp_ft_get_tensor(tr._h, b"image_adapter.layers.0.weight", 0, _ffi.as_ptr(moved, C.c_float)))
    assert not np.array_equal(moved, P["image_adapter.layers.0.weight"])  # the rest of the head did train
    assert tr.dropped_gradients == 0
    tr.close()
