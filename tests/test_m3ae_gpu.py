"""GPU parity of row N1 (frozen M3AE encoder) against oracle/m3ae_np.py, and of the policy step with the
encoder inside against the step fed with pre-computed encodings."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TINY_ENC = dict(patch=16, width=64, layers=2, heads=2, img_res=64)      # 17 tokens, head_dim 32 (VALU attention)
SMALL_ENC = dict(patch=16, width=128, layers=2, heads=2, img_res=64)    # head_dim 64 (MFMA attention in bf16)


@pytest.mark.parametrize("kw", [TINY_ENC, SMALL_ENC])
@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("f16", 8e-3), ("bf16", 6e-2)])
def test_encoder_parity(gpu_lib, kw, mode, tol):
    from arp_amd import m3ae, synth_policy as S
    from oracle import m3ae_np as M
    cfg, ocfg = m3ae.EncoderConfig(**kw), M.EncConfig(**kw)
    P = S.m3ae_params(ocfg, seed=3)
    x = S.normalized_frames(5, cfg.img_res, seed=4)
    ref = M.forward_representation(P, ocfg, x)
    enc = m3ae.M3AEEncoder(cfg, P, mode=mode, max_frames=2)  # chunked: 2 + 2 + 1 frames
    got = enc.forward_representation(x)
    err = np.abs(got - ref).max()
    print(f"m3ae {kw['width']} {mode}: max err {err:.2e} (outputs are LayerNorm'ed, O(1))")
    assert got.shape == ref.shape and err < tol, f"encoder {mode}: max err {err}"
    enc.close()


def test_full_size_encoder_parity(gpu_lib):
    """ViT-B/16 geometry at 256x256 (257 tokens), seeded random-init weights."""
    from arp_amd import m3ae, synth_policy as S
    from oracle import m3ae_np as M
    cfg, ocfg = m3ae.EncoderConfig(), M.EncConfig()
    P = S.m3ae_params(ocfg, seed=0)
    x = S.normalized_frames(2, 256, seed=1)
    ref = M.forward_representation(P, ocfg, x)
    # f16x3: every GEMM operand an (hi, lo) pair of binary16 values, three MFMAs per product (round 4): f32-level error on the 16-bit MFMA
    # f16c (round 5): binary16 products with their operand roundings corrected on the fp4 MFMA: between f16 (3.1e-3 max, 5.5e-4 rms) and f16x3
    for mode, tol in (("f32", 1e-4), ("f16x3", 1e-4), ("f16c", 5e-3), ("f16", 1e-2), ("bf16", 8e-2)):
        enc = m3ae.M3AEEncoder(cfg, P, mode=mode)
        got = enc.forward_representation(x)
        err = np.abs(got - ref).max()
        print(f"m3ae ViT-B/16 {mode}: max err {err:.2e}, rms {np.sqrt(((got - ref) ** 2).mean()):.2e}")
        assert err < tol
        enc.close()


def test_train_step_with_encoder_inside(gpu_lib):
    """Frames in (the reference's own boundary) == encodings in, when the encodings are the encoder's output."""
    from arp_amd import m3ae, synth_policy as S
    from arp_amd.train import PolicyConfig, PolicyTrainer
    ecfg = m3ae.EncoderConfig(**TINY_ENC)
    pcfg = PolicyConfig(emb=64, depth=2, heads=4, window=3, enc_tokens=ecfg.tokens, enc_dim=ecfg.width, lambda_ret=0.5)
    from oracle import m3ae_np as M
    EP = S.m3ae_params(M.EncConfig(**TINY_ENC), seed=5)
    P = S.policy_params(pcfg, seed=6)
    rng = np.random.default_rng(7)
    B = 3
    frames = S.normalized_frames(B * pcfg.window, ecfg.img_res, seed=8).reshape(B, pcfg.window, ecfg.img_res, ecfg.img_res, 3)
    act = rng.integers(0, pcfg.n_actions, (B, pcfg.window)).astype(np.int32)
    rtg = rng.random((B, pcfg.window, 1)).astype(np.float32)
    enc = m3ae.M3AEEncoder(ecfg, EP, mode="f32")
    codes = enc.forward_representation(frames.reshape(-1, ecfg.img_res, ecfg.img_res, 3)).reshape(B, pcfg.window, ecfg.tokens, ecfg.width)
    a = PolicyTrainer(pcfg, mode="f32"); a.set_params(P); a.set_batch(codes, act, rtg)
    b = PolicyTrainer(pcfg, mode="f32"); b.set_params(P); b.attach_encoder(enc); b.set_batch_images(frames, act, rtg)
    fa, fb = a.forward(), b.forward()
    assert np.abs(fa["action_pred"] - fb["action_pred"]).max() < 1e-6
    xa, xb = a.train_step(1e-3), b.train_step(1e-3)
    assert abs(xa["loss"] - xb["loss"]) < 1e-6 and abs(xa["grad_norm"] - xb["grad_norm"]) < 1e-6
    pa, pb = a.get_params(), b.get_params()
    assert max(np.abs(pa[k] - pb[k]).max() for k in pa) < 1e-6
    a.close(); b.close(); enc.close()


def test_policy_logits_with_the_encoder_inside_full_geometry(gpu_lib):
    """Row N1 at the real geometry (VERDICT r3 next #1a): frames in, the frozen encoder (ViT-B/16 at 256 x 256, 257 tokens) in front of the policy,
    against oracle/m3ae_np -> oracle/arpdt_torch in fp64, EIGHT seeds.
    * f32 mode -- the mode `bench.py`'s `policy_with_encoder` line times, and the reference's own arithmetic type (its JAX model runs in float32):
      asserted at north_star's 1e-3 on every seed (measured 1.4e-6 ... 2.1e-6).
    * f16x3 encoder (every GEMM operand an (hi, lo) binary16 pair, three 16-bit MFMAs per product, attention / LayerNorm in f32) in front of the f32
      policy: the 16-bit-MFMA mode VERDICT r3 asked for -- asserted at 1e-3 on every seed too (and at 1e-4: it is f32-accurate).
    * f16c encoder + f16 policy with adapter corrections (round 5): binary16 products whose operand roundings are corrected on the scaled
      fp4 MFMA (1.5x the binary16 product instead of f16x3's 3x): asserted at 1e-3 on every seed (measured 4.2e-4 ... 6.7e-4).
    * f16 mode -- a THROUGHPUT mode with a stated error, NOT a parity claim: measured over these eight seeds (scripts/n1_parity_probe.py,
      profiles/r4_n1_probe.txt) 0.74e-3 ... 1.54e-3, four seeds outside 1e-3.  The probe also separates the two sources: f16 encoder in front of an
      f32 policy 0.50 ... 1.31e-3, f32 encoder in front of the f16 policy 0.57 ... 1.23e-3 -- each about one f16 rounding step (2^-11) per operand
      of a chain of dependent contractions whose outputs are themselves random sums, so there is no averaging to count on; halving it needs
      hi + lo operand pairs on both sides of every GEMM and an f32 attention, i.e. three MFMAs per product.  Asserted here at the
      measured level with headroom, on four seeds."""
    import torch
    from arp_amd import m3ae, synth_policy as S
    from arp_amd.train import PolicyConfig, PolicyTrainer
    from oracle import arpdt_torch as O, m3ae_np as M
    ecfg, eocfg = m3ae.EncoderConfig(), M.EncConfig()
    pcfg, pocfg = PolicyConfig(lambda_ret=0.01), O.PolicyConfig(lambda_ret=0.01)
    B, T = 2, pcfg.window
    errs = {}
    for seed in range(8):
        EP = S.m3ae_params(eocfg, seed=50 + seed)
        P = S.policy_params(pcfg, seed=60 + seed)
        rng = np.random.default_rng(70 + seed)
        frames = S.normalized_frames(B * T, 256, seed=80 + seed).reshape(B, T, 256, 256, 3)
        act = rng.integers(0, pcfg.n_actions, (B, T)).astype(np.int32)
        rtg = rng.random((B, T, 1)).astype(np.float32)
        codes = M.forward_representation(EP, eocfg, frames.reshape(-1, 256, 256, 3)).reshape(B, T, ecfg.tokens, ecfg.width)
        Pt = {k: torch.from_numpy(v).double() for k, v in P.items()}
        ref = O.forward(Pt, pocfg, torch.from_numpy(np.asarray(codes, np.float64)), torch.from_numpy(act).long(), torch.from_numpy(rtg).double())
        for mode in ("f32", "f16x3", "f16c") + (("f16",) if seed < 4 else ()):
            enc = m3ae.M3AEEncoder(ecfg, EP, mode=mode)
            # f16x3 is an ENCODER mode: the policy behind it runs in f32.  f16c (round 5): the binary16 encoder AND the binary16 policy, each with the operand
            # roundings of its big products corrected on the fp4 MFMA -- the 16-bit configuration `bench.py`'s policy_with_encoder_f16c line times
            tr = PolicyTrainer(pcfg, mode={"f16x3": "f32", "f16c": "f16"}.get(mode, mode), adapter_corrections=mode == "f16c")
            tr.set_params(P)
            tr.attach_encoder(enc)
            tr.set_batch_images(frames, act, rtg)
            out = tr.forward()
            e = max(float(np.abs(out["action_pred"] - ref["action_pred"].numpy()).max()), float(np.abs(out["return_pred"] - ref["return_pred"].numpy()).max()))
            errs[(mode, seed)] = e
            tr.close(); enc.close()
    print("policy logits / return with the encoder inside, full geometry: " + ", ".join(f"{m} seed {s}: {e:.2e}" for (m, s), e in errs.items()))
    f32 = [e for (m, s), e in errs.items() if m == "f32"]
    assert len(f32) == 8 and max(f32) < 1e-3  # north_star, on every seed, in the mode that is timed
    assert max(f32) < 5e-5
    x3 = [e for (m, s), e in errs.items() if m == "f16x3"]
    assert len(x3) == 8 and max(x3) < 1e-3 and max(x3) < 1e-4  # the 16-bit-MFMA mode that meets north_star (three MFMAs per product)
    fc = [e for (m, s), e in errs.items() if m == "f16c"]
    assert len(fc) == 8 and max(fc) < 1e-3  # north_star on every seed at 16-bit speed (measured 4.2e-4 ... 6.7e-4; profiles/r5_n1_probe.txt)
    assert max(e for (m, s), e in errs.items() if m == "f16") < 2.5e-3  # NOT north_star's 1e-3: see the docstring


def test_prefetched_frames_with_the_encoder_inside_equal_the_synchronous_path(gpu_lib):
    """prefetch_to_device with the frozen encoder attached: FRAMES go up into the device slots (arp_dt_upload_batch_images_async) while the
    step on the other slot runs; the trajectory equals set_batch_images' synchronous one."""
    from arp_amd import m3ae, synth_policy as S
    from arp_amd.train import PolicyConfig, TrainState, create_train_step, prefetch_to_device
    from oracle import m3ae_np as M
    ecfg = m3ae.EncoderConfig(**TINY_ENC)
    pcfg = PolicyConfig(emb=64, depth=2, heads=4, window=3, enc_tokens=ecfg.tokens, enc_dim=ecfg.width, lambda_ret=0.5)
    EP = S.m3ae_params(M.EncConfig(**TINY_ENC), seed=5)
    P = S.policy_params(pcfg, seed=6)
    rng = np.random.default_rng(9)
    batches = []
    for i in range(5):
        frames = S.normalized_frames(4 * pcfg.window, ecfg.img_res, seed=20 + i).reshape(4, pcfg.window, ecfg.img_res, ecfg.img_res, 3)
        batches.append({"image": {"ob": frames}, "action": rng.integers(0, pcfg.n_actions, (4, pcfg.window)).astype(np.int32),
                        "rtg": {"ob": rng.random((4, pcfg.window, 1)).astype(np.float32)}})
    out = {}
    for name in ("sync", "prefetch"):
        enc = m3ae.M3AEEncoder(ecfg, EP, mode="f32")
        state = TrainState.create(pcfg, P, mode="f32")
        state.trainer.attach_encoder(enc)
        fn = create_train_step(pcfg, lambda step: 1e-3, pcfg.weight_decay)
        losses = []
        if name == "sync":
            for b in batches:
                state.trainer.set_batch_images(b["image"]["ob"], b["action"], b["rtg"]["ob"])
                losses.append(state.trainer.train_step(1e-3)["loss"])
        else:
            for b in prefetch_to_device(iter(batches), 2, state.trainer):
                state, aux, _ = fn(state, b, None)
                losses.append(aux["loss"])
        out[name] = (losses, state.trainer.get_params())
        state.trainer.close(); enc.close()
    assert out["sync"][0] == out["prefetch"][0]
    assert all(np.array_equal(out["sync"][1][k], out["prefetch"][1][k]) for k in out["sync"][1])
