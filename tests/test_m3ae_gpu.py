"""GPU parity of row N1 (frozen M3AE encoder) against oracle/m3ae_np.py, and of the policy step with the
encoder inside against the step fed with pre-computed encodings."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TINY_ENC = dict(patch=16, width=64, layers=2, heads=2, img_res=64)      # 17 tokens, head_dim 32 (VALU attention)
SMALL_ENC = dict(patch=16, width=128, layers=2, heads=2, img_res=64)    # head_dim 64 (MFMA attention in bf16)


@pytest.mark.parametrize("kw", [TINY_ENC, SMALL_ENC])
@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 6e-2)])
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
    for mode, tol in (("f32", 1e-4), ("bf16", 8e-2)):
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
