"""CPU tests: the oracle against its pins (PIL, HF CLIPModel goldens, the torch port), and the
reference-generated plumbing goldens.  No GPU."""
import os

import numpy as np
import pytest

from conftest import TINY

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_preprocess_oracle_matches_pil_goldens():
    from oracle import preprocess as P
    g = np.load(os.path.join(G, "preprocess.npz"))
    assert (P.preprocess_u8(g["frames"]) == g["resized"]).all()
    assert (P.preprocess_u8(g["frames"], use_crop=True) == g["cropped"]).all()
    assert (P.preprocess_u8(g["small"]) == g["small_resized"]).all()


def test_preprocess_oracle_matches_live_pil():
    from PIL import Image
    from arp_amd import synth
    from oracle import preprocess as P
    fr = synth.noise_frames(1, 200, 200, seed=5)
    ref = np.asarray(Image.fromarray(fr[0]).resize((224, 224), Image.BICUBIC))
    assert (P.preprocess_u8(fr)[0] == ref).all()
    x = P.preprocess(fr)
    assert x.shape == (1, 3, 224, 224) and x.dtype == np.float32
    manual = ((ref.astype(np.float32) / np.float32(255) - P.CLIP_MEAN) / P.CLIP_STD).transpose(2, 0, 1)
    assert (x[0] == manual).all()


@pytest.mark.parametrize("name,tol", [("clip_tiny", 2e-5), ("clip_b32", 2e-5), ("clip_b16", 2e-5)])
def test_clip_oracle_matches_hf_golden(name, tol):
    from arp_amd import synth
    from oracle import clip_np as C, preprocess as P
    g = np.load(os.path.join(G, f"{name}.npz"), allow_pickle=True)
    cfg = C.ClipConfig(**{k: int(v) for k, v in g["cfg"]})
    W = C.cast_weights(synth.clip_weights(cfg, seed=int(g["seed"])), np.float64)
    x = P.preprocess(g["frames"]).astype(np.float64)
    f = C.encode_image(W, cfg, x)
    t = C.encode_text(W, cfg, g["tokens"])
    assert np.abs(C.l2n(f) - g["image_embeds"]).max() < 1e-6
    assert np.abs(C.l2n(t) - g["text_embeds"]).max() < 1e-6
    r = C.rewards_from_features(W, f, t)
    assert np.abs(r - g["rewards"]).max() < tol * 100
    # the reference's quirk Q1: only prompt 0 is ever used
    assert np.abs(C.compute_reward(synth.clip_weights(cfg, seed=int(g["seed"])), cfg, g["frames"], g["tokens"]) - g["rewards"]).max() < 1e-3


def test_torch_port_matches_numpy_oracle():
    from arp_amd import synth
    from oracle import clip_np as C, clip_torch as T
    cfg = C.ClipConfig(**TINY)
    W = synth.clip_weights(cfg, seed=1)
    fr = synth.procgen_like_frames(3, seed=2)
    tok = synth.prompt_tokens(1, 6, vocab=cfg.vocab, seed=3)
    for uc in (False, True):
        a = C.compute_reward(W, cfg, fr, tok, use_crop=uc)
        b = T.compute_reward(T.to_torch(W), cfg, fr, tok, use_crop=uc)
        assert np.abs(a - b).max() < 1e-4


def test_rtg_oracle_matches_reference_goldens():
    """oracle/rtg.py against datasets written by the reference's own label_reward()."""
    from oracle import rtg
    g = np.load(os.path.join(G, "rtg.npz"))
    for case in "abct":
        rewards, done = g[f"{case}_rewards"], g[f"{case}_done"]
        keys = list(g[f"{case}_keys"])
        assert keys == ["ob_clip_pos_rtg", "ob_clip_reward"]
        L, nf = done.shape
        store = {"done": done, "ob": np.zeros((L, nf, 1), np.float32)}
        if case == "t":  # the `time` fallback (label_reward.py:84-87): a 1-D `done` makes the done-key path raise
            store = {"done": done[:, -1].copy(), "time": g["t_time"], "ob": store["ob"]}
        store["ob"][:, -1, 0] = rewards
        out = rtg.label_file(store, lambda imgs: imgs[:, 0])
        for k in keys:
            assert out[k].shape == g[f"{case}__{k}"].shape
            assert (out[k] == g[f"{case}__{k}"]).all(), (case, k)
