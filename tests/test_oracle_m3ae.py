"""CPU test: the M3AE-encoder oracle (oracle/m3ae_np.py) against HuggingFace ViTModel, an independent
pre-LN ViT implementation (tanh-GELU, eps 1e-6), on seeded weights."""
import numpy as np

from hf_vit import build_hf_vit, hf_forward


def test_m3ae_oracle_matches_hf_vit():
    from arp_amd import synth_policy as S
    from oracle import m3ae_np as M
    cfg = M.EncConfig(patch=16, width=64, layers=2, heads=2, img_res=64)
    assert S.m3ae_param_shapes(cfg) == M.param_shapes(cfg)
    P = S.m3ae_params(cfg, seed=1)
    x = S.normalized_frames(3, cfg.img_res, seed=2)
    ref = M.forward_representation(P, cfg, x)
    got = hf_forward(build_hf_vit(P, cfg, M.sincos_2d(cfg.width, cfg.tokens - 1)), x)
    assert ref.shape == (3, 17, 64) and np.abs(ref - got).max() < 1e-6


def test_sincos_matches_reference_formula():
    """get_2d_sincos_pos_embed (m3ae/model.py:118-136): first half of the channels encodes the w coordinate."""
    from oracle import m3ae_np as M
    pe = M.sincos_2d(16, 9)  # 3x3 grid
    assert pe.shape == (9, 16)
    om = 1.0 / 10000 ** (np.arange(4) / 4.0)
    i, j = 2, 1  # patch index i*3 + j
    exp = np.concatenate([np.sin(j * om), np.cos(j * om), np.sin(i * om), np.cos(i * om)])
    assert np.abs(pe[i * 3 + j] - exp).max() < 1e-12
