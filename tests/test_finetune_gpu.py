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


def test_bf16_mode_tracks_oracle(gpu_lib):
    from arp_amd import finetune as FT
    from oracle import finetune_torch as O
    cfg = O.HeadConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64)
    P = O.init_params(cfg, seed=5)
    batch = FT.synth_batch(FT.FinetuneConfig(layers=2, width_v=64, width_t=64, embed=64, hidden=64), 6, seed=6)
    g_ref, aux = O.grads(P, cfg, batch)
    tr = _trainer(cfg, "bf16")
    tr.set_params(P)
    tr.set_batch(*batch)
    out = tr.forward()
    assert abs(out["loss"] - aux["loss"]) < 5e-2 * max(1.0, abs(aux["loss"]))
    tr.backward()
    g = tr.get_grads()
    num = sum(float(g[k].ravel().astype(np.float64) @ g_ref[k].ravel()) for k in g_ref)
    den = np.sqrt(sum(float((g[k].astype(np.float64) ** 2).sum()) for k in g_ref) * sum(float((g_ref[k] ** 2).sum()) for k in g_ref))
    assert num / den > 0.99, num / den
    tr.close()


@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("bf16", 5e-3)])
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
