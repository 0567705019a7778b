"""GPU parity of path (2) against the torch oracle, through the C ABI: forward (logits / return /
losses), every gradient tensor, and multi-step training trajectories.  Tolerances: f32 mode tight; the 16-bit throughput
mode is f16 (IEEE half MFMA operands, what bench.py --path policy times): north_star's "policy logits within 1e-3" is asserted
for it; bf16 operands (8 significand bits) stay available with their measured ~1e-2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOGIT_TOL_16BIT = 1e-3  # north_star: policy logits within 1e-3 in the 16-bit mode

TINY = dict(emb=64, depth=2, heads=4, window=3, enc_tokens=5, enc_dim=64, lambda_ret=0.5)
SMALL = dict(emb=128, depth=2, heads=8, window=4, enc_tokens=9, enc_dim=128, lambda_ret=0.01)


def _setup(kw, B, seed):
    from arp_amd import synth_policy as S
    from arp_amd.train import PolicyConfig
    from oracle import arpdt_torch as O
    cfg, ocfg = PolicyConfig(**kw), O.PolicyConfig(**kw)
    P = S.policy_params(cfg, seed=seed)
    enc, act, rtg = S.policy_batch(cfg, B, seed=seed + 1)
    Pt = {k: torch.from_numpy(v).double() for k, v in P.items()}
    tb = (torch.from_numpy(enc).double(), torch.from_numpy(act).long(), torch.from_numpy(rtg).double())
    return cfg, ocfg, P, (enc, act, rtg), Pt, tb


FULL = dict(lambda_ret=0.01)  # PolicyConfig defaults = BASELINE configs[3]: emb 128, depth 2, 8 heads, window 4, 257 x 768 encodings


@pytest.mark.parametrize("kw,B", [(TINY, 4), (SMALL, 6), (TINY, 1), (FULL, 2)])
@pytest.mark.parametrize("mode,tol", [("f32", 2e-5), ("f16", LOGIT_TOL_16BIT), ("bf16", 3e-2)])
def test_forward_parity(gpu_lib, kw, B, mode, tol):
    """f16 (the default mode, timed by bench.py --path policy): north_star's 1e-3 on the logits at the REAL geometry (measured
    4.9e-4).  The toy geometries contract 64..128 terms where the real one contracts 768 and 197 376 -- less averaging of the
    operand rounding, measured 0.92-1.04e-3 -- and get 1.5e-3, as the labelling tests do for their toy geometry."""
    from arp_amd.train import PolicyTrainer
    from oracle import arpdt_torch as O
    if mode == "f16" and kw is not FULL:
        tol = 1.5e-3
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(kw, B, 3)
    ref = O.forward(Pt, ocfg, *tb)
    tr = PolicyTrainer(cfg, mode=mode)
    tr.set_params(P)
    tr.set_batch(enc, act, rtg)
    out = tr.forward()
    e1 = np.abs(out["action_pred"] - ref["action_pred"].numpy()).max()
    e2 = np.abs(out["return_pred"] - ref["return_pred"].numpy()).max()
    print(f"{mode} logits err {e1:.2e} return err {e2:.2e}")
    assert e1 < tol and e2 < tol, f"forward {mode}: logits err {e1}, return err {e2}"
    for k in ("loss", "acc", "trans_loss", "return_loss"):
        assert abs(out[k] - float(ref[k])) < max(tol, 1e-5), (k, out[k], float(ref[k]))
    # round trip of the parameter tree through the device layout
    got = tr.get_params()
    assert all((got[k] == P[k]).all() for k in P)
    tr.close()


@pytest.mark.parametrize("kw,B", [(TINY, 4), (SMALL, 6), (TINY, 3)])  # B = 3: 45 adapter rows, ragged for every 4-wide access
@pytest.mark.parametrize("mode,rtol", [("f32", 2e-4), ("f16", None), ("bf16", None)])
def test_gradient_parity(gpu_lib, kw, B, mode, rtol):
    from arp_amd.train import PolicyTrainer
    from oracle import arpdt_torch as O
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(kw, B, 5)
    g_ref, aux_ref, _ = O.grads(Pt, ocfg, *tb)
    tr = PolicyTrainer(cfg, mode=mode)
    tr.set_params(P)
    tr.set_batch(enc, act, rtg)
    tr.backward()
    g = tr.get_grads()
    bad = []
    if rtol is None:
        # bf16 throughput mode: the forward already differs by ~1e-2, so compare directions: every tensor's
        # gradient must be strongly aligned with the oracle's and the whole gradient nearly parallel
        num = den1 = den2 = 0.0
        for k in P:
            r = g_ref[k].numpy().ravel()
            a = g[k].ravel().astype(np.float64)
            num += a @ r; den1 += a @ a; den2 += r @ r
            if np.linalg.norm(r) > 1e-7 and (a @ r) / (np.linalg.norm(a) * np.linalg.norm(r) + 1e-30) < 0.9:
                bad.append((k, float((a @ r) / (np.linalg.norm(a) * np.linalg.norm(r)))))
        cos = num / np.sqrt(den1 * den2)
        cmin, nmax = (0.9999, 2e-3) if mode == "f16" else (0.995, 0.02)
        print(f"{mode} gradient: cosine {cos:.6f}, norm ratio {np.sqrt(den1 / den2):.5f}")
        assert cos > cmin and abs(np.sqrt(den1 / den2) - 1) < nmax, f"{mode} gradient: cosine {cos}, norm ratio {np.sqrt(den1 / den2)}"
    else:
        for k in P:
            r = g_ref[k].numpy()
            scale = max(np.abs(r).max(), 1e-6)
            err = np.abs(g[k] - r).max() / scale
            if not err < rtol:
                bad.append((k, float(err), float(scale)))
    assert not bad, f"gradient mismatch ({mode}): {bad}"
    tr.close()


@pytest.mark.parametrize("mode,tol", [("f32", 5e-5), ("f16", 2e-3), ("bf16", 2e-2)])
def test_train_steps_match_oracle(gpu_lib, mode, tol):
    """3 steps incl. global-norm clipping active (clip_norm small) and a warm-up schedule starting at lr 0."""
    from arp_amd.train import PolicyTrainer, PolicyConfig
    from oracle import arpdt_torch as O
    kw = dict(TINY, clip_norm=0.5)
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(kw, 4, 7)
    lr_fn = lambda t: 2e-3 * min(1.0, t / 2.0)
    tr = PolicyTrainer(cfg, mode=mode)
    tr.set_params(P)
    st = O.init_state(Pt)
    for s in range(3):
        tr.set_batch(enc, act, rtg)
        aux = tr.train_step(lr_fn(tr.step))
        st, oaux = O.train_step(st, ocfg, [tb], lr_fn)
        assert aux["train_state_step"] == s and abs(aux["learning_rate"] - lr_fn(s)) < 1e-9
        for k in ("loss", "trans_loss", "return_loss", "weight_penalty", "weight_l2", "acc"):
            assert abs(aux[k] - oaux[k]) < max(tol, 1e-4 * abs(oaux[k])), (s, k, aux[k], oaux[k])
        assert abs(aux["grad_norm"] - oaux["grad_norm"]) < ({"bf16": 5e-2, "f16": 5e-3}[mode] * oaux["grad_norm"] if mode != "f32" else 1e-4)
    got = tr.get_params()
    err = max(np.abs(got[k] - st["params"][k].numpy()).max() for k in P)
    # Adam's m/sqrt(v) update is sign-like on its first steps, so an element whose gradient is ~0 amplifies
    # f32-vs-f64 noise up to a fraction of lr (2e-3 here): measured 3.5e-5 in f32 mode
    # In bf16 mode a near-zero gradient can flip sign, which moves that weight by up to 2*lr per step (measured
    # max 5.8e-3 after 3 steps at lr <= 2e-3); the MEAN error stays small.
    assert err < (1e-4 if mode == "f32" else 1.2e-2), f"params after 3 steps: max err {err}"
    mean_err = float(np.mean([np.abs(got[k] - st["params"][k].numpy()).mean() for k in P]))
    assert mean_err < {"f32": 2e-6, "f16": 1e-4, "bf16": 6e-4}[mode], f"params after 3 steps: mean err {mean_err}"
    mu = tr.get_tensors(2)
    assert max(np.abs(mu[k] - st["mu"][k].numpy()).max() for k in P) < {"f32": 1e-5, "f16": 2e-3, "bf16": 2e-2}[mode]
    tr.close()


def test_create_train_step_surface(gpu_lib):
    """The reference call surface: create_train_step(model, lr_schedule, wd) -> fn(state, batch, rng)."""
    from arp_amd import train
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(TINY, 4, 9)
    fn = train.create_train_step(cfg, lambda step: 1e-3, cfg.weight_decay)
    state = train.TrainState.create(cfg, P, mode="f32")
    batch = {"image": {"ob": enc}, "action": act, "rtg": {"ob": rtg}, "instruct": None, "text_padding_mask": None}
    new_state, aux, rng = fn(state, batch, 123)
    assert rng == 123 and new_state.step == 1 and set(train.AUX_KEYS) <= set(aux)
    with pytest.raises(RuntimeError):
        fn(state, batch, 0)  # donated
    _, aux2, _ = fn(new_state, batch, 0)
    assert aux2["loss"] < aux["loss"]
    new_state.trainer.close()


def test_single_rank_comm(gpu_lib):
    """RCCL communicator with world = 1: the all-reduce / broadcast calls are exercised and are identities."""
    from arp_amd.train import PolicyTrainer
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(TINY, 4, 11)
    a = PolicyTrainer(cfg, mode="f32"); a.set_params(P); a.set_batch(enc, act, rtg)
    b = PolicyTrainer(cfg, mode="f32"); b.set_params(P); b.set_batch(enc, act, rtg)
    b.comm_init(PolicyTrainer.new_unique_id(), 1, 0)
    b.broadcast_state()
    x, y = a.train_step(1e-3), b.train_step(1e-3)
    assert x == y
    pa, pb = a.get_params(), b.get_params()
    assert all((pa[k] == pb[k]).all() for k in pa)
    a.close(); b.close()


def test_greedy_action_and_return(gpu_lib):
    """ARPDT.greedy_action / greedy_return (ARPDT.py:488-495), batch 1 as in the rollout loop."""
    from arp_amd.train import PolicyTrainer
    from oracle import arpdt_torch as O
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(TINY, 1, 13)
    ref = O.forward(Pt, ocfg, *tb)
    tr = PolicyTrainer(cfg, mode="f32")
    tr.set_params(P)
    assert (tr.greedy_action(enc, act, rtg) == ref["action_pred"][:, -1].argmax(-1).numpy()).all()
    r = ref["return_pred"].numpy()
    assert np.abs(tr.greedy_return(enc, act, rtg) - np.sign(r) * (np.exp(np.abs(r)) - 1)).max() < 1e-5
    tr.close()


@pytest.mark.parametrize("kw,B", [(TINY, 4), (SMALL, 6)])
def test_fused_policy_kernel_matches_unfused_path(gpu_lib, kw, B, monkeypatch):
    """policy_fused_kernel + the grouped gradient launches against the one-kernel-per-op path (ARP_DT_FUSED=0):
    same f32 arithmetic up to summation order."""
    from arp_amd.train import PolicyTrainer
    cfg, _, P, (enc, act, rtg), _, _ = _setup(kw, B, 11)
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("ARP_DT_FUSED", fused)
        tr = PolicyTrainer(cfg, mode="f32")
        tr.set_params(P)
        tr.set_batch(enc, act, rtg)
        out = tr.forward()
        tr.backward()
        res[fused] = (out, tr.get_grads())
        tr.close()
    (o1, g1), (o0, g0) = res["1"], res["0"]
    assert np.abs(o1["action_pred"] - o0["action_pred"]).max() < 1e-5
    assert np.abs(o1["return_pred"] - o0["return_pred"]).max() < 1e-5
    for k in ("loss", "acc", "trans_loss", "return_loss"):
        assert abs(o1[k] - o0[k]) < 1e-5, k
    bad = [(k, float(np.abs(g1[k] - g0[k]).max() / max(np.abs(g0[k]).max(), 1e-6))) for k in P]
    bad = [b for b in bad if not b[1] < 5e-5]
    assert not bad, bad


@pytest.mark.parametrize("kw,B", [(TINY, 4), (SMALL, 6)])
def test_fused_kernel_hi_lo_binary16_linears_are_f32_level(gpu_lib, kw, B, monkeypatch):
    """The 16-bit modes run the fused kernel's big linears on (hi, lo) binary16 operand pairs (policy_fused.h::pf_lin_x3: three 16x16x32 MFMAs per
    product, per-linear power-of-two scale).  Switched on inside the f32 mode (ARP_PF_X3=1) nothing else rounds to 16 bits, so the comparison with the
    per-op f32 path isolates them: same bars as the f32-MFMA fused kernel above -- 1e-5 on the outputs, 5e-5 relative on every gradient (backward operands
    of 1e-3 .. 1e-8 included: that is what the scale is for)."""
    from arp_amd.train import PolicyTrainer
    cfg, _, P, (enc, act, rtg), _, _ = _setup(kw, B, 11)
    res = {}
    for fused, x3 in (("1", "1"), ("0", "0")):
        monkeypatch.setenv("ARP_DT_FUSED", fused)
        monkeypatch.setenv("ARP_PF_X3", x3)
        tr = PolicyTrainer(cfg, mode="f32")
        tr.set_params(P)
        tr.set_batch(enc, act, rtg)
        out = tr.forward()
        tr.backward()
        res[fused] = (out, tr.get_grads())
        tr.close()
    (o1, g1), (o0, g0) = res["1"], res["0"]
    assert np.abs(o1["action_pred"] - o0["action_pred"]).max() < 1e-5
    assert np.abs(o1["return_pred"] - o0["return_pred"]).max() < 1e-5
    for k in ("loss", "acc", "trans_loss", "return_loss"):
        assert abs(o1[k] - o0[k]) < 1e-5, k
    bad = [(k, float(np.abs(g1[k] - g0[k]).max() / max(np.abs(g0[k]).max(), 1e-6))) for k in P]
    bad = [b for b in bad if not b[1] < 5e-5]
    assert not bad, bad


def test_fused_kernel_hi_lo_linears_keep_tiny_gradients(gpu_lib, monkeypatch):
    """Backward operands far below binary16's normal range: with both heads' last layers scaled by 1e-7 the gradients that reach the trunk are 1e-8 ..
    1e-12 (measured: 3e-9 .. 2e-8 on the biases and LayerNorm parameters).  Unscaled, their hi halves would be subnormal (or zero) and the gradients garbage; with the per-linear power-of-two scale they match the
    f32 per-op path to the same RELATIVE bar as at ordinary magnitudes."""
    from arp_amd.train import PolicyTrainer
    cfg, _, P, (enc, act, rtg), _, _ = _setup(TINY, 4, 5)
    P = {k: v.copy() for k, v in P.items()}
    P["action_outputs_0/layers_2/kernel"] *= 1e-7
    P["return_outputs_0/layers_2/kernel"] *= 1e-7
    res = {}
    for fused, x3 in (("1", "1"), ("0", "0")):
        monkeypatch.setenv("ARP_DT_FUSED", fused)
        monkeypatch.setenv("ARP_PF_X3", x3)
        tr = PolicyTrainer(cfg, mode="f32")
        tr.set_params(P)
        tr.set_batch(enc, act, rtg)
        tr.forward()
        tr.backward()
        res[fused] = tr.get_grads()
        tr.close()
    g1, g0 = res["1"], res["0"]
    # (the kernels' gradients carry the L2 term wd * W of main_procgen.py:114-117, ~2e-5 here whatever the loss does: biases and LayerNorm parameters
    #  are the ones that see only what came back through the linears)
    trunk = [k for k in P if k.startswith("policy/Block") and not k.endswith("kernel") and np.abs(g0[k]).max() > 0]
    sizes = {k: float(np.abs(g0[k]).max()) for k in trunk}
    assert trunk and max(sizes.values()) < 1e-6, ("the setup no longer makes the trunk gradients tiny", sizes)
    bad = [(k, float(np.abs(g1[k] - g0[k]).max() / np.abs(g0[k]).max())) for k in P if np.abs(g0[k]).max() > 0]
    bad = [b for b in bad if not b[1] < 5e-5]
    assert not bad, bad


@pytest.mark.parametrize("kw,B", [(TINY, 4), (SMALL, 6), (FULL, 2)])
def test_image_text_input_on_hi_lo_binary16_pairs_is_the_f32_product(gpu_lib, kw, B, monkeypatch):
    """The 16-bit modes keep image_text_input's forward contraction at f32 level (ARP_DT_ITI_F32).  Since round 4 it runs on (hi, lo) binary16 pairs split
    in flight from the f32 operands (dtops.h::iti_x3_kernel) instead of the f32 MFMA: same operands, products to 2^-22 -- the two must agree far inside
    what one binary16 rounding of either operand would cost (measured on the f16 ITI path: ~1e-4 on the logits)."""
    from arp_amd.train import PolicyTrainer
    cfg, _, P, (enc, act, rtg), _, _ = _setup(kw, B, 21)
    res = {}
    monkeypatch.setenv("ARP_DT_MIX_X16", "0")  # the same operands on both sides: the x3 kernel's mix from the f32 encodings, as the f32-MFMA path's mix launch forms it
    monkeypatch.setenv("ARP_DT_ADAPTER_PLAN", "22e")  # ... and from the f32 adapter output (the default e2m1 hand-off exists inside the x3 kernel's operand load only)
    for x3 in ("1", "0"):
        monkeypatch.setenv("ARP_DT_ITI_X3", x3)
        tr = PolicyTrainer(cfg, mode="f16")
        tr.set_params(P)
        tr.set_batch(enc, act, rtg)
        out = tr.forward()
        tr.backward()
        res[x3] = (out, tr.get_grads())
        tr.close()
    (o1, g1), (o0, g0) = res["1"], res["0"]
    e = max(np.abs(o1["action_pred"] - o0["action_pred"]).max(), np.abs(o1["return_pred"] - o0["return_pred"]).max())
    print(f"x3 vs f32-MFMA image_text_input: outputs differ by {e:.2e}")
    assert e < 5e-6
    bad = [(k, float(np.abs(g1[k] - g0[k]).max() / max(np.abs(g0[k]).max(), 1e-6))) for k in P]
    bad = [b for b in bad if not b[1] < 5e-4]  # (an image embedding that differs by 1e-7 flips a binary16 rounding of the backward activations here and there: measured up to 2.1e-4)
    assert not bad, bad


def test_long_window_uses_unfused_path(gpu_lib):
    """window 6 -> 18 tokens per sample: beyond the fused kernel's 16-row tile, served by the per-op kernels."""
    from arp_amd.train import PolicyTrainer
    from oracle import arpdt_torch as O
    kw = dict(TINY, window=6)
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(kw, 3, 13)
    g_ref, _, _ = O.grads(Pt, ocfg, *tb)
    tr = PolicyTrainer(cfg, mode="f32")
    tr.set_params(P)
    tr.set_batch(enc, act, rtg)
    tr.backward()
    g = tr.get_grads()
    bad = [(k, float(np.abs(g[k] - g_ref[k].numpy()).max() / max(np.abs(g_ref[k].numpy()).max(), 1e-6))) for k in P]
    bad = [b for b in bad if not b[1] < 2e-4]
    assert not bad, bad
    tr.close()


def test_full_size_step_is_bitwise_reproducible(gpu_lib):
    """BASELINE configs[3] geometry (B = 32 per GPU, window 4, 257x768 encodings, 26.9 M parameters): two runs of three train
    steps from the same state give bit-identical parameters and losses -- every reduction has a fixed order, no float atomics."""
    from arp_amd import synth_policy as S
    from arp_amd.train import PolicyConfig, PolicyTrainer
    cfg = PolicyConfig(lambda_ret=0.01)
    P = S.policy_params(cfg, seed=0)
    batch = S.policy_batch(cfg, 32, seed=1)
    runs = []
    for _ in range(2):
        tr = PolicyTrainer(cfg, mode="f16")
        tr.set_params(P)
        tr.set_batch(*batch)
        aux = [tr.train_step(5e-4) for _ in range(3)]
        runs.append((aux, tr.get_params()))
        tr.close()
    (a0, p0), (a1, p1) = runs
    assert [a["loss"] for a in a0] == [a["loss"] for a in a1]
    assert all(np.array_equal(p0[k], p1[k]) for k in p0)
    assert a0[-1]["loss"] < a0[0]["loss"]  # and it learns on a fixed batch


def test_full_geometry_matches_oracle(gpu_lib, monkeypatch):
    """The REAL shapes against the fp64 oracle: B = 2, window 4, enc_tokens 257, enc_dim 768 -- the 257-way
    split-K over K = 197 376 of image_text_input, the adapter GEMMs at M = 2056, the transposed K-padded operand copies of the
    weight-gradient GEMMs.  f32 mode: forward, every gradient tensor and one clipped Adam step; f16 mode (what bench.py --path
    policy times): logits within north_star's 1e-3, gradient direction, and the same Adam step to its 16-bit tolerance."""
    from arp_amd.train import PolicyTrainer
    from oracle import arpdt_torch as O
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(FULL, 2, 21)
    assert cfg.enc_tokens == 257 and cfg.enc_dim == 768 and cfg.window == 4
    ref = O.forward(Pt, ocfg, *tb)
    g_ref, _, _ = O.grads(Pt, ocfg, *tb)
    lr = 1e-3
    st, oaux = O.train_step(O.init_state(Pt), ocfg, [tb], lambda t: lr)
    # "f16+": the same with the ReLU backward + bias column sums forced into gemm256's epilogue (what B = 32 runs; at B = 2 the grid
    # is below the size where that kernel is chosen by itself)
    for mode, ltol, gtol in (("f32", 2e-5, 3e-4), ("f16", LOGIT_TOL_16BIT, None), ("f16+", LOGIT_TOL_16BIT, None)):
        if mode.endswith("+"):
            monkeypatch.setenv("ARP_DT_FUSE_RELU_BWD", "2")
            mode = mode[:-1]
        tr = PolicyTrainer(cfg, mode=mode)
        tr.set_params(P)
        tr.set_batch(enc, act, rtg)
        out = tr.forward()
        e1 = np.abs(out["action_pred"] - ref["action_pred"].numpy()).max()
        e2 = np.abs(out["return_pred"] - ref["return_pred"].numpy()).max()
        print(f"full geometry {mode}: logits err {e1:.2e} return err {e2:.2e}")
        assert e1 < ltol and e2 < ltol, f"full geometry {mode}: logits err {e1}, return err {e2}"
        tr.backward()
        g = tr.get_grads()
        if gtol is not None:
            # The adapter has 2 x 1.58 M ReLU inputs here; one or two of them sit within f32 noise of zero, and a flipped mask bit
            # moves one row of a weight gradient by a single term (~1e-3 of the tensor's max).  An indexing bug moves EVERYTHING:
            # so every tensor must agree in relative L2 norm, all but a sliver of its entries to gtol, and the tensors that do
            # not sit behind a ReLU to gtol everywhere.
            bad = []
            for k in P:
                r = g_ref[k].numpy()
                err = np.abs(g[k] - r)
                scale = max(np.abs(r).max(), 1e-9)
                rel_l2 = float(np.linalg.norm(g[k] - r) / max(np.linalg.norm(r), 1e-30))
                frac = float((err > gtol * scale).mean())
                behind_relu = k.startswith("AdapterMLP_0/")
                print(f"  {k}: max err {err.max() / scale:.2e}, rel L2 {rel_l2:.2e}, entries over tol {frac:.2e}")
                if rel_l2 > 1e-4 or frac > (2e-3 if behind_relu else 0.0):
                    bad.append((k, float(err.max() / scale), rel_l2, frac))
            assert not bad, f"full geometry {mode} gradients: {bad}"
        else:
            num = sum(float(g[k].ravel().astype(np.float64) @ g_ref[k].numpy().ravel()) for k in P)
            d1 = sum(float((g[k].astype(np.float64) ** 2).sum()) for k in P)
            d2 = sum(float((g_ref[k].numpy() ** 2).sum()) for k in P)
            cos = num / np.sqrt(d1 * d2)
            print(f"full geometry {mode}: gradient cosine {cos:.6f}, norm ratio {np.sqrt(d1 / d2):.5f}")
            assert cos > 0.9995 and abs(np.sqrt(d1 / d2) - 1) < 5e-3
            # ... and tensor by tensor (the 25 M-entry image_text_input kernel would hide a wrong bias gradient in the global cosine):
            # direction and size of every gradient tensor to 16-bit-operand accuracy
            bad = []
            for k in P:
                a, r = g[k].ravel().astype(np.float64), g_ref[k].numpy().ravel().astype(np.float64)
                nr = np.linalg.norm(r)
                if nr < 1e-12:
                    continue
                ck = float(a @ r / max(np.linalg.norm(a) * nr, 1e-300))
                rel = float(np.linalg.norm(a - r) / nr)
                print(f"  {k}: cosine {ck:.6f}, rel L2 {rel:.2e}")
                if ck < 0.999 or rel > 3e-2:
                    bad.append((k, ck, rel))
            assert not bad, f"full geometry {mode} per-tensor gradients: {bad}"
        aux = tr.train_step(lr)
        assert abs(aux["loss"] - oaux["loss"]) < (1e-4 if mode == "f32" else 2e-3), (aux["loss"], oaux["loss"])
        assert abs(aux["grad_norm"] - oaux["grad_norm"]) < (1e-4 if mode == "f32" else 5e-3) * max(1.0, oaux["grad_norm"])
        got = tr.get_params()
        # first Adam step = lr * sign-like update: compare the MEAN parameter error (a ~0 gradient may flip sign in any finite precision)
        mean_err = float(np.mean([np.abs(got[k] - st["params"][k].numpy()).mean() for k in P]))
        assert mean_err < (2e-5 if mode == "f32" else 2e-4), f"full geometry {mode}: mean param err after one step {mean_err}"
        tr.close()


def test_f16_training_tracks_f32_over_many_steps(gpu_lib):
    """The timed 16-bit mode against the f32 parity mode over a TRAINING RUN, not one step: 150 clipped Adam steps on a fixed batch at
    the real geometry (B = 8).  The f16 operand rounding perturbs every gradient (adapter tensors to ~2e-2 relative, see the
    full-geometry test), the 2^14 gradient scale must neither overflow nor flush gradients to zero as the loss falls -- so the two
    loss curves have to stay together all the way down, and both have to learn.  Both f16 forms run: the default (round 6: the adapter's
    forward products corrected on the fp4 MFMA) and the plain binary16 products.  Adam divides by the gradient's own magnitude, so a
    perturbation of either kind moves the early steps by O(lr) whatever its size: measured max over the first 20 steps 6.4e-3 (corrected) /
    below 5e-3 (plain), over all 150 steps 8.5e-3 -- the bound is 1e-2 / 5e-2 for both."""
    from arp_amd.train import PolicyTrainer
    cfg, _, P, (enc, act, rtg), _, _ = _setup(FULL, 8, 77)
    curves = {}
    for name, mode, corr in (("f32", "f32", None), ("f16", "f16", None), ("f16 plain", "f16", False)):
        tr = PolicyTrainer(cfg, mode=mode, adapter_corrections=corr)
        tr.set_params(P)
        tr.set_batch(enc, act, rtg)
        losses = []
        for step in range(150):
            aux = tr.train_step(3e-4)
            assert np.isfinite(aux["loss"]) and np.isfinite(aux["grad_norm"]), (name, step, aux)
            losses.append(aux["loss"])
        curves[name] = np.array(losses)
        tr.close()
    a = curves["f32"]
    for name in ("f16", "f16 plain"):
        b = curves[name]
        rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-3)
        print(f"loss f32: {a[[0, 9, 49, 99, 149]]}  {name}: {b[[0, 9, 49, 99, 149]]}  rel first 20 {rel[:20].max():.2e} all {rel.max():.2e}")
        assert a[-1] < 0.5 * a[0] and b[-1] < 0.5 * b[0], "both modes must learn the fixed batch"
        assert rel[:20].max() < 1e-2 and rel.max() < 5e-2, (name, float(rel[:20].max()), float(rel.max()))


def _fp4_grid_round(x):
    """OCP e2m1, round to nearest even, saturating at 6 (tests/test_ops_gpu.py::_quant_fp4)"""
    grid = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])
    x = np.asarray(x, np.float64)
    a = np.minimum(np.abs(x), 6.0)
    idx = np.clip(np.searchsorted(grid, a, side="left"), 1, 7)
    lo, hi = grid[idx - 1], grid[idx]
    mid = 0.5 * (lo + hi)
    return np.sign(x) * np.where((a > mid) | ((a == mid) & (idx % 2 == 0)), hi, lo)


def test_adapter_operand_rows_hold_what_the_products_assume(gpu_lib, monkeypatch):
    """The corrected adapter INSIDE the step, segment by segment (round 6).  Rounds 5 and 6 ran with every x4 segment an EPILOGUE wrote (fc1 -> fc2 here, c_fc -> c_proj
    in the f16c encoder) holding its first pair of codes four times over -- `__builtin_bit_cast(f16x2_v, v[i])` on an ext-vector element compiles to a read of element 0
    (gemm256.h) -- so the x4 . dW4 term of the product behind it added noise of the size it was meant to remove, and no test saw it: the unit test of the product builds
    its operand rows on the host, and the logits stayed inside 1e-3.  This reads the step's buffers back (arp_dt_debug_read) and restates each from the layer before:
    Xc / W1c / W2c / the hidden rows' x4 EXACTLY, the hidden rows' hi and dx4 up to f32 summation noise of the restated product, the output against the float64
    product of the segments as read back."""
    import ctypes as C
    from arp_amd.train import PolicyTrainer
    monkeypatch.setenv("ARP_DT_ADAPTER_PLAN", "22e")
    cfg, _, P, (enc, act, rtg), _, _ = _setup(FULL, 2, 100)
    D = cfg.enc_dim
    M = enc.size // D
    tr = PolicyTrainer(cfg, mode="f16", adapter_corrections=True)
    tr.set_params(P)
    tr.set_batch(enc, act, rtg)
    tr.forward()
    dec = np.concatenate([np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0]), -np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])])

    def read(name, nbytes):
        buf = np.empty(nbytes, np.uint8)
        assert gpu_lib.lib.arp_dt_debug_read(tr._h, name.encode(), buf.ctypes.data_as(C.c_void_p), nbytes) >= nbytes, name
        return buf

    def rows(buf, R):
        b = buf.reshape(R, 3 * D)

        def seg(x):
            out = np.empty((R, D), np.float64)
            out[:, 0::2] = dec[x & 15]
            out[:, 1::2] = dec[x >> 4]
            return out
        return b[:, : 2 * D].copy().view(np.float16).astype(np.float64), seg(b[:, 2 * D: 2 * D + D // 2]), seg(b[:, 2 * D + D // 2:])

    h16 = lambda a: a.astype(np.float16).astype(np.float64)  # noqa: E731
    sc = read("wc_scal", 64).view(np.int32)
    x = enc.reshape(M, D).astype(np.float64)
    Xc, H1c = rows(read("Xc", M * 3 * D), M), rows(read("H1c", M * 3 * D), M)
    A32 = read("A32", M * D * 4).view(np.float32).reshape(M, D).astype(np.float64)
    W1c, W2c = rows(read("W1c", D * 3 * D), D), rows(read("W2c", D * 3 * D), D)
    tr.close()
    assert np.array_equal(Xc[0], h16(x)) and np.array_equal(Xc[1], _fp4_grid_round(2.0 * h16(x))) and np.array_equal(Xc[2], _fp4_grid_round((x - h16(x)) * 2.0 ** 13))
    b1 = np.asarray(P["AdapterMLP_0/Dense_0/bias"]).astype(np.float64)
    b2 = np.asarray(P["AdapterMLP_0/Dense_1/bias"]).astype(np.float64)

    def wrows(name, sd, sw):
        Wt = np.asarray(P[name]).astype(np.float64).T
        return h16(Wt), _fp4_grid_round((Wt - h16(Wt)) * 2.0 ** sd), _fp4_grid_round(Wt * 2.0 ** sw)

    def product(Ac, Wr, sd, sw):
        return Ac[0] @ Wr[0].T + 2.0 ** -(1 + sd) * (Ac[1] @ Wr[1].T) + 2.0 ** -(13 + sw) * (Ac[2] @ Wr[2].T)

    W1r, W2r = wrows("AdapterMLP_0/Dense_0/kernel", sc[4], sc[5]), wrows("AdapterMLP_0/Dense_1/kernel", sc[12], sc[13])
    for nm, name, got, want, sd, sw in (("W1c", "AdapterMLP_0/Dense_0/kernel", W1c, W1r, sc[4], sc[5]), ("W2c", "AdapterMLP_0/Dense_1/kernel", W2c, W2r, sc[12], sc[13])):
        Wt = np.asarray(P[name]).astype(np.float64).T  # the product's rows are the kernel's output columns
        assert 6 < np.abs(Wt - h16(Wt)).max() * 2.0 ** sd <= 12 and 6 < np.abs(Wt).max() * 2.0 ** sw <= 12, (nm, sd, sw)  # one binade into saturation, as designed
        assert all(np.array_equal(g, w) for g, w in zip(got, want)), nm
    v1 = np.maximum(product(Xc, W1r, sc[4], sc[5]) + b1, 0.0)
    assert (np.abs(H1c[0] - h16(v1)) > 2.0 ** -10 * np.maximum(np.abs(v1), 2.0 ** -14)).mean() < 1e-3   # a binary16 tie moved by f32 summation noise: measured 3e-5
    assert np.array_equal(H1c[1], _fp4_grid_round(2.0 * H1c[0])), float((H1c[1] != _fp4_grid_round(2.0 * H1c[0])).mean())
    assert (H1c[2] != _fp4_grid_round((v1 - H1c[0]) * 2.0 ** 13)).mean() < 1e-2                              # measured 8e-4
    v2 = np.maximum(product(H1c, W2r, sc[12], sc[13]) + b2, 0.0)
    assert np.abs(A32 - v2).max() < 1e-5, float(np.abs(A32 - v2).max())                                     # measured 1.3e-6
    exact = np.maximum(np.maximum(x @ np.asarray(P["AdapterMLP_0/Dense_0/kernel"], np.float64) + b1, 0.0) @ np.asarray(P["AdapterMLP_0/Dense_1/kernel"], np.float64) + b2, 0.0)
    plain = np.maximum(h16(np.maximum(h16(x) @ W1r[0].T + b1, 0.0)) @ W2r[0].T + b2, 0.0)
    r_c, r_p = float(np.sqrt(((A32 - exact) ** 2).mean())), float(np.sqrt(((plain - exact) ** 2).mean()))
    print(f"adapter output, rms error against float64: corrected products {r_c:.2e}, plain binary16 products {r_p:.2e} ({r_c / r_p:.2f}x)")
    assert r_c < 0.35 * r_p, (r_c, r_p)


def test_f16_full_geometry_logits_over_sixteen_seeds(gpu_lib, monkeypatch):
    """VERDICT r2 next #2b: north_star's 1e-3 on the f16 logits AND return prediction at the REAL geometry (257 x 768 encodings,
    K = 197 376) across 16 seeds, not on two or three.  Prints max / p99 over all seeds (the committed log is the evidence).
    Measured (round 3): with every adapter-path operand in f16 (ARP_DT_ITI_F32=0, round 2's default) p99 is 7e-4 / 9.7e-4 and TWO seeds
    in sixteen reach 1.05e-3 -- the path rounds seven operands (X, W1, H1, W2, A, Y, Wi) at 2.4-6.1e-4 each.  The default now
    takes two of them out (image_text_input's forward contraction on the un-rounded mix and the f32 master weights, +7 % step
    time): max 8.7e-4, no seed outside.  Asserted: the default inside 1e-3 on every seed; the all-f16 switch inside 1.2e-3 (max)
    and 1e-3 (p99)."""
    from arp_amd.train import PolicyTrainer
    from oracle import arpdt_torch as O
    cases = []
    for seed in range(16):
        cfg, ocfg, P, batch, Pt, tb = _setup(FULL, 2, 100 + 7 * seed)
        ref = O.forward(Pt, ocfg, *tb)
        cases.append((cfg, P, batch, ref["action_pred"].numpy(), ref["return_pred"].numpy()))
    stats = {}
    # "corrected" (round 5): the adapter's forward products with their operand roundings corrected on the fp4 MFMA (arp_dt_set_adapter_corrections)
    for name, flag in (("all_f16", "0"), ("default", None), ("corrected", None)):
        monkeypatch.delenv("ARP_DT_ITI_F32", raising=False)
        if flag is not None:
            monkeypatch.setenv("ARP_DT_ITI_F32", flag)
        tr = PolicyTrainer(cases[0][0], mode="f16", adapter_corrections=name == "corrected")
        e_log, e_ret = [], []
        for cfg, P, (enc, act, rtg), r_log, r_ret in cases:
            tr.set_params(P)
            tr.set_batch(enc, act, rtg)
            out = tr.forward()
            e_log.append(np.abs(out["action_pred"] - r_log).ravel())
            e_ret.append(np.abs(out["return_pred"] - r_ret).ravel())
        tr.close()
        el, er = np.concatenate(e_log), np.concatenate(e_ret)
        per_seed = [float(max(a.max(), b.max())) for a, b in zip(e_log, e_ret)]
        stats[name] = (el.max(), np.quantile(el, 0.99), er.max(), np.quantile(er, 0.99), sum(v >= LOGIT_TOL_16BIT for v in per_seed))
        print(f"f16 FULL geometry, 16 seeds [{name}]: logits max {el.max():.2e} p99 {np.quantile(el, 0.99):.2e}; return max {er.max():.2e} p99 "
              f"{np.quantile(er, 0.99):.2e}; per-seed max {min(per_seed):.2e} .. {max(per_seed):.2e}; seeds outside 1e-3: {stats[name][4]} of 16")
    for name in stats:
        assert stats[name][1] < LOGIT_TOL_16BIT and stats[name][3] < LOGIT_TOL_16BIT, (name, stats[name])
    assert max(stats["all_f16"][0], stats["all_f16"][2]) < 1.2e-3, stats["all_f16"]
    assert max(stats["default"][0], stats["default"][2]) < LOGIT_TOL_16BIT, stats["default"]
    # the corrected adapter: logits max 1.3e-4 against 8.7e-4, return prediction 1.6e-4 against 7.4e-4 on these synthetic encodings (behind REAL encoder
    # outputs 2.1e-4 against 1.18e-3 over eight seeds: profiles/r6_adapter_plans.txt; rounds 5 and 6 read 3.9e-4 / 6.6e-4 here until the x4 segment fc1's epilogue
    # writes was repaired, gemm256.h) -- asserted: better on both heads, inside 1e-3
    assert stats["corrected"][0] < stats["default"][0] and stats["corrected"][2] < stats["default"][2], (stats["corrected"], stats["default"])
    assert max(stats["corrected"][0], stats["corrected"][2]) < LOGIT_TOL_16BIT
    # round 6: the corrected adapter is what PolicyTrainer(mode="f16") runs unless told otherwise (plan 22d: both operand roundings of both products, the output
    # handed to the mix as binary16 + the e2m1 code of its rounding error); VERDICT r5 next #2 asked for a 16-seed maximum of at most 7.5e-4: measured
    # 1.27e-4 logits / 1.63e-4 return, asserted at 3e-4
    assert max(stats["corrected"][0], stats["corrected"][2]) <= 3e-4, stats["corrected"]
    tr = PolicyTrainer(cases[0][0], mode="f16")  # the default
    cfg, P, (enc, act, rtg), r_log, r_ret = cases[0]
    tr.set_params(P)
    tr.set_batch(enc, act, rtg)
    out = tr.forward()
    tr.close()
    tr = PolicyTrainer(cases[0][0], mode="f16", adapter_corrections=True)
    tr.set_params(P)
    tr.set_batch(enc, act, rtg)
    out_c = tr.forward()
    tr.close()
    assert np.array_equal(out["action_pred"], out_c["action_pred"]) and np.array_equal(out["return_pred"], out_c["return_pred"]), "the default is not the corrected adapter"


def test_f16_step_survives_an_overflowing_backward(gpu_lib):
    """f16 mode scales the adapter path's backward activations by a fixed 2^14; a loss spike (huge rtg targets at
    B = 1) pushes them past binary16's range.  The scaled dz saturates instead of becoming inf, and a step whose gradient norm is
    still not finite is dropped -- either way parameters and moments stay finite and training can go on."""
    from arp_amd.train import PolicyTrainer
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(SMALL, 1, 11)
    tr = PolicyTrainer(cfg, mode="f16")
    tr.set_params(P)
    big = np.full_like(rtg, 3.0e4)
    big[0, 0, 0] = -3.0e4
    tr.set_batch(enc, act, big)
    aux = tr.train_step(1e-3)
    got = tr.get_params()
    assert all(np.isfinite(v).all() for v in got.values()), "a parameter went non-finite"
    assert all(np.isfinite(v).all() for v in tr.get_tensors(2).values()) and all(np.isfinite(v).all() for v in tr.get_tensors(3).values())
    print(f"overflow guard: loss {aux['loss']:.3e}, grad_norm {aux['grad_norm']:.3e}, "
          f"params moved: {any((got[k] != P[k]).any() for k in P)}")
    # ... and an ordinary batch afterwards trains normally
    tr.set_batch(enc, act, rtg)
    aux2 = tr.train_step(1e-3)
    assert np.isfinite(aux2["loss"]) and np.isfinite(aux2["grad_norm"]) and all(np.isfinite(v).all() for v in tr.get_params().values())
    tr.close()


@pytest.mark.parametrize("mode", ["f32", "f16"])
def test_bucketed_overlapped_allreduce_equals_serial(gpu_lib, monkeypatch, mode):
    """VERDICT r2 next #3.  The data-parallel step all-reduces the gradient in two buckets on a communication stream, bucket 1 while
    the adapter's backward still runs.  What one GPU can check: with the all-reduce path forced on at world = 1 (an identity
    reduction through RCCL, same streams / events / staged graphs) the overlapped step equals the serial step -- and the step
    without any communicator -- bit for bit, over several steps, at a geometry that takes the fused + TN path."""
    from arp_amd.train import PolicyTrainer
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(SMALL, 6, 21)
    res = {}
    for name, env in (("plain", None), ("serial", {"ARP_DT_FORCE_COMM": "1", "ARP_DT_OVERLAP": "0"}), ("overlap", {"ARP_DT_FORCE_COMM": "1", "ARP_DT_OVERLAP": "1"})):
        for k in ("ARP_DT_FORCE_COMM", "ARP_DT_OVERLAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        tr = PolicyTrainer(cfg, mode=mode)
        tr.set_params(P)
        if env:
            tr.comm_init(PolicyTrainer.new_unique_id(), 1, 0)
            tr.broadcast_state()
        if name == "overlap":
            tr.profile(True)
        auxs = []
        for i in range(5):  # eager, eager, capture, replay, replay of every staged graph
            tr.set_batch(enc, act, rtg)
            auxs.append(tr.train_step(1e-3))
        if name == "overlap":
            sites = tr.profile_read()
            assert "dt.allreduce_b1" in sites and "dt.allreduce_b2" in sites, sorted(sites)
        res[name] = (tr.get_params(), auxs, tr.get_grads())
        tr.close()
    for other in ("serial", "overlap"):
        for k in P:
            assert np.array_equal(res["plain"][0][k], res[other][0][k]), (other, k)
            assert np.array_equal(res["plain"][2][k], res[other][2][k]), (other, "grad", k)
        assert [a["loss"] for a in res["plain"][1]] == [a["loss"] for a in res[other][1]]


def test_validation_and_greedy_actions_between_prefetched_steps_leave_the_trajectory_alone(gpu_lib):
    """A validation step or a greedy action in between prefetched train steps stages its batch synchronously -- into a slot of its own
    (slot 2), never into one of the two slots the uploader thread may be filling at that moment (found by scripts/soak_policy.py:
    with two slots the training losses changed from run to run)."""
    from arp_amd import synth_policy as S
    from arp_amd.train import TrainState, create_train_step, create_val_step, prefetch_to_device
    cfg, ocfg, P, _, Pt, tb = _setup(SMALL, 4, 31)
    batches = []
    for i in range(5):
        enc, act, rtg = S.policy_batch(cfg, 4, seed=60 + i)
        batches.append({"image": {"ob": enc}, "action": act, "rtg": {"ob": rtg}})
    one = S.policy_batch(cfg, 1, seed=77)

    def run(interleave):
        state = TrainState.create(cfg, P, mode="f32")
        fn, vfn = create_train_step(cfg, lambda s: 1e-3, cfg.weight_decay), create_val_step(cfg)
        rng, losses, extras = np.array([0, 3], np.uint32), [], []
        for i, b in enumerate(prefetch_to_device((batches[j % 5] for j in range(25)), 2, state.trainer)):
            state, aux, rng = fn(state, b, rng)
            losses.append(aux["loss"])
            if interleave and i % 3 == 1:
                extras.append(vfn(state, batches[(i + 2) % 5], rng)[0]["loss"])
            if interleave and i % 4 == 2:
                extras.append(float(state.trainer.greedy_action(*one)[0]))
        p = state.params
        state.trainer.close()
        return losses, extras, p

    la, _, pa = run(False)
    lb, eb, pb = run(True)
    lc, ec, _ = run(True)
    assert la == lb == lc and eb == ec and len(eb) > 10
    assert all(np.array_equal(pa[k], pb[k]) for k in pa)


def test_prefetched_slots_equal_synchronous_upload(gpu_lib):
    """prefetch_to_device (main_procgen.py:703): batches uploaded into the two device slots by a background thread while the step on
    the other slot runs give the same trajectory as the synchronous set_batch path; val_step_fn leaves the state alone."""
    from arp_amd import synth_policy as S
    from arp_amd.train import PolicyTrainer, TrainState, create_train_step, create_val_step, prefetch_to_device
    cfg, ocfg, P, _, Pt, tb = _setup(SMALL, 4, 31)
    batches = []
    for i in range(6):
        enc, act, rtg = S.policy_batch(cfg, 4, seed=40 + i)
        batches.append({"image": {"ob": enc}, "action": act, "rtg": {"ob": rtg}})
    lr = lambda step: 1e-3
    key = np.array([0, 42], np.uint32)

    def run(prefetch):
        state = TrainState.create(cfg, P, mode="f32")
        fn, vfn = create_train_step(cfg, lr, cfg.weight_decay), create_val_step(cfg)
        rng, out = key, []
        src = prefetch_to_device(iter(batches), 2, state.trainer) if prefetch else iter(batches)
        for b in src:
            state, aux, rng = fn(state, b, rng)
            out.append(aux)
        vaux, vrng = vfn(state, batches[0], rng)
        p = state.params
        state.trainer.close()
        return out, p, vaux, rng, vrng

    a, pa, va, ra, vra = run(False)
    b, pb, vb, rb, vrb = run(True)
    assert [x["loss"] for x in a] == [x["loss"] for x in b] and [x["train_state_step"] for x in b] == list(range(6))
    assert all(np.array_equal(pa[k], pb[k]) for k in pa)
    assert va == vb and set(va) == {"loss", "trans_loss", "return_loss", "acc"}
    # the rng is jax.random.split's first output at every call (threefry, known-answer-tested on the CPU side)
    assert np.array_equal(ra, rb) and not np.array_equal(ra, key) and not np.array_equal(vra, ra)
    # val_step = the forward's metrics, rank mean at world 1
    tr = PolicyTrainer(cfg, mode="f32")
    tr.set_params(pa)
    tr.set_batch(batches[0]["image"]["ob"], batches[0]["action"], batches[0]["rtg"]["ob"])
    f = tr.forward()
    tr.close()
    assert abs(f["loss"] - va["loss"]) < 1e-6 and abs(f["acc"] * 100 - va["acc"]) < 1e-4 and abs(f["return_loss"] - va["return_loss"]) < 1e-6


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("kw", [dict(TINY, alibi_bias=True), dict(emb=128, depth=2, heads=8, window=4, enc_tokens=5, enc_dim=64, lambda_ret=0.5, alibi_bias=True),
                                dict(emb=64, depth=1, heads=2, window=7, enc_tokens=3, enc_dim=64, lambda_ret=1.0, alibi_bias=True)])
def test_alibi_bias_branch(gpu_lib, monkeypatch, kw, fused):
    """config.alibi_bias (arp_dt/ARPDT.py:88 -> layers.py:74-78, off in the shipped configuration): slope_h * key index added to the attention scores of the
    policy transformer, slopes = _get_attention_slopes (layers.py:97-110).  Forward and every gradient against the oracle with the same switch, on the fused
    kernel and on the per-op path (the third geometry -- 21 tokens, 2 heads of 32 -- runs the per-op path either way); and the switch really changes the logits."""
    from arp_amd.train import PolicyConfig, PolicyTrainer
    from oracle import arpdt_torch as O
    monkeypatch.setenv("ARP_DT_FUSED", fused)
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(kw, 3, 21)
    ref = O.forward(Pt, ocfg, *tb)
    g_ref, _, _ = O.grads(Pt, ocfg, *tb)
    tr = PolicyTrainer(cfg, mode="f32")
    tr.set_params(P)
    tr.set_batch(enc, act, rtg)
    out = tr.forward()
    assert np.abs(out["action_pred"] - ref["action_pred"].numpy()).max() < 2e-5 and np.abs(out["return_pred"] - ref["return_pred"].numpy()).max() < 2e-5
    tr.backward()
    g = tr.get_grads()
    bad = [(k, float(np.abs(g[k] - g_ref[k].numpy()).max() / max(np.abs(g_ref[k].numpy()).max(), 1e-6))) for k in P]
    bad = [b for b in bad if not b[1] < 1e-4]
    assert not bad, bad
    tr.close()
    off = PolicyTrainer(PolicyConfig(**dict(kw, alibi_bias=False)), mode="f32")
    off.set_params(P)
    off.set_batch(enc, act, rtg)
    assert np.abs(off.forward()["action_pred"] - out["action_pred"]).max() > 1e-4
    off.close()


def test_adapter_corrections_leave_the_backward_alone(gpu_lib):
    """arp_dt_set_adapter_corrections changes the FORWARD values of the adapter (operand roundings corrected on the fp4 MFMA, f32 hand-off to the mix);
    the backward reads the same plain binary16 Xb / H1 / A as before.  At the real geometry, B = 2: the corrected step's loss is the closer one to the fp64
    oracle's, its gradients agree with the plain f16 step's to 16-bit tolerance, and three steps of training track the plain mode."""
    from arp_amd.train import PolicyTrainer
    from oracle import arpdt_torch as O
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(FULL, 2, 31)
    ref = O.forward(Pt, ocfg, *tb)
    outs, grads, finals = {}, {}, {}
    for name in ("plain", "corrected"):
        tr = PolicyTrainer(cfg, mode="f16", adapter_corrections=name == "corrected")
        tr.set_params(P)
        tr.set_batch(enc, act, rtg)
        outs[name] = tr.forward()
        tr.backward()
        grads[name] = tr.get_grads()
        for _ in range(3):
            tr.set_batch(enc, act, rtg)
            tr.train_step(1e-3)
        finals[name] = tr.get_params()
        tr.close()
    err = {n: float(np.abs(outs[n]["action_pred"] - ref["action_pred"].numpy()).max()) for n in outs}
    print(f"adapter corrections, full geometry: logits err plain {err['plain']:.2e} corrected {err['corrected']:.2e}")
    assert err["corrected"] < err["plain"] and err["corrected"] < LOGIT_TOL_16BIT
    for k in P:
        a, b = grads["plain"][k].ravel().astype(np.float64), grads["corrected"][k].ravel().astype(np.float64)
        if np.linalg.norm(a) < 1e-12:
            continue
        assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))) > 0.999, k
    # three Adam steps at lr 1e-3 move a parameter by up to 3e-3, and a ~0 gradient may flip sign between the two modes: compare the MEAN difference
    assert float(np.mean([np.abs(finals["plain"][k] - finals["corrected"][k]).mean() for k in P])) < 3e-4


def test_round5_launch_forms_give_the_same_step(gpu_lib, monkeypatch):
    """Round 5's restructurings of the 16-bit step change WHERE work runs, not its arithmetic.  At the real geometry (B = 2), against the round-4 forms selected by
    their switches: the adapter's mix formed inside image_text_input's operand load (ARP_DT_ITI_MIX), the merged small launches (ARP_DT_MERGE: one gradient launch behind
    the fused kernel, dz's operand copy written by it, one launch for the adapter's three small reductions, one for the two norms), dWi produced last and Adam walking
    the state from its end (ARP_DT_DWI_LAST, ARP_DT_ADAM_REV) -- forward outputs, every gradient, the metrics and the parameters after three steps are BIT-identical.
    The two places that now read the encodings' binary16 copy instead of the f32 encodings are the exceptions, bounded here: adapter_dy_kernel (ARP_DT_DY_X16) touches one
    number, d loss / d residual_weight, by less than 1e-3 of itself; the mix inside image_text_input (ARP_DT_MIX_X16: the skip term weighs 1 - sigmoid(4) = 0.018) moves
    the logits by less than 2e-5 and every gradient by less than 1e-3 of its norm.
    (image_text_input's K-tiles dealt round-robin regroup its f32 partial sums, so that switch is held fixed here.)"""
    from arp_amd.train import PolicyTrainer
    cfg, ocfg, P, (enc, act, rtg), Pt, tb = _setup(FULL, 2, 37)
    # (round 6: the adapter output reaches the mix in f32 in every arm -- the default binary16 + e2m1 hand-off lives inside image_text_input's operand load, which
    #  the ARP_DT_ITI_MIX=0 arm does not have; what this test pins is where the work runs, not that hand-off)
    monkeypatch.setenv("ARP_DT_ADAPTER_PLAN", "22e")
    old = {"ARP_DT_ITI_MIX": "0", "ARP_DT_MERGE": "0", "ARP_DT_DWI_LAST": "0", "ARP_DT_ADAM_REV": "0", "ARP_DT_DY_X16": "0", "ARP_DT_MIX_X16": "0"}

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = PolicyTrainer(cfg, mode="f16")
        for k in env:
            monkeypatch.delenv(k)
        tr.set_params(P)
        tr.set_batch(enc, act, rtg)
        out = tr.forward()
        tr.backward()
        g = tr.get_grads()
        aux = []
        for _ in range(3):
            tr.set_batch(enc, act, rtg)
            aux.append(tr.train_step(1e-3))
        p = tr.get_params()
        tr.close()
        return out, g, aux, p

    o_old, g_old, a_old, p_old = run(old)
    o_x32, g_x32, a_x32, p_x32 = run({"ARP_DT_DY_X16": "0", "ARP_DT_MIX_X16": "0"})  # every round-5 form except the two binary16 reads of the encodings
    for k in o_old:
        assert np.array_equal(np.asarray(o_old[k]), np.asarray(o_x32[k])), k
    for k in P:
        assert np.array_equal(g_old[k], g_x32[k]), k
        assert np.array_equal(p_old[k], p_x32[k]), k
    assert [a["loss"] for a in a_old] == [a["loss"] for a in a_x32] and [a["grad_norm"] for a in a_old] == [a["grad_norm"] for a in a_x32]
    o_dy, g_dy, _, _ = run({"ARP_DT_MIX_X16": "0"})
    for k in P:
        if k == "residual_weight":
            d = float(np.abs(g_dy[k] - g_x32[k]).max() / max(float(np.abs(g_x32[k]).max()), 1e-30))
            print(f"d loss / d residual_weight, binary16 encodings in the dY kernel: relative change {d:.2e}")
            assert d < 1e-3
        else:
            assert np.array_equal(g_dy[k], g_x32[k]), k
    o_new, g_new, _, _ = run({})
    dl = float(np.abs(np.asarray(o_new["action_pred"]) - np.asarray(o_x32["action_pred"])).max())
    worst = max((float(np.linalg.norm((g_new[k] - g_x32[k]).ravel().astype(np.float64)) / max(float(np.linalg.norm(g_x32[k].ravel().astype(np.float64))), 1e-30)), k)
                for k in P if np.linalg.norm(g_x32[k]) > 1e-12)
    print(f"binary16 encodings in the mix: logits move by {dl:.2e}, worst gradient by {worst[0]:.2e} of its norm ({worst[1]})")
    assert dl < 2e-5 and worst[0] < 1e-3
