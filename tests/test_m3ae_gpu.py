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


def test_policy_logits_with_the_encoder_inside_full_geometry(gpu_lib, monkeypatch):
    """Row N1 at the real geometry (VERDICT r3 next #1a): frames in, the frozen encoder (ViT-B/16 at 256 x 256, 257 tokens) in front of the policy,
    against oracle/m3ae_np -> oracle/arpdt_torch in fp64, EIGHT seeds.
    * f32 mode -- the mode `bench.py`'s `policy_with_encoder` line times, and the reference's own arithmetic type (its JAX model runs in float32):
      asserted at north_star's 1e-3 on every seed (measured 1.4e-6 ... 2.1e-6).
    * f16x3 encoder (every GEMM operand an (hi, lo) binary16 pair, three 16-bit MFMAs per product, attention / LayerNorm in f32) in front of the f32
      policy: the 16-bit-MFMA mode VERDICT r3 asked for -- asserted at 1e-3 on every seed too (and at 1e-4: it is f32-accurate).
    * f16c encoder + f16 policy with adapter corrections (round 5): binary16 products whose operand roundings are corrected on the scaled
      fp4 MFMA: asserted at 1e-3 on every seed.  Round 6: the default plan corrects the WEIGHT roundings of in_proj / out_proj / fc1 (ARP_F16C_PLAN=1110,
      1.25x the binary16 K loop on three of the four products): measured 3.0e-4 ... 7.0e-4; round 5's plan (1221: + the activation roundings of out_proj / fc1 and
      fc2's weights, +0.7 ms per 32-sample step) runs beside it and is asserted at 5e-4 (measured 1.5e-4 ... 3.7e-4 -- it read 4.2e-4 ... 6.7e-4 in rounds 5 and 6
      until the x4 segment that fc1's epilogue writes for fc2 was repaired, gemm256.h: this arm is what pins that repair in the encoder).
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
        for mode in ("f32", "f16x3", "f16c", "f16c-1221") + (("f16",) if seed < 4 else ()):
            if mode == "f16c-1221":
                monkeypatch.setenv("ARP_F16C_PLAN", "1221")  # read by arp_enc_create
            enc = m3ae.M3AEEncoder(ecfg, EP, mode=mode.split("-")[0])
            monkeypatch.delenv("ARP_F16C_PLAN", raising=False)
            # f16x3 is an ENCODER mode: the policy behind it runs in f32.  f16c (round 5): the binary16 encoder AND the binary16 policy, each with the operand
            # roundings of its big products corrected on the fp4 MFMA -- the 16-bit configuration `bench.py`'s policy_with_encoder_f16c line times
            tr = PolicyTrainer(pcfg, mode={"f16x3": "f32", "f16c": "f16", "f16c-1221": "f16"}.get(mode, mode), adapter_corrections=mode.startswith("f16c"))
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
    assert len(fc) == 8 and max(fc) < 1e-3  # north_star on every seed at 16-bit speed (measured 3.0e-4 ... 7.0e-4; profiles/r6_n1_plan_sweep.txt)
    fc2 = [e for (m, s), e in errs.items() if m == "f16c-1221"]
    assert len(fc2) == 8 and max(fc2) < 5e-4  # (measured 1.5e-4 ... 3.7e-4)
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
    from arp_amd.train import create_val_step
    out = {}
    for name in ("sync", "host", "prefetch"):
        enc = m3ae.M3AEEncoder(ecfg, EP, mode="f32")
        state = TrainState.create(pcfg, P, mode="f32")
        state.trainer.attach_encoder(enc)
        fn = create_train_step(pcfg, lambda step: 1e-3, pcfg.weight_decay)
        losses = []
        if name == "sync":
            for b in batches:
                state.trainer.set_batch_images(b["image"]["ob"], b["action"], b["rtg"]["ob"])
                losses.append(state.trainer.train_step(1e-3)["loss"])
        elif name == "host":  # round 6: HOST batches of frames straight through the step functions (no prefetcher): the synchronous slot, set_batch_images
            vfn = create_val_step(pcfg)
            for i, b in enumerate(batches):
                state, aux, _ = fn(state, b, None)
                losses.append(aux["loss"])
                if i == 2:  # a validation step and a greedy action on frames in between leave the training state alone
                    vaux, _ = vfn(state, batches[0], None)
                    assert np.isfinite(vaux["loss"])
                    b0 = batches[0]
                    ga = state.trainer.greedy_action(b0["image"]["ob"], b0["action"], b0["rtg"]["ob"])
                    state.trainer.set_batch_images(b0["image"]["ob"], b0["action"], b0["rtg"]["ob"])
                    assert np.array_equal(ga, state.trainer.forward()["action_pred"][:, -1, :].argmax(-1)) and ga.shape == (4,)
        else:
            for b in prefetch_to_device(iter(batches), 2, state.trainer):
                state, aux, _ = fn(state, b, None)
                losses.append(aux["loss"])
        out[name] = (losses, state.trainer.get_params())
        state.trainer.close(); enc.close()
    assert out["sync"][0] == out["prefetch"][0] == out["host"][0]
    assert all(np.array_equal(out["sync"][1][k], out["prefetch"][1][k]) and np.array_equal(out["sync"][1][k], out["host"][1][k]) for k in out["sync"][1])


@pytest.mark.parametrize("mode", ["f32", "f16", "bf16"])
def test_part_streams_are_exact(gpu_lib, mode):
    """Round 6: a call's frames are encoded as contiguous parts on HIP streams of their own (arp_enc_set_streams).  A frame's encoding does not depend on the
    part it is in or on how many rows its part has: every setting gives the same bits."""
    from arp_amd import m3ae, synth_policy as S
    from oracle import m3ae_np as M
    cfg, ocfg = m3ae.EncoderConfig(**SMALL_ENC), M.EncConfig(**SMALL_ENC)
    P = S.m3ae_params(ocfg, seed=3)
    x = S.normalized_frames(11, cfg.img_res, seed=4)
    enc = m3ae.M3AEEncoder(cfg, P, mode=mode, max_frames=16)
    enc.set_streams(1)
    ref = enc.forward_representation(x)
    for n, first in ((2, 0), (2, 3), (3, 0), (4, 0)):
        enc.set_streams(n, first, 1)
        got = enc.forward_representation(x)
        assert np.array_equal(got, ref), (mode, n, first, float(np.abs(got - ref).max()))
    enc.set_streams(2, 0, 1)
    got = enc.forward_representation(x[:1])  # fewer frames than parts x min_part_frames: one part
    assert np.array_equal(got, ref[:1])
    enc.close()


def test_part_streams_are_exact_full_geometry(gpu_lib):
    """... and at the real geometry in the modes whose kernels exist there only (f16c: widths that are multiples of 256; f16x3's K-concatenated products)."""
    from arp_amd import m3ae, synth_policy as S
    from oracle import m3ae_np as M
    cfg, ocfg = m3ae.EncoderConfig(), M.EncConfig()
    P = S.m3ae_params(ocfg, seed=0)
    x = S.normalized_frames(5, 256, seed=1)
    for mode in ("f16c", "f16x3", "f16"):
        enc = m3ae.M3AEEncoder(cfg, P, mode=mode, max_frames=8)
        enc.set_streams(1)
        ref = enc.forward_representation(x)
        enc.set_streams(2, 2, 1)
        got = enc.forward_representation(x)
        assert np.array_equal(got, ref), (mode, float(np.abs(got - ref).max()))
        enc.close()


def test_train_steps_with_part_streams_inside(gpu_lib, monkeypatch):
    """The encoder's part streams in front of the policy step's captured chain: eager, eager, capture, replay, replay -- with the encoder enqueued eagerly ahead
    of the chain (the default), captured INTO it with its forks and joins (ARP_DT_ENC_EAGER=0), and on one stream: the same trajectory bit for bit."""
    from arp_amd import m3ae, synth_policy as S
    from arp_amd.train import PolicyConfig, PolicyTrainer
    from oracle import m3ae_np as M
    ecfg = m3ae.EncoderConfig(**TINY_ENC)
    pcfg = PolicyConfig(emb=64, depth=2, heads=4, window=3, enc_tokens=ecfg.tokens, enc_dim=ecfg.width, lambda_ret=0.5)
    EP = S.m3ae_params(M.EncConfig(**TINY_ENC), seed=5)
    P = S.policy_params(pcfg, seed=6)
    rng = np.random.default_rng(7)
    B = 4
    frames = S.normalized_frames(B * pcfg.window, ecfg.img_res, seed=8).reshape(B, pcfg.window, ecfg.img_res, ecfg.img_res, 3)
    act = rng.integers(0, pcfg.n_actions, (B, pcfg.window)).astype(np.int32)
    rtg = rng.random((B, pcfg.window, 1)).astype(np.float32)
    out = {}
    for name, streams, eager in (("one", 1, "1"), ("parts_eager", 3, "1"), ("parts_captured", 3, "0")):
        monkeypatch.setenv("ARP_DT_ENC_EAGER", eager)
        enc = m3ae.M3AEEncoder(ecfg, EP, mode="f32")
        enc.set_streams(streams, 0, 1)
        tr = PolicyTrainer(pcfg, mode="f32")
        tr.set_params(P)
        tr.attach_encoder(enc)
        losses = []
        for i in range(6):
            tr.set_batch_images(frames, act, rtg)
            losses.append(tr.train_step(1e-3)["loss"])
        fwd = tr.forward()["action_pred"]
        out[name] = (losses, tr.get_params(), fwd)
        tr.close(); enc.close()
    for name in ("parts_eager", "parts_captured"):
        assert out[name][0] == out["one"][0], (name, out[name][0], out["one"][0])
        assert all(np.array_equal(out[name][1][k], out["one"][1][k]) for k in out["one"][1]), name
        assert np.array_equal(out[name][2], out["one"][2]), name


SMALL_C = dict(patch=16, width=512, layers=2, heads=8, img_res=64)  # the smallest geometry the f16c products exist at (K % 256 == 0, K >= 512), 17 tokens


def _n1_trajectory(ecfg_kw, pcfg, B, steps, lr, seeds):
    """(f32 encoder + f32 policy) and (f16c encoder + f16 policy + adapter corrections) from the same frames and parameters: per step loss, gradient norms per tensor
    (of the step's raw loss gradient) and the parameters at the end."""
    from arp_amd import m3ae, synth_policy as S
    from arp_amd.train import PolicyTrainer
    from oracle import m3ae_np as M
    ecfg = m3ae.EncoderConfig(**ecfg_kw)
    EP = S.m3ae_params(M.EncConfig(**ecfg_kw), seed=seeds[0])
    P = S.policy_params(pcfg, seed=seeds[1])
    rng = np.random.default_rng(seeds[2])
    frames = S.normalized_frames(B * pcfg.window, ecfg.img_res, seed=seeds[3]).reshape(B, pcfg.window, ecfg.img_res, ecfg.img_res, 3)
    act = rng.integers(0, pcfg.n_actions, (B, pcfg.window)).astype(np.int32)
    rtg = rng.random((B, pcfg.window, 1)).astype(np.float32)
    res = {}
    for name, emode, pmode in (("f32", "f32", "f32"), ("f16c", "f16c", "f16")):
        enc = m3ae.M3AEEncoder(ecfg, EP, mode=emode)
        tr = PolicyTrainer(pcfg, mode=pmode, adapter_corrections=name == "f16c")
        tr.set_params(P)
        tr.attach_encoder(enc)
        losses, gnorms = [], []
        for i in range(steps):
            tr.set_batch_images(frames, act, rtg)
            aux = tr.train_step(lr)
            assert np.isfinite(aux["loss"]) and np.isfinite(aux["grad_norm"]), (name, i, aux)
            losses.append(aux["loss"])
            gnorms.append({k: float(np.linalg.norm(v.astype(np.float64))) for k, v in tr.get_grads().items()})
        res[name] = (np.array(losses), gnorms, tr.get_params())
        tr.close(); enc.close()
    return res


def _n1_trajectory_report(tag, res, lr, steps):
    a, b = res["f32"], res["f16c"]
    rel = np.abs(a[0] - b[0]) / np.maximum(np.abs(a[0]), 1e-6)
    worst, first = {}, {}
    for i in range(steps):
        for k, v in a[1][i].items():
            if v > 1e-12:
                worst[k] = max(worst.get(k, 0.0), abs(b[1][i][k] - v) / v)
                if i == 0:
                    first[k] = abs(b[1][i][k] - v) / v
    # a SCALAR parameter's gradient "norm" is one signed sum (residual_weight: sum of dY (A - x) over 25 M products of both signs) -- its relative error is not
    # the error of a norm over many entries and is reported apart
    scalars = {k for k, v in a[2].items() if v.size == 1}
    dp = {k: float(np.abs(a[2][k].astype(np.float64) - b[2][k]).mean()) for k in a[2]}
    moved = {k: float(np.abs(a[2][k].astype(np.float64)).mean()) for k in a[2]}
    wk = max(worst, key=worst.get)
    pk = max(dp, key=dp.get)
    print("   gradient-norm differences per tensor (FIRST step: same parameters on both sides, arithmetic only): " + ", ".join(f"{k.split('/')[-2] if '/' in k else k}/{k.split('/')[-1]} {v:.1e}" for k, v in sorted(first.items(), key=lambda kv: -kv[1])[:8]))
    print("   gradient-norm differences per tensor (worst step): " + ", ".join(f"{k.split('/')[-2] if '/' in k else k}/{k.split('/')[-1]} {v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:12]))
    print(f"N1 trajectory [{tag}] {steps} steps: loss f32 {a[0][[0, -1]]}, f16c {b[0][[0, -1]]}; max relative loss difference {rel.max():.2e} (first step {rel[0]:.2e}); "
          f"gradient norms: worst tensor {wk} {worst[wk]:.2e}; parameters: worst mean |dp| {pk} {dp[pk]:.2e} (lr x steps = {lr * steps:.1e})")
    return rel, {k: v for k, v in worst.items() if k not in scalars}, dp, {k: v for k, v in worst.items() if k in scalars}, first


def test_f16c_training_trajectory_tracks_f32_small(gpu_lib):
    """row N1's 16-bit configuration -- f16c encoder, f16 policy, adapter corrections -- against the f32 configuration over a TRAINING RUN with
    frames in, not one forward: 10 clipped Adam steps at the smallest geometry the f16c products exist at.  Loss within 1e-3 relative on every step (measured 1.7e-4;
    2e-7 on the first step).  Gradient norms per tensor: on the FIRST step -- the same parameters on both sides, so the difference is arithmetic -- within 0.5 %
    (measured 0.05 %), the scalar residual_weight included; over all ten steps within 10 %, the SCALAR residual_weight within 50 % (its gradient is one sum of 35 k
    products of opposite signs: once the parameters differ it measures how much of that sum cancels, not an error): Adam's first steps move every parameter by ~lr per step whatever its gradient's size, a flipped sign of a near-zero gradient
    entry costs 2 lr on that entry, and from there the two runs are different trajectories (measured at the worst step 5.8 % / 22 %; 512-wide contractions average
    the operand roundings less than the real 768 / 197 376-wide ones -- the full-geometry test below holds the 2 % the verdict asked for).  Parameters after the run
    within a tenth of the distance the run moved them (the bound is on the MEAN; measured 0.4 %)."""
    from arp_amd.train import PolicyConfig
    pcfg = PolicyConfig(emb=128, depth=2, heads=8, window=4, enc_tokens=17, enc_dim=512, lambda_ret=0.01)
    lr, steps = 3e-4, 10
    res = _n1_trajectory(SMALL_C, pcfg, 4, steps, lr, (11, 12, 13, 14))
    rel, worst, dp, worst_scalar, first = _n1_trajectory_report("small", res, lr, steps)
    assert rel.max() < 1e-3, float(rel.max())
    assert max(first.values()) < 5e-3, max(first.items(), key=lambda kv: kv[1])  # measured 5.0e-4 (AdapterMLP_0/Dense_0/bias); the scalar 2.2e-4
    assert max(worst.values()) < 0.1, max(worst.items(), key=lambda kv: kv[1])
    assert max(worst_scalar.values()) < 0.5, worst_scalar
    assert max(dp.values()) < 0.1 * lr * steps, max(dp.items(), key=lambda kv: kv[1])


def test_f16c_training_trajectory_tracks_f32_full_geometry(gpu_lib):
    """... and two steps at the real geometry (ViT-B/16 at 256 x 256 in front of the 26.9 M-parameter policy, B = 2): loss within 1e-3 relative (measured 2.6e-5; 2.2e-6 on the first step),
    every gradient tensor's norm within 0.5 % on the first step (measured 0.07 %) and 2 % on the second (measured 1.1 %; the scalar residual_weight within 6 %, measured
    2.7 %), parameters within a tenth of the distance moved (measured 0.7 %)."""
    from arp_amd.train import PolicyConfig
    pcfg = PolicyConfig(lambda_ret=0.01)
    lr, steps = 3e-4, 2
    res = _n1_trajectory(dict(), pcfg, 2, steps, lr, (50, 60, 70, 80))
    rel, worst, dp, worst_scalar, first = _n1_trajectory_report("full", res, lr, steps)
    assert rel.max() < 1e-3, float(rel.max())
    assert max(first.values()) < 5e-3, max(first.items(), key=lambda kv: kv[1])  # the first step, arithmetic only: measured 6.5e-4 (residual_weight), 2.2e-4 the worst tensor
    assert max(worst.values()) < 2e-2, max(worst.items(), key=lambda kv: kv[1])
    assert max(worst_scalar.values()) < 6e-2, worst_scalar
    assert max(dp.values()) < 0.1 * lr * steps, max(dp.items(), key=lambda kv: kv[1])


def test_encode_ahead_gives_the_same_trajectory(gpu_lib):
    """Round 6: the frozen encoder's pass for batch i + 1 enqueued on the encoder's stream while step i runs (arp_dt_encode_ahead) -- two device slots alternating,
    as prefetch_to_device feeds them -- against every step encoding its own batch at its head: same losses, same parameters, bit for bit.  Also: a slot that is
    selected again WITHOUT a new encode-ahead call is encoded by its step (the call's result is consumed once)."""
    from arp_amd import m3ae, synth_policy as S
    from arp_amd.train import PolicyConfig, PolicyTrainer
    from oracle import m3ae_np as M
    ecfg = m3ae.EncoderConfig(**TINY_ENC)
    pcfg = PolicyConfig(emb=64, depth=2, heads=4, window=3, enc_tokens=ecfg.tokens, enc_dim=ecfg.width, lambda_ret=0.5)
    EP = S.m3ae_params(M.EncConfig(**TINY_ENC), seed=5)
    P = S.policy_params(pcfg, seed=6)
    rng = np.random.default_rng(7)
    B = 4
    batches = []
    for k in range(2):
        frames = S.normalized_frames(B * pcfg.window, ecfg.img_res, seed=8 + k).reshape(B, pcfg.window, ecfg.img_res, ecfg.img_res, 3)
        batches.append((frames, rng.integers(0, pcfg.n_actions, (B, pcfg.window)).astype(np.int32), rng.random((B, pcfg.window, 1)).astype(np.float32)))
    out = {}
    for name in ("at_head", "ahead", "ahead_every_other"):
        enc = m3ae.M3AEEncoder(ecfg, EP, mode="f32")
        enc.set_streams(2, 0, 1)
        tr = PolicyTrainer(pcfg, mode="f32")
        tr.set_params(P)
        tr.attach_encoder(enc)
        for k in (0, 1):
            tr.upload_async(k, *batches[k], images=True)
        if name != "at_head":
            tr.encode_ahead(0)
        losses = []
        for i in range(9):
            tr.select(i & 1)
            tr.train_step_async(1e-3)
            if name == "ahead" or (name == "ahead_every_other" and i % 2 == 0):
                tr.encode_ahead((i + 1) & 1)
            if i % 3 == 2:
                tr.sync()
        tr.sync()
        aux = tr.train_step(1e-3)
        out[name] = (aux["loss"], tr.get_params())
        tr.close(); enc.close()
    for name in ("ahead", "ahead_every_other"):
        assert out[name][0] == out["at_head"][0], (name, out[name][0], out["at_head"][0])
        assert all(np.array_equal(out[name][1][k], out["at_head"][1][k]) for k in out["at_head"][1]), name


def test_default_f16_policy_behind_f32_encoder_outputs(gpu_lib):
    """the f16 policy behind REAL encoder outputs (the f32 encoder on the GPU: 5e-6 from the fp64 oracle) read 1.18e-3 on one seed of eight
    (profiles/r5_n1_probe.txt) -- encodings are not N(0,1).  The round-6 default (adapter corrections on, plan 22d) must hold north_star's 1e-3 on all eight
    with room: measured 2.1e-4 (max) / 1.0e-4 (median), asserted at 4e-4; the plain products (adapter_corrections=False) are printed beside it."""
    import torch
    from arp_amd import m3ae, synth_policy as S
    from arp_amd.train import PolicyConfig, PolicyTrainer
    from oracle import arpdt_torch as O
    ecfg = m3ae.EncoderConfig()
    pcfg, pocfg = PolicyConfig(lambda_ret=0.01), O.PolicyConfig(lambda_ret=0.01)
    errs = {"default": [], "plain": []}
    trs = {"default": PolicyTrainer(pcfg, mode="f16"), "plain": PolicyTrainer(pcfg, mode="f16", adapter_corrections=False)}
    for seed in range(8):
        EP = S.m3ae_params(ecfg, seed=50 + seed)
        P = S.policy_params(pcfg, seed=60 + seed)
        rng = np.random.default_rng(70 + seed)
        frames = S.normalized_frames(2 * pcfg.window, 256, seed=80 + seed)
        act = rng.integers(0, pcfg.n_actions, (2, pcfg.window)).astype(np.int32)
        rtg = rng.random((2, pcfg.window, 1)).astype(np.float32)
        e = m3ae.M3AEEncoder(ecfg, EP, mode="f32")
        enc = e.forward_representation(frames).reshape(2, pcfg.window, ecfg.tokens, ecfg.width)
        e.close()
        ref = O.forward({k: torch.from_numpy(v).double() for k, v in P.items()}, pocfg, torch.from_numpy(np.asarray(enc, np.float64)), torch.from_numpy(act).long(), torch.from_numpy(rtg).double())
        for name, tr in trs.items():
            tr.set_params(P)
            tr.set_batch(enc, act, rtg)
            out = tr.forward()
            errs[name].append(max(float(np.abs(out["action_pred"] - ref["action_pred"].numpy()).max()), float(np.abs(out["return_pred"] - ref["return_pred"].numpy()).max())))
    for tr in trs.values():
        tr.close()
    print("f16 policy behind f32 encoder outputs, 8 seeds: " + "; ".join(f"{k}: max {max(v):.2e} median {np.median(v):.2e}" for k, v in errs.items()))
    assert max(errs["default"]) < 4e-4, errs["default"]  # measured 2.1e-4 (4.8e-4 before the x4 segment of fc1's epilogue was repaired)
