"""GPU parity tests, one kernel at a time, through the C ABI (arp_op_* / arp_preprocess)."""
import ctypes as C

import numpy as np
import pytest

from conftest import bf16_round

pytestmark = pytest.mark.gpu


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _act(x, act):
    if act == 0:
        return x
    if act == 1:
        return x / (1.0 + np.exp(-1.702 * x))
    if act == 2:
        return np.maximum(x, 0)
    if act == 3:
        return np.tanh(x)
    if act == 4:
        return 0.5 * x * (1 + np.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))
    raise ValueError


GEMM_SHAPES = [  # M, N, K
    (200, 768, 768), (128, 128, 64), (1, 64, 64), (77, 1536, 512), (333, 192, 3072), (257, 15, 128), (50, 1, 128),
    (1024, 2304, 768), (130, 516, 96),
]


@pytest.mark.parametrize("kernel", ["128", "256", "paired"])
@pytest.mark.parametrize("mode", [0, 1, 2])  # f32, bf16, f16 operands
@pytest.mark.parametrize("shape", GEMM_SHAPES)
def test_gemm_nt(gpu_lib, mode, shape, kernel, monkeypatch):
    """The three GEMM kernels (ARP_GEMM=1: 128x128 double buffer; ARP_GEMM=2: 256x256 two-phase pipelined; ARP_GEMM=3: 128x192,
    two workgroups per CU -- 16-bit operands and its instantiated epilogues, everything else falls through to the auto choice)."""
    monkeypatch.setenv("ARP_GEMM", {"128": "1", "256": "2", "paired": "3"}[kernel])
    M, N, K = shape
    if mode != 0 and K % 64:
        pytest.skip("16-bit GEMM needs K % 64 == 0")
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32)
    for act, use_b, use_r in ((0, True, True), (1, True, False), (0, False, False), (4, True, True), (2, True, False), (3, False, True)):
        out = np.empty((M, N), np.float32)
        gpu_lib.check(gpu_lib.lib.arp_op_gemm_nt(mode, act, _fp(A), _fp(W), _fp(bias) if use_b else None,
                                                 _fp(resid) if use_r else None, _fp(out), M, N, K))
        rnd = {0: lambda x: x, 1: bf16_round, 2: lambda x: x.astype(np.float16).astype(np.float32)}[mode]
        a64, w64 = rnd(A).astype(np.float64), rnd(W).astype(np.float64)
        ref = a64 @ w64.T
        if use_b:
            ref = ref + bias
        ref = _act(ref, act)
        if use_r:
            ref = ref + resid
        err = np.abs(out - ref).max()
        tol = 2e-5 * np.sqrt(K / 64) if mode == 0 else 3e-4 * max(1.0, np.abs(ref).max())
        assert err < tol, f"gemm mode={mode} shape={shape} act={act} bias={use_b} resid={use_r}: max err {err} (tol {tol})"


SKINNY_SHAPES = [  # M, N, K: the single-frame tower's products (50 / 197 token rows, class rows) and the edges of the kernel's domain
    (50, 2304, 768), (197, 768, 768), (50, 768, 3072), (197, 3072, 768), (1, 512, 768), (8, 768, 3072), (256, 64, 64), (17, 16, 32),
    (49, 768, 3072), (100, 192, 160), (208, 128, 256), (600, 2304, 768), (1000, 768, 768), (394, 768, 3072),
]


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("shape", SKINNY_SHAPES)
def test_skinny_gemm(gpu_lib, mode, shape):
    """csrc/skinny.hip: direct epilogues, and split-K slabs + the reduce / residual / LayerNorm row kernel,
    against float64 on the rounded operands."""
    M, N, K = shape
    rng = np.random.default_rng(M * 5 + N * 3 + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32)
    rnd = {1: bf16_round, 2: lambda x: x.astype(np.float16).astype(np.float32)}[mode]
    prod = rnd(A).astype(np.float64) @ rnd(W).astype(np.float64).T
    for act, use_b, use_r in ((0, True, True), (1, True, False), (0, False, False), (4, True, True)):
        out = np.empty((M, N), np.float32)
        gpu_lib.check(gpu_lib.lib.arp_op_skinny_gemm(mode, act, _fp(A), _fp(W), _fp(bias) if use_b else None, _fp(resid) if use_r else None, _fp(out),
                                                     M, N, K, 0, None, None, 0.0, None))
        ref = _act(prod + (bias if use_b else 0.0), act) + (resid if use_r else 0.0)
        err = np.abs(out - ref).max()
        assert err < 3e-4 * max(1.0, np.abs(ref).max()), f"skinny mode={mode} shape={shape} act={act}: max err {err}"
    lw = (1 + 0.2 * rng.standard_normal(N)).astype(np.float32)
    lb = rng.standard_normal(N).astype(np.float32)
    for S in (1, 2, 4, 8):
        if K % (32 * S) or N > 2048:  # the row kernel holds a row in registers: widths up to 2048 (the tower reduces [rows, width] only)
            continue
        out, h = np.empty((M, N), np.float32), np.empty((M, N), np.float32)
        gpu_lib.check(gpu_lib.lib.arp_op_skinny_gemm(mode, 0, _fp(A), _fp(W), _fp(bias), _fp(resid), _fp(out), M, N, K, S, _fp(lw), _fp(lb), 1e-5, _fp(h)))
        ref = prod + bias + resid
        assert np.abs(out - ref).max() < 3e-4 * max(1.0, np.abs(ref).max()), f"skinny split {S} mode={mode} shape={shape}"
        x = out.astype(np.float64)  # the LayerNorm is judged on the kernel's own x (its f32 rounding is not the subject)
        ln = (x - x.mean(1, keepdims=True)) / np.sqrt(x.var(1, keepdims=True) + 1e-5) * lw + lb
        assert np.abs(h - ln).max() < (1.6e-2 if mode == 1 else 2e-3) * max(1.0, np.abs(ln).max()), f"skinny split {S} LayerNorm mode={mode} shape={shape}"
        # the slab order is fixed: a second launch gives the same bits
        out2 = np.empty((M, N), np.float32)
        gpu_lib.check(gpu_lib.lib.arp_op_skinny_gemm(mode, 0, _fp(A), _fp(W), _fp(bias), _fp(resid), _fp(out2), M, N, K, S, None, None, 0.0, None))
        assert (out2 == out).all()


def test_skinny_gemm_rejects(gpu_lib):
    A = np.zeros((1100, 64), np.float32); W = np.zeros((16, 64), np.float32); out = np.zeros((1100, 16), np.float32)
    assert gpu_lib.lib.arp_op_skinny_gemm(2, 0, _fp(A), _fp(W), None, None, _fp(out), 1100, 16, 64, 0, None, None, 0.0, None) != 0   # M > 1024
    assert gpu_lib.lib.arp_op_skinny_gemm(2, 0, _fp(A), _fp(W), None, None, _fp(out), 8, 15, 64, 0, None, None, 0.0, None) != 0     # N % 16
    assert gpu_lib.lib.arp_op_skinny_gemm(2, 0, _fp(A), _fp(W), None, None, _fp(out), 8, 16, 48, 0, None, None, 0.0, None) != 0     # K % 32
    assert gpu_lib.lib.arp_op_skinny_gemm(0, 0, _fp(A), _fp(W), None, None, _fp(out), 8, 16, 64, 0, None, None, 0.0, None) != 0     # f32 mode


@pytest.mark.parametrize("kernel", ["2", "3"])
def test_gemm256_race_screen(gpu_lib, monkeypatch, kernel):
    """The pipelined kernel's LDS hand-offs are ordered by counted vmcnt + barriers: repeated launches on
    a chip-filling shape must be bit-identical to each other and correct (a race shows up as rare
    wrong tiles)."""
    monkeypatch.setenv("ARP_GEMM", kernel)
    rng = np.random.default_rng(5)
    for (M, N, K) in ((8192, 1536, 768), (4100, 768, 3072), (2048, 2304, 64), (2048, 2304, 128)):
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
        ref = bf16_round(A).astype(np.float64) @ bf16_round(W).astype(np.float64).T
        first = None
        for rep in range(6):
            out = np.empty((M, N), np.float32)
            gpu_lib.check(gpu_lib.lib.arp_op_gemm_nt(1, 0, _fp(A), _fp(W), None, None, _fp(out), M, N, K))
            if first is None:
                first = out
                err = np.abs(out - ref).max()
                assert err < 3e-4 * max(1.0, np.abs(ref).max()), f"gemm256 {M}x{N}x{K}: max err {err}"
            else:
                assert (out == first).all(), f"gemm256 {M}x{N}x{K}: launch {rep} differs from launch 0 in {(out != first).sum()} elements"


@pytest.mark.parametrize("D", [64, 128, 512, 768, 1024])
def test_layernorm(gpu_lib, D):
    rng = np.random.default_rng(D)
    rows = 37
    x = (rng.standard_normal((rows, D)) * 3 + 1).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    b = rng.standard_normal(D).astype(np.float32)
    for eps in (1e-5, 1e-6):
        out = np.empty_like(x)
        gpu_lib.check(gpu_lib.lib.arp_op_layernorm(_fp(x), _fp(w), _fp(b), _fp(out), rows, D, eps))
        x64 = x.astype(np.float64)
        mu = x64.mean(-1, keepdims=True)
        ref = (x64 - mu) / np.sqrt(((x64 - mu) ** 2).mean(-1, keepdims=True) + eps) * w + b
        assert np.abs(out - ref).max() < 5e-6 * max(1, np.abs(ref).max())


def _attn_ref(qkv, B, N, D, heads, causal):
    hd = D // heads
    q, k, v = np.split(qkv.astype(np.float64).reshape(B, N, 3 * D), 3, axis=-1)
    sh = lambda a: a.reshape(B, N, heads, hd).transpose(0, 2, 1, 3)
    q, k, v = sh(q), sh(k), sh(v)
    s = (q @ k.transpose(0, 1, 3, 2)) * hd ** -0.5
    if causal:
        s = np.where(np.triu(np.ones((N, N), bool), 1), -np.inf, s)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return (p @ v).transpose(0, 2, 1, 3).reshape(B * N, D)


ATTN_CASES = [  # B, N, D, heads, causal
    (3, 50, 768, 12, 0), (2, 77, 512, 8, 1), (2, 197, 768, 12, 0), (1, 257, 128, 2, 0), (5, 5, 64, 1, 0), (4, 12, 128, 8, 1),
    (2, 64, 128, 2, 1), (2, 33, 64, 2, 0),
    (1, 258, 128, 2, 0), (2, 129, 128, 2, 1),  # a last query block of one / two rows behind a multiple of eight full blocks: the x3 kernel's cooperative tail (also causal)
]


@pytest.mark.parametrize("case", ATTN_CASES)
# (0, 0): the exact-f32 MFMA kernel where head_dim is 64; (0, 3): the (hi, lo) binary16 MFMA kernel of the f16x3 encoder mode (f32-level: same bar)
@pytest.mark.parametrize("mode,impl", [(0, 1), (0, 0), (0, 3), (1, 1), (1, 0), (2, 1), (2, 0)])
def test_attention(gpu_lib, case, mode, impl):
    B, N, D, heads, causal = case
    rng = np.random.default_rng(N * 13 + D)
    qkv = (rng.standard_normal((B * N, 3 * D)) * 1.5).astype(np.float32)
    out = np.empty((B * N, D), np.float32)
    gpu_lib.check(gpu_lib.lib.arp_op_attention(mode, impl, _fp(qkv), _fp(out), B, N, D, heads, causal))
    rnd = {0: lambda x: x, 1: bf16_round, 2: lambda x: x.astype(np.float16).astype(np.float32)}[mode]
    ref = _attn_ref(rnd(qkv), B, N, D, heads, causal)
    err = np.abs(out - ref).max()
    tol = {0: 1e-5, 1: 2.5e-2, 2: 3.5e-3}[mode]  # bf16 / f16: P and the output are rounded to 8 / 11 significand bits
    assert err < tol, f"attention case={case} mode={mode} impl={impl}: max err {err}"
    if mode == 1:
        assert np.abs(out - ref).mean() < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1, 2, 4, 5, 15, 16, 17, 31, 32, 33, 36, 37, 64, 100, 129, 130, 224, 225, 229, 255, 256, 259, 260, 261, 271, 272, 273, 287, 288])
@pytest.mark.parametrize("causal", [0, 1])
def test_attention_x3_every_tile_boundary(gpu_lib, N, causal):
    """attn_x3_kernel permutes the keys of its 16-row score tiles (tile kt holds keys 32 (kt / 2) + 8 (i / 4) + 4 (kt % 2) + i % 4), so which tiles a sequence
    length needs, which of them carry masked keys and whether the last query block takes the cooperative path all change at lengths that are not multiples of
    16: one sequence length on either side of every such boundary up to the 288-key limit, with and without the causal mask, f32-level bar."""
    B, D, heads = 2, 128, 2
    rng = np.random.default_rng(N * 7 + causal)
    qkv = (rng.standard_normal((B * N, 3 * D)) * 1.5).astype(np.float32)
    out = np.full((B * N, D), np.nan, np.float32)
    gpu_lib.check(gpu_lib.lib.arp_op_attention(0, 3, _fp(qkv), _fp(out), B, N, D, heads, causal))
    err = np.abs(out - _attn_ref(qkv, B, N, D, heads, causal)).max()
    assert err < 1e-5, f"N={N} causal={causal}: max err {err}"


PRE_CASES = [(256, 256, False), (256, 256, True), (64, 64, False), (128, 96, False), (200, 300, False), (512, 512, True)]


@pytest.mark.parametrize("case", PRE_CASES)
def test_preprocess_bit_exact(gpu_lib, case):
    from arp_amd import clip, synth
    from oracle import preprocess as P
    H, W, use_crop = case
    fr = np.concatenate([synth.noise_frames(2, H, W, seed=H + W), synth.procgen_like_frames(1, H, W, seed=3)])
    got = clip.preprocess(fr, use_crop=use_crop)
    ref = P.preprocess(fr, use_crop=use_crop)
    assert got.shape == ref.shape
    nbad = int((got != ref).sum())
    assert nbad == 0, f"{nbad} of {ref.size} f32 values differ; max abs diff {np.abs(got - ref).max()}"


def _quant_fp4(x):
    """OCP e2m1: 0, 0.5, 1, 1.5, 2, 3, 4, 6 and their negatives; round-to-nearest-even on the one significand bit, saturating at 6 (what
    v_cvt_scalef32_pk_fp4_f32 and csrc/common.h::host_f2fp4 produce)"""
    x = np.asarray(x, np.float64)
    a = np.abs(x)
    grid = np.array([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])
    idx = np.clip(np.searchsorted(grid, a, side="left"), 1, 7)       # grid[idx - 1] <= a <= grid[idx] (a > 6: idx = 7 twice over)
    lo, hi = grid[idx - 1], grid[idx]
    mid = 0.5 * (lo + hi)
    q = np.where(a < mid, lo, np.where(a > mid, hi, np.where((idx - 1) % 2 == 0, lo, hi)))   # ties to the even code
    return np.sign(x) * np.minimum(q, 6.0)


@pytest.mark.parametrize("outliers", [False, True])
@pytest.mark.parametrize("shape", [(300, 512, 512), (257, 768, 768), (1030, 768, 3072), (77, 2304, 512)])
def test_gemm_f16c_corrects_the_operand_roundings(gpu_lib, shape, outliers):
    """ARP_MODE_F16C's product (gemm256 MIXC: binary16 K-tiles followed by e2m1 K-tiles on the scaled fp4 MFMA, round 5).  Two checks per shape:
    (i) EXACT restatement -- the same quantised operands multiplied in float64 (hi.W_hi + 2^-(1+sd) x4.dW4 + 2^-(13+sw) dx4.W4) agree with the kernel to
    f32 summation noise, so a wrong nibble order, k-slot assignment, segment offset or block scale shows at the size of a correction term, not hidden
    inside it; (ii) the point of it -- against the UNROUNDED product, plan 1 takes most of the weight rounding out and plan 2 most of both: on Gaussian
    rows the rms error falls to ~0.72x / ~0.2x of the plain binary16 product's; rows with 8-sigma outliers (1 % of the entries) keep more of it (measured
    0.78x / 0.46x): e2m1 saturates at 6, so an outlier's own correction terms are clipped -- one scale per tensor, no per-block scales."""
    M, N, K = shape
    rng = np.random.default_rng(M * 7 + N + K)
    A = (rng.standard_normal((M, K)) * (np.where(rng.random((M, K)) < 0.01, 8.0, 1.0) if outliers else 1.0)).astype(np.float32)  # LayerNorm-like rows
    W = (rng.standard_normal((N, K)) * 0.03).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    A64, W64 = A.astype(np.float64), W.astype(np.float64)
    ref = A64 @ W64.T + bias
    terms = np.abs(A64) @ np.abs(W64).T
    hi, whi = A.astype(np.float16).astype(np.float64), W.astype(np.float16).astype(np.float64)
    x4, dx4 = _quant_fp4(hi * 2.0), _quant_fp4((A64 - hi) * 2.0 ** 13)
    errs, rms = {}, {}
    for plan in (0, 1, 2):
        out = np.empty((M, N), np.float32)
        sc = (C.c_int32 * 2)()
        gpu_lib.check(gpu_lib.lib.arp_op_gemm_f16c(plan, _fp(A), _fp(W), _fp(bias), _fp(out), M, N, K, sc))
        sd, sw = sc[0], sc[1]
        assert 6 < np.abs(W64 - whi).max() * 2.0 ** sd <= 12 and 6 < np.abs(W64).max() * 2.0 ** sw <= 12
        want = hi @ whi.T + bias
        if plan >= 1:
            want = want + 2.0 ** -(1 + sd) * (x4 @ _quant_fp4((W64 - whi) * 2.0 ** sd).T)
        if plan >= 2:
            want = want + 2.0 ** -(13 + sw) * (dx4 @ _quant_fp4(W64 * 2.0 ** sw).T)
        tol = 3e-7 * terms + 1e-6 * (np.abs(want) + 1.0)   # f32 accumulation of K (+ K / 2) terms; a misplaced correction term is >= 1e-5 * terms
        assert (np.abs(out - want) <= tol).all(), (plan, float((np.abs(out - want) / tol).max()))
        errs[plan] = float((np.abs(out - ref) / terms).max())
        rms[plan] = float(np.sqrt((((out - ref) / terms) ** 2).mean()))
    print(f"f16c gemm {shape}: max err / sum|a||w| plan 0 {errs[0]:.2e}, 1 {errs[1]:.2e}, 2 {errs[2]:.2e}; rms {rms[0]:.2e} {rms[1]:.2e} {rms[2]:.2e}")
    assert rms[1] < 0.85 * rms[0] and rms[2] < (0.6 if outliers else 0.3) * rms[0], (errs, rms)


@pytest.mark.parametrize("shape", [(300, 512, 256), (256, 256, 128), (1030, 768, 3072), (77, 1024, 384)])
def test_gemm_fp8_exact_on_representable_operands(gpu_lib, shape):
    """The fp8 instances of the 256x256 kernel (v_mfma_scale_f32_16x16x128_f8f6f4, BASELINE configs[4]).  Operands that are exactly
    representable in e4m3 make the check EXACT up to f32 summation: any error in the fragment layout, the k-slot assignment of the
    two register halves, the unit block scales or the alpha / bias / residual epilogue shows up at O(1), not at the 6 % of an
    e4m3 rounding step.  The e4m3 OUTPUT path (QuickGELU, out_scale, saturating convert) is checked against the oracle's rounding."""
    from oracle import clip_np as O
    M, N, K = shape
    rng = np.random.default_rng(M + N + K)
    A = O.quant_e4m3(rng.standard_normal((M, K)) * 4).astype(np.float32)
    W = O.quant_e4m3(rng.standard_normal((N, K)) * 8).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    resid = rng.standard_normal((M, N)).astype(np.float32)
    alpha = 1.0 / 64
    ref = alpha * (A.astype(np.float64) @ W.astype(np.float64).T)
    out = np.empty((M, N), np.float32)
    gpu_lib.check(gpu_lib.lib.arp_op_gemm_fp8(0, _fp(A), _fp(W), _fp(bias), _fp(resid), _fp(out), M, N, K, alpha, 0, 1.0))
    r = ref + bias + resid
    # the error scales with the magnitude of the TERMS (they cancel), not of the result; measured 4e-6 of sum |a||w| -- the scaled
    # fp8 MFMA does not carry a full f32 significand through its 128-term sum -- against O(1) for any layout / k-slot mistake
    tol = 2e-5 * alpha * (np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T) + 2e-6 * (np.abs(r) + 1.0)
    assert (np.abs(out - r) <= tol).all(), float((np.abs(out - r) / tol).max())
    gpu_lib.check(gpu_lib.lib.arp_op_gemm_fp8(0, _fp(A), _fp(W), None, None, _fp(out), M, N, K, alpha, 0, 1.0))
    assert (np.abs(out - ref) <= tol).all(), float((np.abs(out - ref) / tol).max())
    # IEEE-half output (in_proj of the fp8 attention projections): alpha * acc + bias rounded to f16
    gpu_lib.check(gpu_lib.lib.arp_op_gemm_fp8(0, _fp(A), _fp(W), _fp(bias), None, _fp(out), M, N, K, alpha, 2, 1.0))
    want16 = (ref + bias).astype(np.float16).astype(np.float32)
    assert (np.abs(out - want16) <= tol + np.abs(want16) * 2.0 ** -10).all()
    # e4m3 output: 0.25 * quickgelu(alpha * acc + bias), saturating; allow one rounding step where f32 noise crosses a boundary
    gpu_lib.check(gpu_lib.lib.arp_op_gemm_fp8(1, _fp(A), _fp(W), _fp(bias), None, _fp(out), M, N, K, alpha, 1, 0.25))
    want = O.quant_e4m3(0.25 * _act(ref + bias, 1))
    step = np.maximum(np.abs(want), 2.0 ** -6) / 8
    bad = np.abs(out - want) > 1e-6
    assert bad.mean() < 2e-3 and (np.abs(out - want) <= step * 1.01 + 1e-6).all(), (bad.mean(), np.abs(out - want).max())


# ---- kernels of the policy train step (round 2): TN weight-gradient GEMMs, the ReLU-backward GEMM epilogue, the fused dY pass ----

def _rnd16(mode):
    return {1: bf16_round, 2: lambda x: x.astype(np.float16).astype(np.float32)}[mode]


def _ulp16(mode):
    return {1: 2.0 ** -8, 2: 2.0 ** -11}[mode]  # half a unit in the last place, relative, of a value stored in the operand type


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("case", [  # M, N, K, ksplit, tile256
    (128, 128, 64, 1, 0), (256, 384, 704, 3, 0), (128, 256, 1344, 5, 0),
    (256, 256, 64, 1, 1), (256, 256, 128, 1, 1), (256, 512, 704, 1, 1), (512, 256, 1344, 5, 1), (768, 768, 2112, 8, 1),
    (768, 768, 2112, 16, 1), (768, 768, 2112, 24, 1), (256, 256, 192, 3, 1),
])
def test_gemm_tn(gpu_lib, mode, case):
    """dW = alpha * A^T B from two ROW-major operands (gemm_tn.hip): the 128-tile kernel and the pipelined 256-tile kernel (ring of
    four K-tiles, hand-counted waits), K-tile counts below / at / above the ring depth, ragged K-splits, the per-XCD slice placement
    (ksplit % 8 == 0).  Exact up to f32 summation: operands are pre-rounded to the operand type."""
    M, N, K, S, t256 = case
    rng = np.random.default_rng(M + 3 * N + 7 * K + S)
    rnd = _rnd16(mode)
    A = rnd(rng.standard_normal((K, M)).astype(np.float32))
    B = rnd((rng.standard_normal((K, N)) * 0.25).astype(np.float32))
    alpha = 0.5
    out = np.full((M, N), np.nan, np.float32)
    gpu_lib.check(gpu_lib.lib.arp_op_gemm_tn(mode, t256, S, _fp(A), _fp(B), _fp(out), M, N, K, alpha))
    ref = alpha * (A.astype(np.float64).T @ B.astype(np.float64))
    tol = 2e-6 * alpha * (np.abs(A).astype(np.float64).T @ np.abs(B).astype(np.float64)) + 1e-6
    assert np.isfinite(out).all()
    assert (np.abs(out - ref) <= tol).all(), float((np.abs(out - ref) / tol).max())


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("shape", [(512, 256, 128), (1000, 520, 192), (2056, 768, 768), (255, 8, 64)])
def test_gemm_relu_bwd_epilogue(gpu_lib, mode, shape):
    """out = (A W^T) * (mask > 0) and the column sums of what was stored -- gemm256's masked epilogue (the adapter's dH1 and Dense_0
    bias gradient).  The mask holds positives, +0 and -0 (a ReLU output can be either zero); masked entries must be EXACT zeros, kept
    entries within a rounding step of the operand type, the column sums equal to the sums of the returned values."""
    M, N, K = shape
    rng = np.random.default_rng(M + N + K)
    rnd = _rnd16(mode)
    A = rnd(rng.standard_normal((M, K)).astype(np.float32))
    W = rnd((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    r = rng.standard_normal((M, N)).astype(np.float32)
    mask = np.where(r > 0, r, np.where(r < -1.0, -0.0, 0.0)).astype(np.float32)
    mask = rnd(mask)
    out = np.full((M, N), np.nan, np.float32)
    cs = np.full(N, np.nan, np.float32)
    gpu_lib.check(gpu_lib.lib.arp_op_gemm_relu_bwd(mode, _fp(A), _fp(W), _fp(mask), _fp(out), _fp(cs), M, N, K))
    keep = mask > 0
    assert keep.any() and (~keep).any() and np.signbit(mask[~keep]).any()
    assert (out[~keep] == 0).all()
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    tol = 1.01 * _ulp16(mode) * np.abs(ref) + 2e-6 * (np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64).T) + 1e-7
    assert (np.abs(out - ref)[keep] <= tol[keep]).all(), float((np.abs(out - ref)[keep] / tol[keep]).max())
    want = out.astype(np.float64).sum(0)
    assert (np.abs(cs - want) <= 2e-6 * np.abs(out).astype(np.float64).sum(0) + 1e-6).all()


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("geom", [(8, 128, 5, 128), (130, 64, 3, 256), (128, 128, 9, 768), (40, 32, 2, 128), (257, 128, 2, 384)])
def test_adapter_dy_fused(gpu_lib, mode, geom):
    """The fused adapter-backward pass (adapter_bwd.hip; arp_dt/ARPDT.py:462-484 differentiated): dY = dz Wi, dApre = res dY (A > 0),
    Dense_1's bias gradient (column sums over rows and tokens of the ROUNDED dApre) and d loss / d res = sum dY (A - x).  Row counts
    below, at and across the 128-row block, every supported E, widths of one to six 128-column tiles, -0 in the mask."""
    R, E, tokens, D = geom
    Kin = tokens * D
    rng = np.random.default_rng(R * 13 + E + tokens * 5 + D)
    rnd = _rnd16(mode)
    dz = rnd((rng.standard_normal((R, E)) * 4).astype(np.float32))
    Wi = rnd((rng.standard_normal((E, Kin)) / np.sqrt(E)).astype(np.float32))
    r = rng.standard_normal((R, Kin)).astype(np.float32)
    A = rnd(np.where(r > 0, r, np.where(r < -1.2, -0.0, 0.0)).astype(np.float32))
    x = rng.standard_normal((R, Kin)).astype(np.float32)
    rw = float(rng.standard_normal())
    out = np.full((R, Kin), np.nan, np.float32)
    cs = np.full(D, np.nan, np.float32)
    dres = np.full(1, np.nan, np.float32)
    gpu_lib.check(gpu_lib.lib.arp_op_adapter_dy(mode, _fp(dz), _fp(Wi), _fp(A), _fp(x), rw, _fp(out), _fp(cs), _fp(dres), R, E, tokens, D))
    res = 1.0 / (1.0 + np.exp(-np.float64(np.float32(rw))))
    dY = dz.astype(np.float64) @ Wi.astype(np.float64)
    mag = np.abs(dz).astype(np.float64) @ np.abs(Wi).astype(np.float64)
    keep = A > 0
    assert (out[~keep] == 0).all() and np.signbit(A[~keep]).any()
    ref = res * dY
    tol = 1.01 * _ulp16(mode) * np.abs(ref) + 3e-6 * mag + 1e-7
    assert (np.abs(out - ref)[keep] <= tol[keep]).all(), float((np.abs(out - ref)[keep] / tol[keep]).max())
    want_cs = out.astype(np.float64).reshape(R, tokens, D).sum((0, 1))
    assert (np.abs(cs - want_cs) <= 3e-6 * np.abs(out).astype(np.float64).reshape(R, tokens, D).sum((0, 1)) + 1e-6).all()
    want_dres = float((dY * (A.astype(np.float64) - x)).sum())
    tol_dres = 3e-6 * float((mag * np.abs(A.astype(np.float64) - x)).sum()) + 1e-6
    assert abs(float(dres[0]) - want_dres) <= tol_dres, (float(dres[0]), want_dres, tol_dres)
