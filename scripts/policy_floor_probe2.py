#!/usr/bin/env python3
"""Second bisection of the corrected f16 policy's error floor (3.4e-4 median on the GPU where the fp64 emulation of the same roundings leaves 1.1e-4,
gpurun_out/r6_floor_per_seed.txt): which side of the adapter's output is it on?  Per seed, against oracle/arpdt_torch in fp64 with the SAME parameters:

  default      the f16 step with corrections (plan 22e) as it runs
  y = x        residual_weight = -30: the adapter's output has no weight, y is the encodings -> what everything BEHIND the mix leaves
  y = a        residual_weight = +30: the adapter's output alone
  chain        the adapter computed OUTSIDE the step with arp_op_gemm_f16c (plan 2, the f32 hidden rows handed over on the host: the unit-tested product,
               tests/test_ops_gpu.py::test_gemm_f16c_corrects_the_operand_roundings), mixed in fp64, and fed to the step as encodings with residual_weight = -30
  exact y      the fp64 adapter + mix fed the same way (the floor behind the mix on the real y distribution)

    python scripts/policy_floor_probe2.py [n_seeds]      (needs a GPU; test infrastructure: uses the oracle)
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arp_amd import _ffi, synth_policy as S  # noqa: E402
from arp_amd.train import PolicyConfig, PolicyTrainer  # noqa: E402
from oracle import arpdt_torch as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg, ocfg = PolicyConfig(lambda_ret=0.01), O.PolicyConfig(lambda_ret=0.01)
W1, B1, W2, B2 = "AdapterMLP_0/Dense_0/kernel", "AdapterMLP_0/Dense_0/bias", "AdapterMLP_0/Dense_1/kernel", "AdapterMLP_0/Dense_1/bias"


def fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def oracle(P, enc, act, rtg):
    r = O.forward({k: torch.from_numpy(np.asarray(v)).double() for k, v in P.items()}, ocfg, torch.from_numpy(np.asarray(enc, np.float64)), torch.from_numpy(act).long(),
                  torch.from_numpy(rtg).double())
    return r["action_pred"].numpy(), r["return_pred"].numpy()


def gemm_c(x, w_flax, b):
    """relu(x W + b) on the GPU's corrected product; W in Flax layout [in, out] -> the op's [N][K] rows"""
    M, K = x.shape
    Wt = np.ascontiguousarray(w_flax.T.astype(np.float32))
    out = np.empty((M, Wt.shape[0]), np.float32)
    sc = (C.c_int32 * 2)()
    _ffi.check(_ffi.lib.arp_op_gemm_f16c(2, fp(np.ascontiguousarray(x, np.float32)), fp(Wt), fp(np.ascontiguousarray(b, np.float32)), fp(out), M, Wt.shape[0], K, sc))
    return np.maximum(out, 0.0)


def with_rw(P, v):
    Q = dict(P)
    Q["residual_weight"] = np.full_like(np.asarray(P["residual_weight"]), v)
    return Q


tr = PolicyTrainer(cfg, mode="f16", adapter_corrections=True)
names = ["default", "y = x", "y = a", "chain", "exact y", "a: chain vs fp64 (rel rms)", "a: rn16 vs fp64 (rel rms)"]
rows = {k: [] for k in names}


def run(P, enc, act, rtg, ref):
    tr.set_params(P)
    tr.set_batch(np.asarray(enc, np.float32), act, rtg)
    out = tr.forward()
    return max(float(np.abs(out["action_pred"] - ref[0]).max()), float(np.abs(out["return_pred"] - ref[1]).max()))


for seed in range(n):
    s = 100 + 7 * seed
    P = S.policy_params(cfg, seed=s)
    enc, act, rtg = S.policy_batch(cfg, 2, seed=s + 1)
    ref = oracle(P, enc, act, rtg)
    rows["default"].append(run(P, enc, act, rtg, ref))
    for name, v in (("y = x", -30.0), ("y = a", 30.0)):
        Q = with_rw(P, v)
        rows[name].append(run(Q, enc, act, rtg, oracle(Q, enc, act, rtg)))
    x = enc.reshape(-1, cfg.enc_dim)
    x64 = x.astype(np.float64)
    a64 = np.maximum(np.maximum(x64 @ P[W1].astype(np.float64) + P[B1], 0.0) @ P[W2].astype(np.float64) + P[B2], 0.0)
    a_c = gemm_c(gemm_c(x, P[W1], P[B1]), P[W2], P[B2]).astype(np.float64)
    res = 1.0 / (1.0 + np.exp(-float(np.asarray(P["residual_weight"]).reshape(-1)[0])))
    Q = with_rw(P, -30.0)
    rows["chain"].append(run(Q, (res * a_c + (1 - res) * x64).reshape(enc.shape), act, rtg, ref))
    rows["exact y"].append(run(Q, (res * a64 + (1 - res) * x64).reshape(enc.shape), act, rtg, ref))
    nz = a64 > 0
    rows["a: chain vs fp64 (rel rms)"].append(float(np.sqrt(np.mean(((a_c - a64)[nz] / a64[nz].clip(1e-3)) ** 2))))
    rows["a: rn16 vs fp64 (rel rms)"].append(float(np.sqrt(np.mean(((a64.astype(np.float16).astype(np.float64) - a64)[nz] / a64[nz].clip(1e-3)) ** 2))))
    print(f"# seed {seed}: " + "  ".join(f"{k} {rows[k][-1]:.2e}" for k in names), flush=True)
tr.close()
for k in names:
    print(f"{k:30s} max {max(rows[k]):.2e} median {np.median(rows[k]):.2e}   " + " ".join(f"{v:.2e}" for v in rows[k]))
