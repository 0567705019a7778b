#!/usr/bin/env python3
"""Where does attn_x3_kernel (arp_op_attention impl 3) differ from the float64 attention?  (round-4 debugging aid)"""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from arp_amd import _ffi as lib

def ref(qkv, B, N, D, heads):
    hd = D // heads
    q, k, v = np.split(qkv.astype(np.float64).reshape(B, N, 3 * D), 3, axis=-1)
    sh = lambda a: a.reshape(B, N, heads, hd).transpose(0, 2, 1, 3)
    q, k, v = sh(q), sh(k), sh(v)
    s = (q @ k.transpose(0, 1, 3, 2)) * hd ** -0.5
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s); p /= p.sum(-1, keepdims=True)
    return (p @ v).transpose(0, 2, 1, 3).reshape(B * N, D)

fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
def run(qkv, B, N, D, heads, impl=3):
    out = np.empty((B * N, D), np.float32)
    lib.lib.arp_op_attention(0, impl, fp(qkv), fp(out), B, N, D, heads, 0)
    return out
B, N, D, heads = 1, 257, 128, 2
rng = np.random.default_rng(N * 13 + D)
qkv0 = (rng.standard_normal((B * N, 3 * D)) * 1.5).astype(np.float32)
Pg = np.zeros((N, N)); Pr = np.zeros((N, N))
for t in range(5):
    q = qkv0.copy(); q[:, 2*D:] = 0
    for d in range(64):
        key = d + 64 * t
        if key < N: q[key, 2*D + d] = 1.0
    got = run(q, B, N, D, heads); want = ref(q, B, N, D, heads)
    n = min(64, N - 64 * t)
    Pg[:, 64*t:64*t+n] = got[:, :n]; Pr[:, 64*t:64*t+n] = want[:, :n]
rel = (Pg - Pr) / np.maximum(Pr, 1e-30)
for r in (55, 56, 122, 100):
    top = np.argsort(-Pr[r])[:12]
    print(f"row {r}: sum got {Pg[r].sum():.7f}; top keys {top.tolist()}\n   P    {np.round(Pr[r, top], 4).tolist()}\n   rel  {[f'{x:+.1e}' for x in rel[r, top]]}", flush=True)
big = np.abs(Pg - Pr).max(1)
print("rows by max abs P error:", np.argsort(-big)[:8].tolist(), np.round(np.sort(big)[::-1][:8], 7).tolist())
