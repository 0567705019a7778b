#!/usr/bin/env python3
"""GEMM micro-benchmark on the GPU box: the ViT-B/32 batch-1024 shapes (and a K sweep) on both kernels."""
import ctypes as C
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from arp_amd import _ffi

SHAPES = {  # name: (M, N, K, act, resid, out_f32)
    "qkv": (51200, 2304, 768, 0, 0, 0),
    "out_proj": (51200, 768, 768, 0, 1, 1),
    "c_fc": (51200, 3072, 768, 1, 0, 0),
    "c_proj": (51200, 768, 3072, 0, 1, 1),
    "qkv_half": (25600, 2304, 768, 0, 0, 0),
    "out_proj_half": (25600, 768, 768, 0, 1, 1),
    "c_fc_half": (25600, 3072, 768, 1, 0, 0),
    "c_proj_half": (25600, 768, 3072, 0, 1, 1),
    "sq4096": (4096, 4096, 4096, 0, 0, 0),
    "sq8192": (8192, 8192, 8192, 0, 0, 0),
    "k64_store_only": (51200, 3072, 64, 0, 0, 0),
    "k64_f32out": (51200, 768, 64, 0, 0, 1),
    "k64_resid": (51200, 768, 64, 0, 1, 1),
    "k768_noepi": (51200, 3072, 768, 0, 0, 0),
    "k1536": (51200, 3072, 1536, 0, 0, 0),
    "k3072": (51200, 3072, 3072, 0, 0, 0),
}
import os
MODE = {"bf16": 1, "f16": 2}[os.environ.get("GEMM_BENCH_MODE", "f16")]
names = sys.argv[1:] or list(SHAPES)
for name in names:
    M, N, K, act, resid, f32 = SHAPES[name]
    for kern in (1, 2, 3):
        ms = C.c_float()
        _ffi.check(_ffi.lib.arp_op_gemm_bench(MODE, kern, act, resid, f32, M, N, K, 20, C.byref(ms)))
        print(f"{name:12s} kernel={ {1: '128', 2: '256', 3: '2w '}[kern] } M={M} N={N} K={K}: {ms.value * 1e3:8.1f} us  {2.0 * M * N * K / ms.value / 1e9:7.1f} TFLOP/s", flush=True)
