#!/usr/bin/env python3
"""Measuring stick only (never used by the product): torch.matmul -> hipBLASLt/rocBLAS on the ViT-B/32 batch-1024 GEMM shapes."""
import time
import torch

dev = torch.device("cuda:0")
shapes = {"qkv": (51200, 2304, 768), "out_proj": (51200, 768, 768), "c_fc": (51200, 3072, 768), "c_proj": (51200, 768, 3072), "sq4096": (4096, 4096, 4096)}
for name, (M, N, K) in shapes.items():
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        c = a @ w.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name:9s} torch.matmul bf16 (no bias / activation / residual): {dt*1e6:8.1f} us  {2.0*M*N*K/dt/1e12:7.1f} TFLOP/s", flush=True)
