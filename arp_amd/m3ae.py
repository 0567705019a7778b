"""The frozen M3AE image encoder on MI355X.

Mirrors ``MaskedMultimodalAutoencoder.forward_representation`` for the image-only call ARP-DT makes
(/root/reference/arp_dt/models/m3ae/model.py:471-496, called at arp_dt/ARPDT.py:451-458 under
``stop_gradient``): float frames ``[n, 256, 256, 3]`` (already normalised) -> ``[n, 257, 768]``.
Weights are the Flax parameter tree flattened with '/' (what ``load_m3ae_model_vars`` returns under
``["params"]``).  All compute is in libarp_hip.so (same kernels as the CLIP image tower).
"""
import ctypes as C
import json
from dataclasses import dataclass

import numpy as np

from . import _ffi
from ._ffi import MODE_BF16, MODE_F16, MODE_F16C, MODE_F16X3, MODE_F32, check, lib


@dataclass(frozen=True)
class EncoderConfig:
    """get_transformer_by_config("base") (m3ae/model.py:935-941) at 256x256, patch 16."""
    patch: int = 16
    width: int = 768
    layers: int = 12
    heads: int = 12
    mlp_ratio: int = 4
    img_res: int = 256

    @property
    def tokens(self):
        return (self.img_res // self.patch) ** 2 + 1


def flops_per_frame(cfg):
    n, d = cfg.tokens, cfg.width
    macs = (n - 1) * cfg.patch * cfg.patch * 3 * d + cfg.layers * (n * (4 * d * d + 2 * cfg.mlp_ratio * d * d) + cfg.heads * 2 * n * n * (d // cfg.heads))
    return 2 * macs


class M3AEEncoder:
    def __init__(self, cfg, params, mode="bf16", device=0, max_frames=128, attn_impl=0):
        """mode: "f16" / "bf16" (16-bit GEMM operands: throughput modes with a stated error), "f32" (f32-input MFMA: the parity mode) or
        "f16x3" (every GEMM operand an (hi, lo) pair of binary16 values, three 16-bit MFMAs per product, attention / LayerNorm in f32: f32-level error at
        about twice the f32 mode's speed)."""
        _ffi.require_gpu()
        self.cfg = cfg
        c = _ffi.EncCfg(cfg.patch, cfg.width, cfg.layers, cfg.heads, cfg.mlp_ratio, cfg.img_res, {"bf16": MODE_BF16, "f16": MODE_F16, "f32": MODE_F32, "f16x3": MODE_F16X3, "f16c": MODE_F16C}[mode],
                        device, max_frames, attn_impl)
        h = C.c_void_p()
        check(lib.arp_enc_create(C.byref(c), C.byref(h)))
        self._h = h
        for name, val in params.items():
            a = np.require(np.asarray(val, dtype=np.float32), requirements="C")
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            check(lib.arp_enc_load_weight(h, name.encode(), _ffi.as_ptr(a, C.c_float), shape, a.ndim))
        check(lib.arp_enc_finalize_weights(h))

    def forward_representation(self, images):
        x = np.require(np.asarray(images, dtype=np.float32), requirements="C")
        if x.ndim != 4 or x.shape[1:] != (self.cfg.img_res, self.cfg.img_res, 3):
            raise ValueError(f"images must be float [n, {self.cfg.img_res}, {self.cfg.img_res}, 3]")
        out = np.empty((x.shape[0], self.cfg.tokens, self.cfg.width), np.float32)
        check(lib.arp_enc_forward(self._h, _ffi.as_ptr(x, C.c_float), x.shape[0], _ffi.as_ptr(out, C.c_float)))
        return out

    def set_streams(self, n_streams=2, first_part_frames=0, min_part_frames=0):
        """Part streams of a call (``arp_enc_set_streams``): contiguous parts of the frames on HIP streams of their own; same encodings for every setting."""
        check(lib.arp_enc_set_streams(self._h, int(n_streams), int(first_part_frames), int(min_part_frames)))

    def profile(self, on=True):
        check(lib.arp_enc_profile_enable(self._h, int(on)))

    def profile_read(self):
        buf = C.create_string_buffer(1 << 16)
        check(lib.arp_enc_profile_json(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())

    def close(self):
        if getattr(self, "_h", None):
            lib.arp_enc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
