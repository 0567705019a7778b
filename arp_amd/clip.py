"""Host-side handle over the HIP CLIP labeller.

Stands where ``clip.load("ViT-B/16", device)`` + ``model(images, text)`` stand in the reference
(/root/reference/arp_dt/label_reward.py:125-146): holds the weights on one GPU, caches the text
features, and turns uint8 NHWC frames into rewards.  All compute is in libarp_hip.so.
"""
import ctypes as C
import json
from dataclasses import dataclass

import numpy as np

from . import _ffi
from ._ffi import MODE_BF16, MODE_F16, MODE_F32, check, lib


@dataclass(frozen=True)
class ClipConfig:
    """Geometry of an openai/CLIP ViT model (reference configs: arp_dt/models/openai/model.py:59-79)."""
    patch: int = 32
    width: int = 768
    layers: int = 12
    heads: int = 12
    embed: int = 512
    img_res: int = 224
    txt_width: int = 512
    txt_layers: int = 12
    txt_heads: int = 8
    ctx: int = 77
    vocab: int = 49408

    @property
    def grid(self):
        return self.img_res // self.patch

    @property
    def tokens(self):
        return self.grid * self.grid + 1


VIT_B32 = ClipConfig(patch=32)
VIT_B16 = ClipConfig(patch=16)  # what the reference loads (label_reward.py:126)
MODELS = {"ViT-B/32": VIT_B32, "ViT-B/16": VIT_B16}


def flops_per_frame(cfg):
    """Algorithmic FLOPs of the image tower for one frame (2 per MAC)."""
    n, d, g = cfg.tokens, cfg.width, cfg.grid
    lin = n * (d * 3 * d + d * d + 2 * d * 4 * d)
    att = cfg.heads * 2 * n * n * (d // cfg.heads)
    macs = g * g * 3 * cfg.patch * cfg.patch * d + cfg.layers * (lin + att) + d * cfg.embed
    return 2 * macs


class DeviceBuffer:
    """A raw hipMalloc allocation (no torch anywhere on this path)."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        check(lib.arp_dev_malloc(C.byref(p), nbytes))
        self.ptr, self.nbytes = p, nbytes

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(lib.arp_memcpy_h2d(self.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return self

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        check(lib.arp_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib.arp_dev_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Event:
    def __init__(self):
        p = C.c_void_p()
        check(lib.arp_event_create(C.byref(p)))
        self.ptr = p

    def __del__(self):
        try:
            lib.arp_event_destroy(self.ptr)
        except Exception:
            pass


def elapsed_ms(start, stop):
    ms = C.c_float()
    check(lib.arp_event_elapsed_ms(start.ptr, stop.ptr, C.byref(ms)))
    return ms.value


class ClipLabeller:
    """``ClipLabeller(cfg, state_dict).set_text(tokens).label(frames) -> float32 rewards``.

    ``state_dict`` maps openai/CLIP names to arrays (numpy or torch tensors; a real
    ``clip.load(...)[0].state_dict()`` works as is).  ``mode``: "f16" (default: IEEE-half MFMA operands --
    what openai/CLIP itself runs on a GPU -- f32 accumulate / residual / LayerNorm / softmax; rewards within 1e-4 cosine of
    the fp32 reference at the bf16 rate), "bf16" (same rate, 8-bit significands: 3-8e-4) or "f32" (f32-input MFMA, 1e-7).
    """

    def __init__(self, cfg, state_dict, mode="f16", device=0, max_batch=1024, attn_impl=0, n_streams=3, fp8_mlp=False):
        """fp8_mlp: run the vision tower's c_fc / c_proj GEMMs on e4m3 operands (scaled fp8 MFMA, twice the 16-bit rate) -- the
        "fp8 MFMA GEMMs" of BASELINE configs[4]: a throughput mode for the FROZEN towers of the fine-tune step (features ~1e-2 off
        the f32 towers), never the labelling default (1e-4 needs the 11 significand bits of f16)."""
        _ffi.require_gpu()
        self.cfg = cfg
        self.max_batch = int(max_batch)
        self.mode = {"bf16": MODE_BF16, "f16": MODE_F16, "f32": MODE_F32}[mode]
        c = _ffi.ClipCfg(cfg.patch, cfg.width, cfg.layers, cfg.heads, cfg.embed, cfg.img_res, cfg.txt_width,
                         cfg.txt_layers, cfg.txt_heads, cfg.ctx, cfg.vocab, self.mode, device, max_batch, attn_impl, n_streams)
        h = C.c_void_p()
        check(lib.arp_clip_create(C.byref(c), C.byref(h)))
        self._h = h
        for name, val in state_dict.items():
            if name in ("input_resolution", "context_length", "vocab_size"):  # dropped by the reference too
                continue
            if hasattr(val, "detach"):
                val = val.detach().float().cpu().numpy()
            a = np.require(np.asarray(val, dtype=np.float32), requirements="C")  # keeps 0-d (logit_scale) 0-d
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            check(lib.arp_clip_load_weight(h, name.encode(), _ffi.as_ptr(a, C.c_float), shape, a.ndim))
        if fp8_mlp:
            check(lib.arp_clip_set_fp8_mlp(h, 2 if fp8_mlp in (2, "all") else 1))  # 2 / "all": in_proj and out_proj on fp8 operands too
        check(lib.arp_clip_finalize_weights(h))

    def close(self):
        if getattr(self, "_h", None):
            lib.arp_clip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- text ------------------------------------------------------------------------------------
    def set_text(self, tokens):
        t = np.ascontiguousarray(np.asarray(tokens, dtype=np.int32))
        if t.ndim == 1:
            t = t[None]
        if t.shape[1] != self.cfg.ctx:
            raise ValueError(f"tokens must be [n_prompts, {self.cfg.ctx}]")
        check(lib.arp_clip_set_text(self._h, _ffi.as_ptr(t, C.c_int32), t.shape[0]))
        self._n_prompts = t.shape[0]
        return self

    def set_prompt_reduce(self, mode):
        """"first" (default): a reward is logits_per_text[0] -- prompt 0 whatever was cached, the offline pass (label_reward.py:146).
        "mean": logits_per_text.mean(axis=0) over the cached prompts -- the rollout loop's branch for a LIST of prompts (envs/vl_reward.py:19-22)."""
        check(lib.arp_clip_set_prompt_reduce(self._h, {"first": 0, "mean": 1, 0: 0, 1: 1}[mode]))
        return self

    def text_features(self):
        out = np.empty((self._n_prompts, self.cfg.embed), np.float32)
        check(lib.arp_clip_get_text_features(self._h, _ffi.as_ptr(out, C.c_float)))
        return out

    # -- images ----------------------------------------------------------------------------------
    @staticmethod
    def _frames(frames):
        f = np.ascontiguousarray(np.asarray(frames))
        if f.dtype != np.uint8 or f.ndim != 4 or f.shape[-1] != 3:
            raise ValueError("frames must be uint8 [N, H, W, 3]")
        return f

    @staticmethod
    def pin_host(a):
        """Pin a host array that will be passed to :meth:`label` repeatedly (hipHostRegister): its uploads become asynchronous DMA."""
        check(lib.arp_host_register(a.ctypes.data, a.nbytes))
        return a

    @staticmethod
    def unpin_host(a):
        check(lib.arp_host_unregister(a.ctypes.data))

    def label(self, frames, use_crop=False):
        """compute_reward (label_reward.py:132-146): uint8 [N,H,W,3] -> float32 [N]."""
        f = self._frames(frames)
        out = np.empty(f.shape[0], np.float32)
        check(lib.arp_clip_label(self._h, _ffi.as_ptr(f, C.c_uint8), f.shape[0], f.shape[1], f.shape[2],
                                 int(bool(use_crop)), _ffi.as_ptr(out, C.c_float)))
        return out

    def label_submit(self, slot, frames, use_crop=False):
        """Asynchronous :meth:`label` on slot 0 / 1 (at most ``max_batch`` frames): returns once upload, pass and download are enqueued.
        Submit call i+1 before collecting call i and the GPU never drains between calls.  ``frames`` is kept alive here."""
        f = self._frames(frames)
        check(lib.arp_clip_label_submit(self._h, int(slot), _ffi.as_ptr(f, C.c_uint8), f.shape[0], f.shape[1], f.shape[2], int(bool(use_crop))))
        self._pending = getattr(self, "_pending", {})
        self._pending[int(slot)] = f

    def label_collect(self, slot):
        f = getattr(self, "_pending", {}).pop(int(slot), None)
        out = np.empty(f.shape[0] if f is not None else 1, np.float32)  # (an empty slot: the library reports the misuse)
        check(lib.arp_clip_label_collect(self._h, int(slot), _ffi.as_ptr(out, C.c_float)))
        return out

    def encode_image(self, frames, use_crop=False, normalize=False):
        f = self._frames(frames)
        out = np.empty((f.shape[0], self.cfg.embed), np.float32)
        check(lib.arp_clip_encode_image(self._h, _ffi.as_ptr(f, C.c_uint8), f.shape[0], f.shape[1], f.shape[2],
                                        int(bool(use_crop)), int(bool(normalize)), _ffi.as_ptr(out, C.c_float)))
        return out

    def encode_image_multiscale(self, frames, pil=False, use_crop=False):
        """Frozen-tower side of the fine-tune step (finetune_module/clip_multiscale_adapter.py:120-149): per-block
        CLS features [n, layers*width] and the un-normalised CLIP feature [n, embed], through the fine-tune transform -- or, ``pil=True``,
        through the label transform (Pillow bicubic + normalise; ``use_crop`` as in :meth:`label`): what the rollout loop's adapter rewards
        feed the fine-tuned model (envs/vl_reward.py:44-79)."""
        f = self._frames(frames)
        inter = np.empty((f.shape[0], self.cfg.layers * self.cfg.width), np.float32)
        fin = np.empty((f.shape[0], self.cfg.embed), np.float32)
        if pil:
            check(lib.arp_clip_encode_image_multiscale_pil(self._h, _ffi.as_ptr(f, C.c_uint8), f.shape[0], f.shape[1], f.shape[2], int(bool(use_crop)),
                                                           _ffi.as_ptr(inter, C.c_float), _ffi.as_ptr(fin, C.c_float)))
        else:
            check(lib.arp_clip_encode_image_multiscale(self._h, _ffi.as_ptr(f, C.c_uint8), f.shape[0], f.shape[1], f.shape[2],
                                                       _ffi.as_ptr(inter, C.c_float), _ffi.as_ptr(fin, C.c_float)))
        return inter, fin

    def encode_multiscale_to(self, frames, tokens, bufs):
        """Device-resident variant for the fine-tune step: ``bufs`` = (img_inter, img_final, txt_inter, txt_final) DeviceBuffers
        sized for these frames / prompts; the features stay in HBM for ``FinetuneTrainer.set_batch_device``."""
        f = self._frames(frames)
        t = np.require(np.asarray(tokens, dtype=np.int32).reshape(-1, self.cfg.ctx), requirements="C")
        check(lib.arp_clip_encode_image_multiscale_dev(self._h, _ffi.as_ptr(f, C.c_uint8), f.shape[0], f.shape[1], f.shape[2], bufs[0].ptr, bufs[1].ptr))
        check(lib.arp_clip_encode_text_multiscale_dev(self._h, _ffi.as_ptr(t, C.c_int32), t.shape[0], bufs[2].ptr, bufs[3].ptr))

    def encode_text_multiscale(self, tokens):
        """Per-block EOT-token features [n, txt_layers*txt_width] and the un-normalised text feature (:151-166)."""
        t = np.require(np.asarray(tokens, dtype=np.int32).reshape(-1, self.cfg.ctx), requirements="C")
        inter = np.empty((t.shape[0], self.cfg.txt_layers * self.cfg.txt_width), np.float32)
        fin = np.empty((t.shape[0], self.cfg.embed), np.float32)
        check(lib.arp_clip_encode_text_multiscale(self._h, _ffi.as_ptr(t, C.c_int32), t.shape[0], _ffi.as_ptr(inter, C.c_float),
                                                  _ffi.as_ptr(fin, C.c_float)))
        return inter, fin

    def label_device_async(self, frames_dev, n, h, w, rewards_dev, use_crop=False):
        check(lib.arp_clip_label_dev_async(self._h, frames_dev.ptr, n, h, w, int(bool(use_crop)), rewards_dev.ptr))

    def sync(self):
        check(lib.arp_clip_sync(self._h))

    def set_streams(self, n_streams):
        check(lib.arp_clip_set_streams(self._h, int(n_streams)))

    def record(self, event):
        check(lib.arp_clip_event_record(self._h, event.ptr))

    # -- profiling -------------------------------------------------------------------------------
    def clock_probe(self, on=True):
        """c_fc on the clock-diagnostic instance of the GEMM kernel (``arp_clip_clock_probe``); rewards unchanged."""
        check(lib.arp_clip_clock_probe(self._h, int(on)))

    def clock_read(self):
        """``{"clock_ghz", "workgroups", "workgroup_us"}`` accumulated since the probe was switched on."""
        out = (C.c_double * 3)()
        check(lib.arp_clip_clock_read(self._h, out))
        return {"clock_ghz": out[0], "workgroups": int(out[1]), "workgroup_us": out[2]}

    def profile(self, on=True):
        check(lib.arp_clip_profile_enable(self._h, int(on)))

    def profile_reset(self):
        check(lib.arp_clip_profile_reset(self._h))

    def profile_read(self):
        buf = C.create_string_buffer(1 << 16)
        check(lib.arp_clip_profile_json(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())


def preprocess(frames, use_crop=False, res=224):
    """The reference's torchvision/PIL transform on the GPU: uint8 NHWC -> float32 NCHW."""
    _ffi.require_gpu()
    f = ClipLabeller._frames(frames)
    out = np.empty((f.shape[0], 3, res, res), np.float32)
    check(lib.arp_preprocess(_ffi.as_ptr(f, C.c_uint8), f.shape[0], f.shape[1], f.shape[2], int(bool(use_crop)), res,
                             _ffi.as_ptr(out, C.c_float)))
    return out


def bicubic_coeffs(in_size, out_size, ksize_cap=16):
    """Host-only: the library's Pillow-exact resample table (no GPU needed)."""
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    w = np.zeros((out_size, ksize_cap), np.int32)
    check(lib.arp_bicubic_coeffs(in_size, out_size, _ffi.as_ptr(xmin, C.c_int32), _ffi.as_ptr(cnt, C.c_int32),
                                 _ffi.as_ptr(w, C.c_int32), ksize_cap))
    return xmin, cnt, w
