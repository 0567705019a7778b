"""Oracle (test infrastructure): the reference's frame preprocessing, restated in numpy.

Follows /root/reference/arp_dt/label_reward.py:109-121 (default transform) and :92-102
(``use_crop`` transform):

    ToPILImage -> Resize(224, BICUBIC) -> CenterCrop(224) -> convert("RGB") -> ToTensor
               -> Normalize(mean, std)

torchvision's ``Resize`` on a PIL image is ``PIL.Image.resize`` (antialiased two-pass resample with
22-bit fixed-point coefficients and a uint8 round/clamp after each pass).  The recipe below is the
one in SURVEY.md Appendix A; ``tests/test_oracle.py`` pins it bit-exactly against
Pillow itself.  Parity status: pinned against PIL (the library the reference calls), not against a
reference-held fixture -- the reference has none.
"""
import numpy as np

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)  # label_reward.py:117
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)

PRECISION_BITS = 32 - 8 - 2  # Pillow's fixed-point coefficient precision (22 bits)


def _cubic(t, a=-0.5):
    t = abs(t)
    if t < 1.0:
        return ((a + 2.0) * t - (a + 3.0)) * t * t + 1.0
    if t < 2.0:
        return (((t - 5.0) * t + 8.0) * t - 4.0) * a
    return 0.0


def bicubic_coeffs(in_size, out_size):
    """Per-output-sample tap window and 22-bit fixed-point weights (Pillow ``precompute_coeffs``).

    Returns (xmin[out], count[out], W[out, ksize] int32).  ``W`` rows are zero-padded past
    ``count``.
    """
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 2.0 * fs
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32)
    cnt = np.zeros(out_size, np.int32)
    W = np.zeros((out_size, ksize), np.int32)
    for o in range(out_size):
        center = (o + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.array([_cubic((k + lo - center + 0.5) / fs) for k in range(n)], dtype=np.float64)
        w = w / w.sum()
        q = np.where(w < 0, np.trunc(w * (1 << PRECISION_BITS) - 0.5), np.trunc(w * (1 << PRECISION_BITS) + 0.5))
        xmin[o] = lo
        cnt[o] = n
        W[o, :n] = q.astype(np.int32)
    return xmin, cnt, W


def _resample_axis_last(img, xmin, cnt, W):
    """img uint8 [..., in]; returns uint8 [..., out]."""
    out_size = xmin.shape[0]
    out = np.empty(img.shape[:-1] + (out_size,), np.uint8)
    src = img.astype(np.int64)
    for o in range(out_size):
        acc = np.full(img.shape[:-1], 1 << (PRECISION_BITS - 1), np.int64)
        for k in range(cnt[o]):
            acc += src[..., xmin[o] + k] * int(W[o, k])
        out[..., o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bicubic_u8(frames, out_h, out_w):
    """Pillow-exact ``Image.resize((out_w, out_h), BICUBIC)`` on uint8 NHWC frames.

    Horizontal pass first (uint8 intermediate), then vertical -- SURVEY.md Appendix A.
    """
    frames = np.asarray(frames)
    assert frames.dtype == np.uint8 and frames.ndim == 4
    n, h, w, c = frames.shape
    x = frames
    if w != out_w:
        xm, ct, W = bicubic_coeffs(w, out_w)
        x = np.moveaxis(_resample_axis_last(np.moveaxis(x, 2, -1), xm, ct, W), -1, 2)
    if h != out_h:
        xm, ct, W = bicubic_coeffs(h, out_h)
        x = np.moveaxis(_resample_axis_last(np.moveaxis(x, 1, -1), xm, ct, W), -1, 1)
    return np.ascontiguousarray(x)


def center_crop(frames, ch, cw):
    """torchvision ``CenterCrop`` on an image at least as large as the crop."""
    n, h, w, c = frames.shape
    top = int(round((h - ch) / 2.0))
    left = int(round((w - cw) / 2.0))
    return frames[:, top : top + ch, left : left + cw, :]


def preprocess_u8(frames, use_crop=False, n_px=224):
    """uint8 NHWC [n,H,W,3] -> uint8 NHWC [n,224,224,3] (the PIL image just before ToTensor)."""
    frames = np.asarray(frames)
    n, h, w, c = frames.shape
    assert c == 3
    if use_crop:
        # label_reward.py:92-102 -- CenterCrop(image_size // 2) then Resize(224).  image_size is
        # g[image_key].shape[-2], i.e. the frame width (label_reward.py:104).
        crop = w // 2
        frames = center_crop(frames, crop, crop)
        n, h, w, c = frames.shape
    # Resize(n_px) scales the SHORTER side to n_px keeping the aspect ratio (torchvision
    # semantics: the long side is int(n_px * long / short)), then CenterCrop(n_px).
    if h <= w:
        oh, ow = n_px, int(n_px * w / h)
    else:
        oh, ow = int(n_px * h / w), n_px
    x = resize_bicubic_u8(frames, oh, ow)
    if not use_crop:
        x = center_crop(x, n_px, n_px)
    elif (oh, ow) != (n_px, n_px):
        # the use_crop transform has no CenterCrop after the resize; square input only.
        raise ValueError("use_crop path expects square frames")
    return np.ascontiguousarray(x)


def preprocess(frames, use_crop=False, n_px=224):
    """Full reference transform: uint8 NHWC -> float32 NCHW [n,3,224,224], normalised."""
    u8 = preprocess_u8(frames, use_crop=use_crop, n_px=n_px)
    x = u8.astype(np.float32) / np.float32(255.0)  # ToTensor
    x = (x - CLIP_MEAN) / CLIP_STD  # Normalize
    return np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))
