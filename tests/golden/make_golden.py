"""Generates the committed golden fixtures (run in the build container; never on the GPU box).

    python tests/golden/make_golden.py

What pins what:
  * preprocess.npz   -- outputs of PIL 12.2 itself (the library the reference calls through
                        torchvision, label_reward.py:109-121) on seeded frames.
  * clip_tiny.npz /
    clip_b32.npz /
    clip_b16.npz     -- outputs of HuggingFace ``CLIPModel`` (quick_gelu), an independent
                        implementation of openai/CLIP, on seeded weights (regenerated from the seed,
                        never stored) and seeded frames / tokens.
  * rtg.npz          -- outputs of the REFERENCE FILE ITSELF, /root/reference/arp_dt/label_reward.py,
                        executed here with its I/O dependencies replaced by in-memory stand-ins
                        (h5py -> dict of numpy arrays, clip -> a callable returning supplied logits,
                        torchvision transforms -> identity).  The stand-ins supply no arithmetic that
                        is under test: what is captured is the reference's own trajectory splitting,
                        discount_cumsum, stack_outputs and dataset naming (:80-87, :232-289).
The reference holds no fixtures of its own, so these are the pins.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def make_preprocess():
    from PIL import Image
    from arp_amd import synth
    frames = np.concatenate([synth.noise_frames(1, seed=7), synth.procgen_like_frames(1, seed=8)])
    small = synth.procgen_like_frames(1, 48, 64, seed=9)
    out = {"frames": frames, "small": small}
    out["resized"] = np.stack([np.asarray(Image.fromarray(f).resize((224, 224), Image.BICUBIC)) for f in frames])
    out["cropped"] = np.stack([np.asarray(Image.fromarray(f).crop((64, 64, 192, 192)).resize((224, 224), Image.BICUBIC)) for f in frames])
    # non-square: Resize(224) keeps aspect (48x64 -> 224x298), CenterCrop(224)
    im = Image.fromarray(small[0]).resize((int(224 * 64 / 48), 224), Image.BICUBIC)
    left = int(round((im.size[0] - 224) / 2.0))
    out["small_resized"] = np.asarray(im.crop((left, 0, left + 224, 224)))[None]
    np.savez_compressed(os.path.join(HERE, "preprocess.npz"), **out)


def make_clip(name, cfg_kw, n, seed):
    from arp_amd import synth
    from hf_clip import build_hf_clip, hf_rewards
    from oracle import clip_np, preprocess
    cfg = clip_np.ClipConfig(**cfg_kw)
    W = synth.clip_weights(cfg, seed=seed)
    frames = synth.procgen_like_frames(n, seed=seed + 1)
    tokens = synth.prompt_tokens(2, [7, 3], ctx=cfg.ctx, vocab=cfg.vocab, seed=seed + 2)
    model = build_hf_clip(W, cfg, eos_id=int(tokens.max()))
    rewards, img, txt = hf_rewards(model, preprocess.preprocess(frames), tokens)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), cfg=np.array(sorted(cfg_kw.items()), dtype=object), seed=seed,
                        frames=frames, tokens=tokens, rewards=rewards.astype(np.float64), image_embeds=img, text_embeds=txt)


class _DS:
    """h5py.Dataset stand-in over a numpy array."""

    def __init__(self, arr):
        self.a = np.array(arr)

    @property
    def shape(self):
        return self.a.shape

    def __getitem__(self, k):
        return self.a[k]

    def __setitem__(self, k, v):
        self.a[k] = v

    def __len__(self):
        return len(self.a)

    def resize(self, n, axis=0):
        pad = np.zeros((n - self.a.shape[0],) + self.a.shape[1:], self.a.dtype)
        self.a = np.concatenate([self.a, pad], axis=0)


class _File(dict):
    def get(self, k):
        return dict.get(self, k)

    def create_dataset(self, key, data=None, **kw):
        self[key] = _DS(data)

    def close(self):
        pass


def make_rtg():
    """Run the reference's label_reward() under I/O stand-ins and capture its datasets."""
    store = {}
    fake_h5py = types.ModuleType("h5py")
    fake_h5py.File = lambda path, mode="r": store["file"]
    fake_clip = types.ModuleType("clip")

    class _Model:
        def __call__(self, images, text):
            # "logits_per_text": row 0 = the supplied per-frame rewards, smuggled in through the image tensor
            import torch
            return None, images.reshape(images.shape[0], -1)[:, 0][None, :].float()

    fake_clip.load = lambda name, device=None: (_Model(), None)
    fake_clip.tokenize = lambda texts: __import__("torch").zeros((len(texts), 77), dtype=__import__("torch").long)
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")

    class _Compose:
        def __init__(self, ts):
            pass

        def __call__(self, img):  # img: the frame; its [0,0,0] byte carries the reward index
            return np.asarray(img, dtype=np.float32)

    for n_ in ("CenterCrop", "Normalize", "Resize", "ToPILImage", "ToTensor"):
        setattr(tvt, n_, lambda *a, **k: None)
    tvt.Compose = _Compose
    tvt.InterpolationMode = types.SimpleNamespace(BICUBIC=3)
    tv.transforms = tvt
    dp = types.ModuleType("arp_dt.data_procgen")
    dp.get_clip_instruct = lambda t: "x"
    dp.get_clip_special_instruct = lambda e, t: "x"
    pkg = types.ModuleType("arp_dt")
    pkg.__path__ = ["/root/reference/arp_dt"]
    sys.modules.update({"h5py": fake_h5py, "clip": fake_clip, "torchvision": tv, "torchvision.transforms": tvt,
                        "arp_dt": pkg, "arp_dt.data_procgen": dp})
    import importlib.util
    spec = importlib.util.spec_from_file_location("arp_dt.label_reward", "/root/reference/arp_dt/label_reward.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    rng = np.random.default_rng(123)
    out = {}
    # case "t": the `time` fallback of label_reward.py:84-87 -- `done` exists but is 1-D, so unpacking its shape[:2] raises and
    # the boundaries come from time[:, -1, 0] == 1.0 (first step of every trajectory)
    for case, lens in (("a", [1, 3, 9, 17]), ("b", [8, 8]), ("c", [5]), ("t", [4, 1, 6])):
        nf = 8
        L = sum(lens)
        rewards = rng.standard_normal(L).astype(np.float32) * 3
        # frames: float "images" whose first element is the reward the fake model returns
        ob = np.zeros((L, nf, 2, 2, 3), np.float32)
        ob[:, -1, 0, 0, 0] = rewards
        done = np.zeros((L, nf), np.float32)
        ends = np.cumsum(lens) - 1
        done[ends, -1] = 1
        f = _File(ob=_DS(ob), done=_DS(done))
        if case == "t":
            tm = np.zeros((L, nf, 1), np.float32)
            for s0, n in zip(np.cumsum([0] + lens[:-1]), lens):
                tm[s0 : s0 + n, -1, 0] = np.arange(1, n + 1)
            f = _File(ob=_DS(ob), done=_DS(done[:, -1].copy()), time=_DS(tm))
            out["t_time"] = tm
        store["file"] = f
        ref.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path="mem.hdf5", image_keys="ob", num_frames=nf,
                         env_type="none", model_type="clip", use_crop=False, inst_type="none")
        keys = sorted(k for k in f if k not in ("ob", "done", "time"))
        out[f"{case}_rewards"] = rewards
        out[f"{case}_done"] = done
        out[f"{case}_keys"] = np.array(keys)
        for k in keys:
            out[f"{case}__{k}"] = np.asarray(f[k].a)
    np.savez_compressed(os.path.join(HERE, "rtg.npz"), **out)
    for m in ("h5py", "clip", "torchvision", "torchvision.transforms", "arp_dt", "arp_dt.data_procgen"):
        sys.modules.pop(m, None)


if __name__ == "__main__":
    from conftest import TINY
    make_preprocess()
    make_rtg()
    make_clip("clip_tiny", TINY, 3, seed=11)
    make_clip("clip_b32", dict(patch=32), 2, seed=0)
    make_clip("clip_b16", dict(patch=16), 1, seed=0)
    print("goldens written to", HERE)
