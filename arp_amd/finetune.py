"""SURVEY row N2 -- the CLIP multi-scale adapter fine-tune step on the MI355X (BASELINE.json configs[4]).

Host-side mirror of ``finetune_module/clip_multiscale_adapter.py::CLIPMultiscaleAdapter`` (its trainable head) and of the
optimiser step in ``finetune_module/finetune.py:69-93,139-141``, over the C ABI ``arp_ft_*`` (include/arp_hip.h).  The
frozen CLIP towers stay outside this step, as the reference freezes them (``finetune.py:139-140``): their per-block CLS /
EOT features (what the reference's forward hooks collect, ``utils.py:6-18``) and final features are the batch.  Parameter
names and layouts are torch's own ``state_dict`` entries, so a reference checkpoint's non-``clip_model.*`` tensors load
as they are.
"""
import ctypes as C
import json
from dataclasses import asdict, dataclass

import numpy as np

from . import _ffi
from ._ffi import check, lib

MODES = {"f32": _ffi.MODE_F32, "bf16": _ffi.MODE_BF16, "f16": _ffi.MODE_F16}
AUX_KEYS = ("loss", "vip_loss", "id_loss", "lambda_id")


@dataclass
class FinetuneConfig:
    layers: int = 12
    width_v: int = 768
    width_t: int = 512
    embed: int = 512
    hidden: int = 1024
    n_actions: int = 15
    gamma: float = 0.98                       # clip_multiscale_adapter.py:107
    logit_scale: float = float(np.log(1 / 0.07))  # clip_model.logit_scale (:95)
    use_vip: bool = True                      # finetune.py:41-42
    use_id: bool = True
    goal_conditioned: bool = False            # clip_multiscale_adapter.py:208-212,224-230: image3 stands where the prompt stands
    weight_decay: float = 0.001               # finetune.py:31
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8

    @property
    def d_img(self):
        return self.layers * self.width_v

    @property
    def d_txt(self):
        return self.layers * self.width_t

    @property
    def feat(self):
        return self.layers * self.width_t + self.embed


def bucket_plan(cfg):
    """Flat-gradient ranges of the data-parallel step's seven all-reduce buckets in the order the backward produces them
    (``[[(lo, hi), (lo, hi)]] * 7``, empty ranges have lo == hi) and the flat parameter count.  Needs no GPU."""
    c = _ffi.FtCfg(cfg.layers, cfg.width_v, cfg.width_t, cfg.embed, cfg.hidden, cfg.n_actions, MODES["f16"], 0, int(cfg.use_vip), int(cfg.use_id),
                   cfg.gamma, cfg.logit_scale, cfg.weight_decay, cfg.b1, cfg.b2, cfg.eps)
    r = (C.c_int64 * 28)()
    tot = C.c_int64()
    check(lib.arp_ft_bucket_plan(C.byref(c), r, C.byref(tot)))
    return [[(r[4 * b], r[4 * b + 1]), (r[4 * b + 2], r[4 * b + 3])] for b in range(7)], tot.value


class FinetuneTrainer:
    """Parameters, AdamW state and the staged batch live on the GPU; one host thread per handle."""

    def __init__(self, cfg, mode="bf16", device=0):
        self.cfg = cfg
        c = _ffi.FtCfg(cfg.layers, cfg.width_v, cfg.width_t, cfg.embed, cfg.hidden, cfg.n_actions, MODES[mode], device, int(cfg.use_vip), int(cfg.use_id),
                       cfg.gamma, cfg.logit_scale, cfg.weight_decay, cfg.b1, cfg.b2, cfg.eps, int(cfg.goal_conditioned))
        h = C.c_void_p()
        check(lib.arp_ft_create(C.byref(c), C.byref(h)))
        self._h = h
        self._B = 0
        total, n = C.c_int64(), C.c_int32()
        check(lib.arp_ft_num_params(self._h, C.byref(total), C.byref(n)))
        self.n_params = total.value
        self.shapes = {}
        for i in range(n.value):
            name = C.create_string_buffer(256)
            shape = (C.c_int64 * 4)()
            nd = C.c_int32()
            check(lib.arp_ft_param_info(self._h, i, name, 256, shape, C.byref(nd)))
            self.shapes[name.value.decode()] = tuple(int(shape[d]) for d in range(nd.value))

    def close(self):
        if self._h:
            check(lib.arp_ft_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- state ----------------------------------------------------------------------------------------
    def set_tensors(self, tensors, which=0):
        missing = set(self.shapes) - set(tensors)
        if missing and which == 0:
            raise KeyError(f"missing parameters: {sorted(missing)}")
        for name, v in tensors.items():
            if name.startswith("clip_model."):
                continue  # frozen towers: not part of this step
            a = np.require(np.asarray(v, dtype=np.float32), requirements="C")
            if tuple(a.shape) != self.shapes[name]:
                raise ValueError(f"{name}: shape {a.shape}, expected {self.shapes[name]}")
            check(lib.arp_ft_set_tensor(self._h, name.encode(), which, _ffi.as_ptr(a, C.c_float)))

    def get_tensors(self, which=0):
        out = {}
        for name, shp in self.shapes.items():
            a = np.empty(shp, np.float32)
            check(lib.arp_ft_get_tensor(self._h, name.encode(), which, _ffi.as_ptr(a, C.c_float)))
            out[name] = a
        return out

    set_params = set_tensors
    load_state_dict = set_tensors

    def get_params(self):
        return self.get_tensors(0)

    state_dict = get_params

    def get_grads(self):
        return self.get_tensors(1)

    @property
    def step(self):
        s = C.c_int64()
        check(lib.arp_ft_get_step(self._h, C.byref(s)))
        return s.value

    @step.setter
    def step(self, v):
        check(lib.arp_ft_set_step(self._h, int(v)))

    @property
    def dropped_gradients(self):
        """f16 mode: gradient elements that reached AdamW as inf / NaN (binary16 overflow in the scaled backward) and were treated as missing,
        cumulative.  Always 0 in the other modes, where nothing is masked (a non-finite gradient shows up as NaN parameters, as in torch)."""
        n = C.c_uint64()
        check(lib.arp_ft_dropped_gradients(self._h, C.byref(n)))
        return n.value

    # -- compute --------------------------------------------------------------------------------------
    def set_batch(self, img_inter, img_final, txt_inter, txt_final, r, action):
        """img_inter [3,B,layers*width_v], img_final [3,B,embed] (image0..2), txt_inter [B,layers*width_t], txt_final [B,embed],
        r [B] or [B,1] as stored in the batch, action [B] class ids.  ``goal_conditioned``: four image groups (image0..3), txt_* ignored."""
        c = self.cfg
        f32 = lambda x: np.require(np.asarray(x, dtype=np.float32), requirements="C")
        img_inter, img_final = f32(img_inter), f32(img_final)
        r = f32(np.asarray(r).reshape(-1))
        action = np.require(np.asarray(action, dtype=np.int32).reshape(-1), requirements="C")
        B = action.shape[0]
        G = 4 if c.goal_conditioned else 3  # goal_conditioned: image0..image3, no prompt features (txt_* may be None)
        if img_inter.shape != (G, B, c.d_img) or img_final.shape != (G, B, c.embed) or r.shape != (B,):
            raise ValueError(f"batch shapes: {img_inter.shape} {img_final.shape} {r.shape} {action.shape}")
        p = _ffi.as_ptr
        if c.goal_conditioned:
            ti = tf = None
        else:
            txt_inter, txt_final = f32(txt_inter), f32(txt_final)
            if txt_inter.shape != (B, c.d_txt) or txt_final.shape != (B, c.embed):
                raise ValueError(f"batch shapes: {txt_inter.shape} {txt_final.shape}")
            ti, tf = p(txt_inter, C.c_float), p(txt_final, C.c_float)
        check(lib.arp_ft_set_batch(self._h, p(img_inter, C.c_float), p(img_final, C.c_float), ti, tf, p(r, C.c_float), p(action, C.c_int32), B))
        self._B = B

    def feature_buffers(self, B):
        """Device buffers (img_inter, img_final, txt_inter, txt_final) for ``ClipLabeller.encode_multiscale_to``."""
        from .clip import DeviceBuffer
        c = self.cfg
        return (DeviceBuffer(3 * B * c.d_img * 4), DeviceBuffer(3 * B * c.embed * 4), DeviceBuffer(B * c.d_txt * 4), DeviceBuffer(B * c.embed * 4))

    def set_batch_device(self, bufs, r, action):
        """As set_batch, with the tower features already in HBM (image rows ordered image0 | image1 | image2)."""
        r = np.require(np.asarray(r, dtype=np.float32).reshape(-1), requirements="C")
        action = np.require(np.asarray(action, dtype=np.int32).reshape(-1), requirements="C")
        B = action.shape[0]
        check(lib.arp_ft_set_batch_dev(self._h, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, _ffi.as_ptr(r, C.c_float),
                                       _ffi.as_ptr(action, C.c_int32), B))
        self._B = B

    def forward(self):
        """CLIPMultiscaleAdapter.forward (clip_multiscale_adapter.py:177-250) on the staged batch."""
        B = self._B
        m = np.empty(4, np.float32)
        s = np.empty((3, B), np.float32)
        lg = np.empty((B, self.cfg.n_actions), np.float32)
        check(lib.arp_ft_forward(self._h, _ffi.as_ptr(m, C.c_float), _ffi.as_ptr(s, C.c_float), _ffi.as_ptr(lg, C.c_float)))
        return {"loss": float(m[0]), "vip_loss": float(m[1]), "id_loss": float(m[2]), "lambda_id": float(m[3]), "scores": s, "logits": lg}

    def _encode(self, which, inter, final):
        f32 = lambda x: np.require(np.asarray(x, dtype=np.float32), requirements="C")
        inter, final = f32(inter), f32(final)
        n = inter.shape[0]
        din = self.cfg.d_img if which == 0 else self.cfg.d_txt
        if inter.shape != (n, din) or final.shape != (n, self.cfg.embed):
            raise ValueError(f"feature shapes: {inter.shape} {final.shape}")
        out = np.empty((n, self.cfg.feat), np.float32)
        check(lib.arp_ft_encode(self._h, which, _ffi.as_ptr(inter, C.c_float), _ffi.as_ptr(final, C.c_float), n, _ffi.as_ptr(out, C.c_float)))
        self._B = 0
        return out

    def encode_image(self, inter, final):
        """CLIPMultiscaleAdapter.encode_image after the towers (clip_multiscale_adapter.py:141-149): adapted, normalised."""
        return self._encode(0, inter, final)

    def encode_text(self, inter, final):
        """CLIPMultiscaleAdapter.encode_text after the towers (:164-171) for [n, ctx] prompts."""
        return self._encode(1, inter, final)

    def backward(self):
        check(lib.arp_ft_backward(self._h))

    def train_step(self, lr):
        """loss.backward(); optimizer.step() of finetune.py:81-84; returns the pre-update losses."""
        aux = np.empty(4, np.float32)
        check(lib.arp_ft_train_step(self._h, float(lr), _ffi.as_ptr(aux, C.c_float)))
        return {k: float(aux[i]) for i, k in enumerate(AUX_KEYS)}

    def train_step_async(self, lr):
        check(lib.arp_ft_train_step_async(self._h, float(lr)))

    def sync(self):
        check(lib.arp_ft_sync(self._h))

    # -- data parallelism (BASELINE configs[4]: DP = 8; one process per GPU, RCCL over xGMI) ------------------------------------
    @staticmethod
    def new_unique_id():
        buf = C.create_string_buffer(128)
        check(lib.arp_dt_comm_unique_id(buf))  # one RCCL id serves any handle type
        return buf.raw

    def comm_init(self, unique_id, world, rank):
        check(lib.arp_ft_comm_init(self._h, C.create_string_buffer(unique_id, 128), world, rank))
        self.world, self.rank = world, rank

    def broadcast_state(self):
        """every rank takes rank 0's parameters, AdamW moments and step (= loading one checkpoint everywhere)"""
        check(lib.arp_ft_broadcast_state(self._h))

    def comm_info(self):
        v = (C.c_int32 * 5)()
        check(lib.arp_ft_comm_info(self._h, v))
        return {"nranks": v[0], "rank": v[1], "device": v[2], "rccl_version": v[3], "has_comm": bool(v[4])}

    def comm_selfcheck(self):
        d = C.c_double()
        check(lib.arp_ft_comm_selfcheck(self._h, C.byref(d)))
        return d.value

    def record(self, event):
        check(lib.arp_ft_event_record(self._h, event.ptr))

    def profile(self, on=True):
        check(lib.arp_ft_profile_enable(self._h, int(on)))

    def profile_reset(self):
        check(lib.arp_ft_profile_reset(self._h))

    def profile_read(self):
        buf = C.create_string_buffer(1 << 16)
        check(lib.arp_ft_profile_json(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())


class FinetunedClip:
    """``model = CLIPMultiscaleAdapter(...); model.load_state_dict(ckpt)`` of the ``clip_ft`` labelling branch
    (arp_dt/label_reward.py:166-177): frozen towers (``arp_amd.clip.ClipLabeller``) + trained head on one GPU."""

    def __init__(self, towers, head):
        self.towers, self.head = towers, head
        self._text = None

    @classmethod
    def from_state_dict(cls, state_dict, mode="bf16", device=0, model="ViT-B/16", logit_scale=None):
        """state_dict: the reference checkpoint's tensors as numpy (``clip_model.*`` = the CLIP weights, the rest = the head)."""
        from . import clip as aclip
        sd = {k: np.asarray(v) for k, v in state_dict.items()}
        cw = {k[len("clip_model."):]: v for k, v in sd.items() if k.startswith("clip_model.")}
        ccfg = aclip.MODELS[model] if isinstance(model, str) else model
        towers = aclip.ClipLabeller(ccfg, cw, mode=mode, device=device)  # "f16" applies to the towers only
        # (the head has its own f16 mode since round 2: the same operand type end to end)
        ls = float(cw["logit_scale"]) if logit_scale is None else float(logit_scale)  # model.logit_scale = clip's, detached (:95)
        hidden = sd["inverse_layer.layers.0.weight"].shape[0]
        head = FinetuneTrainer(FinetuneConfig(layers=ccfg.layers, width_v=ccfg.width, width_t=ccfg.txt_width, embed=ccfg.embed, hidden=hidden,
                                              n_actions=sd["inverse_layer.layers.3.weight"].shape[0], logit_scale=ls), mode=mode, device=device)
        head.load_state_dict({k: v for k, v in sd.items() if not k.startswith("clip_model.")})
        return cls(towers, head)

    def set_text(self, tokens):
        self._text = self.head.encode_text(*self.towers.encode_text_multiscale(tokens))
        return self

    def label(self, images, use_crop=False):
        """compute_reward of the clip_ft branch (label_reward.py:197-228): exp(logit_scale) * <adapted image, adapted prompt 0>."""
        images = np.asarray(images)
        if use_crop:  # center_crop(images, (image_size // 2,) * 2), image_size = the frame WIDTH (label_reward.py:15-36,104,203)
            h, w = images.shape[1:3]
            cs = w // 2
            sh, sw = int((h - cs) / 2), int((w - cs) / 2)
            images = images[:, sh: sh + cs, sw: sw + cs]
        a = self.head.encode_image(*self.towers.encode_image_multiscale(np.ascontiguousarray(images)))
        return (np.exp(self.head.cfg.logit_scale) * (a @ self._text[0])).astype(np.float32)

    # -- the rollout loop's online rewards with the fine-tuned model (arp_dt/envs/vl_reward.py:44-79) --------------------------------
    def _online_features(self, frame, use_crop):
        """``model.encode_image(preprocess(Image.fromarray(obs)))``: the LABEL transform (Pillow bicubic), not the fine-tune one, in front of
        towers + head; ``use_crop`` = the numpy centre crop to half the frame height applied first (vl_reward.py:46-47)."""
        f = np.asarray(frame)
        if f.ndim == 3:
            f = f[None]
        return self.head.encode_image(*self.towers.encode_image_multiscale(np.ascontiguousarray(f), pil=True, use_crop=use_crop))

    def online_reward(self, obs, mean_over_prompts=False, use_crop=False):
        """get_torch_clip_adapter_reward (vl_reward.py:44-61): exp(logit_scale) * <adapted image, adapted prompt p>, prompt 0 -- or the mean over
        the cached prompts where the reference's ``pos_text`` is a list.  float32 [1]."""
        a = self._online_features(obs, use_crop)
        logit = np.exp(self.head.cfg.logit_scale) * (self._text @ a[0])  # [n_prompts]
        return np.asarray([logit.mean() if mean_over_prompts else logit[0]], np.float32)

    def online_goal_reward(self, obs, goal_image, use_crop=False):
        """get_torch_clip_adapter_goal_conditioned_reward (vl_reward.py:64-79): -||a(obs) - a(goal)||_2 on the ADAPTED (normalised) features."""
        obs, goal_image = np.asarray(obs), np.asarray(goal_image)
        if use_crop:  # the reference crops obs first and then sizes the goal's crop from the CROPPED obs (a quarter of the frame): kept
            from .label_reward import center_crop
            h = obs.shape[0] // 2
            obs = center_crop(obs[None], (h, h))[0]
            goal_image = center_crop(goal_image[None], (h // 2, h // 2))[0]
        if obs.shape == goal_image.shape:
            a = self._online_features(np.stack([obs, goal_image]), False)
        else:
            a = np.concatenate([self._online_features(obs, False), self._online_features(goal_image, False)])
        return -1.0 * float(np.linalg.norm(a[0].astype(np.float64) - a[1].astype(np.float64)))

    def close(self):
        self.head.close()
        self.towers.close()


def flops_per_sample(cfg):
    """Algorithmic FLOPs of one sample's forward + backward through the head (2 per MAC; weight gradients included)."""
    F, Hd = cfg.feat, cfg.hidden * (cfg.layers + 1)
    img = cfg.d_txt * cfg.d_img + 2 * F * Hd      # per image row, forward MACs
    txt = cfg.d_txt * cfg.d_txt + 2 * F * Hd
    inv = 4 * F * cfg.hidden + cfg.hidden * cfg.n_actions
    # backward: dW for every layer, dX for every layer but the two bias-free input projections
    fwd = 3 * img + txt + inv
    bwd = 3 * (img + 2 * F * Hd) + (txt + 2 * F * Hd) + 2 * inv
    return 2.0 * (fwd + bwd)


def synth_params(cfg, seed=0):
    """Seeded stand-in for a checkpoint (numpy PCG64): fan-in scaled weights, small biases, the reference's scalar inits."""
    rng = np.random.Generator(np.random.PCG64(seed))
    F, Hd = cfg.feat, cfg.hidden * (cfg.layers + 1)
    shapes = {"image_intermediate_linear.weight": (cfg.d_txt, cfg.d_img), "text_intermediate_linear.weight": (cfg.d_txt, cfg.d_txt)}
    for a in ("image_adapter", "text_adapter"):
        shapes.update({f"{a}.layers.0.weight": (Hd, F), f"{a}.layers.0.bias": (Hd,), f"{a}.layers.3.weight": (F, Hd), f"{a}.layers.3.bias": (F,)})
    shapes.update({"inverse_layer.layers.0.weight": (cfg.hidden, 4 * F), "inverse_layer.layers.0.bias": (cfg.hidden,),
                   "inverse_layer.layers.3.weight": (cfg.n_actions, cfg.hidden), "inverse_layer.layers.3.bias": (cfg.n_actions,)})
    P = {}
    for k, shp in shapes.items():
        if k.endswith(".bias"):
            P[k] = (0.02 * rng.standard_normal(shp, dtype=np.float32)).astype(np.float32)
        else:
            P[k] = (rng.standard_normal(shp, dtype=np.float32) / np.float32(np.sqrt(shp[1]))).astype(np.float32)
    P["image_residual_weight"] = np.float32(4.0) * np.ones((), np.float32)   # clip_multiscale_adapter.py:91-92
    P["text_residual_weight"] = np.float32(4.0) * np.ones((), np.float32)
    P["lambda_id"] = np.float32(np.log(1 / 0.07)) * np.ones((), np.float32)   # :103
    return P


def synth_batch(cfg, B, seed=0):
    """Random stand-ins for the frozen towers' outputs (O(1) features, as LayerNormed residual streams are)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.standard_normal((3, B, cfg.d_img), dtype=np.float32), rng.standard_normal((3, B, cfg.embed), dtype=np.float32),
            rng.standard_normal((B, cfg.d_txt), dtype=np.float32), rng.standard_normal((B, cfg.embed), dtype=np.float32),
            rng.integers(0, 2, (B,)).astype(np.float32), rng.integers(0, cfg.n_actions, (B,)).astype(np.int32))


def shard_batch(batch, rank, world):
    """This rank's contiguous part of a global fine-tune batch ``(img_inter [3,B,..], img_final [3,B,..], txt_inter [B,..],
    txt_final [B,..], r [B], action [B])``: samples [rank*B/world, (rank+1)*B/world) of every array.  The VIP term couples the
    samples of a batch through its [B,B] matrix (clip_multiscale_adapter.py:177-250), so data parallelism -- as for any
    in-batch contrastive loss -- optimises the mean of the per-shard losses, not the loss of the undivided batch."""
    if world == 1:
        return batch
    B = np.asarray(batch[5]).shape[0]
    if B % world:
        raise ValueError(f"global batch of {B} does not divide over {world} ranks")
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    return tuple(np.asarray(a)[:, sl] if i < 2 else np.asarray(a)[sl] for i, a in enumerate(batch))


class DataParallel:
    """One process per GPU around a :class:`FinetuneTrainer`: rank 0's RCCL id over the control plane, state broadcast, then
    every step = this rank's shard, ONE all-reduce(sum) of the flat gradient inside the library, identical AdamW update on all
    ranks (the scheme of arp_amd.train.DataParallel; bcast as there, e.g. ``train.torch_object_broadcast(dist)``)."""

    def __init__(self, trainer, rank, world, bcast):
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world {world}")
        self.trainer, self.rank, self.world = trainer, int(rank), int(world)
        uid = trainer.new_unique_id() if rank == 0 else None
        uid = bcast(uid, 0)
        if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
            raise ValueError("the control plane did not deliver rank 0's 128-byte RCCL unique id")
        trainer.comm_init(bytes(uid), self.world, self.rank)
        trainer.broadcast_state()

    def certify(self):
        from .train import certify_collective
        return certify_collective(self.trainer, self.rank, self.world)

    def train_step(self, batch, lr):
        self.trainer.set_batch(*shard_batch(batch, self.rank, self.world))
        return self.trainer.train_step(lr)
