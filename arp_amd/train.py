"""Path (2): the ARP-DT policy train step on MI355X, behind the reference's call surface.

Mirrors /root/reference/arp_dt/main_procgen.py:
  * ``create_train_step(model, learning_rate, weight_decay) -> train_step_fn`` (:104-141)
  * ``train_step_fn(state, batch, rng) -> (new_state, aux, next_rng)`` with the reference's ``aux`` keys
    (``loss, acc, trans_loss, return_loss, weight_penalty, weight_l2, train_state_step, learning_rate``)
  * ``create_val_step(model) -> val_step_fn(state, batch, rng) -> (aux, next_rng)`` (:144-169)
  * ``prefetch_to_device(iterator, 2, ...)`` (:703): batch i+1 uploads while step i runs
  * ``sync_state_fn`` (:94-101) = :meth:`PolicyTrainer.broadcast_state`
and ARPDT.__call__ (arp_dt/ARPDT.py:152-236) = :meth:`PolicyTrainer.forward`.

``model`` is the policy configuration (the reference passes an ``ARPDT`` flax module built from
``FLAGS.model``; jax/flax do not exist here, so a :class:`PolicyConfig` or a dict of the same fields
stands in).  ``state`` is a :class:`TrainState` wrapping the device-resident parameters, Adam moments and
step counter; like the reference's donated state it is consumed by the call and the returned one must be
used.  ``batch`` keeps the reference's keys: ``batch["image"]`` holds, per image key, the frozen encoder
output ``[B, T, tokens, dim]`` (the boundary of this round: BASELINE.json configs[3] feeds pre-computed
encodings), ``batch["action"]`` int ``[B, T]``, ``batch["rtg"]`` per key ``[B, T, 1]``.
All compute is in libarp_hip.so; there is no CPU fallback.
"""
import ctypes as C
import json
import os
from dataclasses import asdict, dataclass

import numpy as np

from . import _ffi
from ._ffi import MODE_BF16, MODE_F16, MODE_F32, check, lib

AUX_KEYS = ("loss", "acc", "trans_loss", "return_loss", "weight_penalty", "weight_l2", "train_state_step", "learning_rate")


@dataclass(frozen=True)
class PolicyConfig:
    """ARPDT.get_default_config fields that shape the shipped policy (arp_dt/ARPDT.py:27-66)."""
    emb: int = 128
    depth: int = 2
    heads: int = 8
    mlp_ratio: int = 4
    n_actions: int = 15
    window: int = 4
    enc_tokens: int = 257
    enc_dim: int = 768
    use_adapter: bool = True
    lambda_ret: float = 1.0
    use_symlog: bool = False  # config.use_symlog (ARPDT.py:55): symlog of every return-to-go view before their mean
    alibi_bias: bool = False  # config.alibi_bias (ARPDT.py:88, layers.py:74-78): slope_h * key index on the attention scores; off as shipped
    weight_decay: float = 5e-5
    clip_norm: float = 10.0
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8


class PolicyTrainer:
    """Owns one GPU's copy of the policy: parameters, Adam state, activations, RCCL communicator."""

    def __init__(self, cfg, mode="f16", device=0, adapter_corrections=None):
        """mode: GEMM operand type of the adapter path -- "f16" (default: the 16-bit mode that meets north_star's 1e-3 on the
        logits), "bf16" (8 significand bits: ~1e-2) or "f32" (f32-input MFMA, the parity mode).
        adapter_corrections (f16 only): the adapter's forward products with their operand roundings corrected on the fp4 MFMA
        (arp_dt_set_adapter_corrections) -- what the encoder-inside step (row N1, encoder mode "f16c") pairs with."""
        _ffi.require_gpu()
        self.cfg = cfg
        c = _ffi.DtCfg(cfg.emb, cfg.depth, cfg.heads, cfg.mlp_ratio, cfg.n_actions, cfg.window, cfg.enc_tokens, cfg.enc_dim,
                       int(cfg.use_adapter), {"bf16": MODE_BF16, "f16": MODE_F16, "f32": MODE_F32}[mode], device, 1, 0, cfg.lambda_ret,
                       cfg.weight_decay, cfg.clip_norm, cfg.b1, cfg.b2, cfg.eps, int(getattr(cfg, "alibi_bias", False)))
        h = C.c_void_p()
        check(lib.arp_dt_create(C.byref(c), C.byref(h)))
        self._h = h
        # None: the library's default -- ON in f16 mode where the corrected products exist (enc_dim a multiple of 256, >= 512), since round 6
        if adapter_corrections is not None:
            check(lib.arp_dt_set_adapter_corrections(h, int(bool(adapter_corrections))))
        self.world, self.rank = 1, 0
        self.shapes = {}
        n = C.c_int32()
        tot = C.c_int64()
        check(lib.arp_dt_num_params(h, C.byref(tot), C.byref(n)))
        self.num_params = tot.value
        buf = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd = C.c_int32()
        for i in range(n.value):
            check(lib.arp_dt_param_info(h, i, buf, 256, shape, C.byref(nd)))
            self.shapes[buf.value.decode()] = tuple(shape[d] for d in range(nd.value))

    def close(self):
        if getattr(self, "_h", None):
            lib.arp_dt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- tensors (Flax tree path -> array, Flax layout) ---------------------------------------------
    def set_tensors(self, tree, which=0):
        for name, val in tree.items():
            a = np.require(np.asarray(val, dtype=np.float32), requirements="C")
            if tuple(a.shape) != self.shapes[name]:
                raise ValueError(f"{name}: shape {a.shape}, expected {self.shapes[name]}")
            check(lib.arp_dt_set_tensor(self._h, name.encode(), which, _ffi.as_ptr(a, C.c_float)))

    def get_tensors(self, which=0, names=None):
        out = {}
        for name in (names or self.shapes):
            a = np.empty(self.shapes[name], np.float32)
            check(lib.arp_dt_get_tensor(self._h, name.encode(), which, _ffi.as_ptr(a, C.c_float)))
            out[name] = a
        return out

    set_params = set_tensors

    def get_params(self):
        return self.get_tensors(0)

    def get_grads(self):
        return self.get_tensors(1)

    @property
    def step(self):
        s = C.c_int64()
        check(lib.arp_dt_get_step(self._h, C.byref(s)))
        return s.value

    @step.setter
    def step(self, v):
        check(lib.arp_dt_set_step(self._h, int(v)))

    # -- compute --------------------------------------------------------------------------------------
    def set_batch(self, enc, action, rtg):
        enc = np.require(np.asarray(enc, dtype=np.float32), requirements="C")
        action = np.require(np.asarray(action, dtype=np.int32), requirements="C")
        rtg = np.require(np.asarray(rtg, dtype=np.float32), requirements="C")
        B, T = action.shape
        if enc.shape != (B, T, self.cfg.enc_tokens, self.cfg.enc_dim) or rtg.size != B * T or T != self.cfg.window:
            raise ValueError(f"batch shapes: enc {enc.shape}, action {action.shape}, rtg {rtg.shape}")
        check(lib.arp_dt_set_batch(self._h, _ffi.as_ptr(enc, C.c_float), _ffi.as_ptr(action, C.c_int32), _ffi.as_ptr(rtg, C.c_float), B))
        self._B = B

    # -- two device-resident batch slots: the reference's prefetch_to_device(..., 2) (main_procgen.py:703) ----------------------
    def upload_async(self, slot, enc, action, rtg, images=False):
        """Enqueue the host -> device copy of a batch into slot 0 / 1 on the handle's copy stream.  Safe to call from another
        thread while :meth:`train_step` runs on the other slot.  The arrays must stay alive until :meth:`select` (kept here)."""
        enc = np.require(np.asarray(enc, dtype=np.float32), requirements="C")
        action = np.require(np.asarray(action, dtype=np.int32), requirements="C")
        rtg = np.require(np.asarray(rtg, dtype=np.float32), requirements="C")
        B, T = action.shape
        if rtg.size != B * T or T != self.cfg.window or (not images and enc.shape != (B, T, self.cfg.enc_tokens, self.cfg.enc_dim)):
            raise ValueError(f"batch shapes: enc {enc.shape}, action {action.shape}, rtg {rtg.shape}")
        fn = lib.arp_dt_upload_batch_images_async if images else lib.arp_dt_upload_batch_async
        check(fn(self._h, int(slot), _ffi.as_ptr(enc, C.c_float), _ffi.as_ptr(action, C.c_int32), _ffi.as_ptr(rtg, C.c_float), B))
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[int(slot)] = (enc, action, rtg, B)

    def encode_ahead(self, slot):
        """Enqueue the frozen encoder's pass for the frames in batch slot ``slot`` now, on the encoder's own stream, so that it runs beside the current step's
        policy part (``arp_dt_encode_ahead``); the step that reads the slot then waits for it instead of encoding at its head."""
        check(lib.arp_dt_encode_ahead(self._h, int(slot)))

    def select(self, slot):
        """The next forward / val_step / train_step reads batch slot ``slot`` (ordered behind its upload on the GPU)."""
        check(lib.arp_dt_select_batch(self._h, int(slot)))
        self._B = self._inflight[int(slot)][3]

    def val_step(self):
        """val_step_fn's aux (main_procgen.py:144-169): forward only, rank mean of loss / trans_loss / return_loss / acc*100."""
        m = np.empty(4, np.float32)
        check(lib.arp_dt_val_step(self._h, _ffi.as_ptr(m, C.c_float)))
        return {"loss": float(m[0]), "trans_loss": float(m[1]), "return_loss": float(m[2]), "acc": float(m[3])}

    def attach_encoder(self, encoder):
        """Put the frozen M3AE encoder (arp_amd.m3ae.M3AEEncoder) inside the step: the boundary then is the
        reference's own -- batch["image"] holds frames, not encodings (ARPDT.py:413-462)."""
        check(lib.arp_dt_attach_encoder(self._h, encoder._h))
        self._encoder = encoder

    def set_batch_images(self, images, action, rtg):
        images = np.require(np.asarray(images, dtype=np.float32), requirements="C")
        action = np.require(np.asarray(action, dtype=np.int32), requirements="C")
        rtg = np.require(np.asarray(rtg, dtype=np.float32), requirements="C")
        B, T = action.shape
        if images.shape[:2] != (B, T) or rtg.size != B * T or T != self.cfg.window:
            raise ValueError(f"batch shapes: images {images.shape}, action {action.shape}, rtg {rtg.shape}")
        check(lib.arp_dt_set_batch_images(self._h, _ffi.as_ptr(images, C.c_float), _ffi.as_ptr(action, C.c_int32), _ffi.as_ptr(rtg, C.c_float), B))
        self._B = B

    def forward(self):
        """ARPDT.__call__ (ARPDT.py:152-236) on the staged batch."""
        B, T, NA = self._B, self.cfg.window, self.cfg.n_actions
        logits = np.empty((B, T, NA), np.float32)
        ret = np.empty((B, T, 1), np.float32)
        m = np.empty(4, np.float32)
        check(lib.arp_dt_forward(self._h, _ffi.as_ptr(logits, C.c_float), _ffi.as_ptr(ret, C.c_float), _ffi.as_ptr(m, C.c_float)))
        return {"action_pred": logits, "return_pred": ret, "loss": float(m[0]), "acc": float(m[1]), "trans_loss": float(m[2]),
                "return_loss": float(m[3])}

    def backward(self):
        check(lib.arp_dt_backward(self._h))

    def _set_batch_any(self, enc, action, rtg):
        """encodings [B, T, tokens, dim], or -- with a frozen encoder attached -- normalised frames [B, T, H, W, 3] (row N1: what the rollout loop has)"""
        if getattr(self, "_encoder", None) is not None and np.ndim(enc) == 5 and np.shape(enc)[-1] == 3:
            self.set_batch_images(enc, action, rtg)
        else:
            self.set_batch(enc, action, rtg)

    def greedy_action(self, enc, action, rtg):
        """ARPDT.greedy_action (ARPDT.py:488-492): argmax of the LAST time step's action logits."""
        self._set_batch_any(enc, action, rtg)
        return self.forward()["action_pred"][:, -1, :].argmax(-1)

    def greedy_return(self, enc, action, rtg):
        """ARPDT.greedy_return (ARPDT.py:494-495): symexp(return_pred) (utils.py symexp = sign(x)(exp|x| - 1))."""
        self._set_batch_any(enc, action, rtg)
        r = self.forward()["return_pred"]
        return np.sign(r) * (np.exp(np.abs(r)) - 1.0)

    def train_step(self, lr):
        aux = np.empty(9, np.float32)
        check(lib.arp_dt_train_step(self._h, float(lr), _ffi.as_ptr(aux, C.c_float)))
        d = {k: float(aux[i]) for i, k in enumerate(AUX_KEYS)}
        d["train_state_step"] = int(aux[6])
        d["grad_norm"] = float(aux[8])
        return d

    def train_step_async(self, lr):
        check(lib.arp_dt_train_step_async(self._h, float(lr)))

    def sync(self):
        check(lib.arp_dt_sync(self._h))

    def record(self, event):
        check(lib.arp_dt_event_record(self._h, event.ptr))

    # -- data parallelism (one process per GPU, RCCL over xGMI) ---------------------------------------
    @staticmethod
    def new_unique_id():
        buf = C.create_string_buffer(128)
        check(lib.arp_dt_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, world, rank):
        check(lib.arp_dt_comm_init(self._h, C.create_string_buffer(unique_id, 128), world, rank))
        self.world, self.rank = world, rank

    def broadcast_state(self):
        """sync_state_fn (main_procgen.py:94-101)."""
        check(lib.arp_dt_broadcast_state(self._h))

    def comm_info(self):
        """what the RCCL communicator says about itself: nranks (ncclCommCount), rank (ncclCommUserRank), device, RCCL version code"""
        v = (C.c_int32 * 5)()
        check(lib.arp_dt_comm_info(self._h, v))
        return {"nranks": v[0], "rank": v[1], "device": v[2], "rccl_version": v[3], "has_comm": bool(v[4])}

    def comm_selfcheck(self):
        """all-reduce(sum) of ``rank + 1`` through the step's communicator: world (world + 1) / 2 on every rank"""
        d = C.c_double()
        check(lib.arp_dt_comm_selfcheck(self._h, C.byref(d)))
        return d.value

    def profile(self, on=True):
        check(lib.arp_dt_profile_enable(self._h, int(on)))

    def profile_reset(self):
        check(lib.arp_dt_profile_reset(self._h))

    def profile_read(self):
        buf = C.create_string_buffer(1 << 16)
        check(lib.arp_dt_profile_json(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())


class TrainState:
    """The reference's flax ``TrainState`` (params + optax state + step), resident on the GPU."""

    def __init__(self, trainer):
        self.trainer = trainer
        self._live = True

    @classmethod
    def create(cls, model, params, mode="f16", device=0, adapter_corrections=None):
        cfg = model if isinstance(model, PolicyConfig) else PolicyConfig(**dict(model))
        tr = PolicyTrainer(cfg, mode=mode, device=device, adapter_corrections=adapter_corrections)
        tr.set_params(params)
        return cls(tr)

    @property
    def step(self):
        return self.trainer.step

    @property
    def params(self):
        return self.trainer.get_params()


def symlog(x):
    """``sign(x) * log(1 + |x|)`` (arp_dt/utils.py:445-446)."""
    x = np.asarray(x, np.float32)
    return np.sign(x) * np.log1p(np.abs(x))


def _batch_arrays(batch, use_symlog=False):
    """(encodings, actions, return-to-go) of a reference batch dict.  ``batch["rtg"]`` holds one array per image view; the model
    embeds -- and regresses onto -- their MEAN, each view passed through symlog first when ``config.use_symlog``
    (arp_dt/ARPDT.py:251-258,281-293).  Both uses see the same array, so the transform is applied here, once, on the host."""
    image = batch["image"]
    enc = next(iter(image.values())) if isinstance(image, dict) else image
    rtg = batch["rtg"]
    views = [np.asarray(v, np.float32) for v in rtg.values()] if isinstance(rtg, dict) else [np.asarray(rtg, np.float32)]
    if use_symlog:
        views = [symlog(v) for v in views]
    rtg = np.mean(np.stack(views), axis=0) if isinstance(rtg, dict) else views[0]
    return np.asarray(enc), np.asarray(batch["action"]), np.asarray(rtg)


def shard_batch(batch, rank, world, device_axis=False):
    """This rank's part of a global batch: ``generate_batch``'s ``x.reshape(n_devices, -1, *x.shape[1:])[rank]``
    (main_procgen.py:645-683) -- contiguous, equal parts of the leading axis, every leaf of the (nested) batch dict alike;
    ``None`` leaves stay ``None``.  ``device_axis=True``: the batch already carries the reference's leading ``[n_devices, ...]``
    axis (what its pmapped train_step_fn receives) and the rank's slice is ``x[rank]``."""
    if world == 1 and not device_axis:
        return batch

    def cut(x):
        x = np.asarray(x)
        if device_axis:
            if x.shape[0] != world:
                raise ValueError(f"leading device axis is {x.shape[0]}, expected {world}")
            return x[rank]
        if x.shape[0] % world:
            raise ValueError(f"global batch of {x.shape[0]} does not divide over {world} ranks")
        per = x.shape[0] // world
        return x[rank * per : (rank + 1) * per]

    def walk(v):
        if v is None:
            return None
        if isinstance(v, dict):
            return {k: walk(x) for k, x in v.items()}
        return cut(v)

    return walk(batch)


def torch_object_broadcast(dist):
    """``bcast`` for :class:`DataParallel` on an initialised ``torch.distributed`` group (gloo control plane)."""

    def bcast(obj, src=0):
        box = [obj]
        dist.broadcast_object_list(box, src=src)
        return box[0]

    return bcast


class DataParallel:
    """One process per GPU around a :class:`PolicyTrainer` -- the reference's ``pmap`` over devices (main_procgen.py:94-141):

    * the RCCL unique id is made on rank 0 and handed to every rank over the control plane (``bcast``);
    * ``sync_state_fn`` (``:94-101``): every rank takes rank 0's parameters, Adam moments and step counter, once, at start;
    * a step = this rank's shard of the global batch (:func:`shard_batch`) through forward/backward, ONE all-reduce(sum) of
      the flat gradient + one of the loss scalars inside the library (``pmean``, ``:132``; the 1/world is folded into the
      update), then the identical clip + Adam update on every rank.  No other collective."""

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
        return certify_collective(self.trainer, self.rank, self.world)

    def set_global_batch(self, batch, device_axis=False):
        self.trainer.set_batch(*_batch_arrays(shard_batch(batch, self.rank, self.world, device_axis), self.trainer.cfg.use_symlog))

    def train_step(self, batch, lr, device_axis=False):
        """aux is the rank-averaged aux of the reference's pmean: identical on every rank."""
        self.set_global_batch(batch, device_axis)
        return self.trainer.train_step(lr)


def certify_collective(trainer, rank, world):
    """What a multi-GPU bench line prints about its communicator: the rank count and user rank RCCL itself reports
    (``ncclCommCount`` / ``ncclCommUserRank``, not the environment's), and one all-reduce(sum) of ``rank + 1`` through it -- every rank
    must read world (world + 1) / 2.  ``ok`` is False when any of the three disagrees with (rank, world)."""
    info = trainer.comm_info()
    got = trainer.comm_selfcheck()
    want = world * (world + 1) / 2.0
    ok = info["nranks"] == world and info["rank"] == rank and got == want and (world == 1 or info["has_comm"])
    return dict(info, allreduce_selfcheck=got, expected=want, ok=bool(ok),
                NCCL_ALGO=os.environ.get("NCCL_ALGO"), NCCL_PROTO=os.environ.get("NCCL_PROTO"))


def bucket_plan(cfg):
    """Flat-gradient ranges ``[(lo, hi)] * 4`` of the data-parallel step's two all-reduce buckets (bucket 1 = ranges 0, 1:
    image_text_input's kernel + everything the transformer owns, launched while the adapter's backward still runs; bucket 2 =
    ranges 2, 3) and the flat parameter count.  Needs no GPU."""
    c = _ffi.DtCfg(cfg.emb, cfg.depth, cfg.heads, cfg.mlp_ratio, cfg.n_actions, cfg.window, cfg.enc_tokens, cfg.enc_dim, int(cfg.use_adapter),
                   MODE_F16, 0, 1, 0, cfg.lambda_ret, cfg.weight_decay, cfg.clip_norm, cfg.b1, cfg.b2, cfg.eps)
    r = (C.c_int64 * 8)()
    tot = C.c_int64()
    check(lib.arp_dt_bucket_plan(C.byref(c), r, C.byref(tot)))
    return [(r[2 * i], r[2 * i + 1]) for i in range(4)], tot.value


def _threefry2x32(key, x0, x1):
    """Threefry-2x32, 20 rounds (Salmon et al. 2011) on uint32 arrays: the block cipher behind jax.random's default PRNG."""
    rot = ((13, 15, 26, 6), (17, 29, 16, 24))
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    ks = (k0, k1, np.uint32(k0 ^ k1 ^ np.uint32(0x1BD11BDA)))
    x0 = (x0 + ks[0]).astype(np.uint32)
    x1 = (x1 + ks[1]).astype(np.uint32)
    for i in range(5):
        for r in rot[i % 2]:
            x0 = (x0 + x1).astype(np.uint32)
            x1 = ((x1 << np.uint32(r)) | (x1 >> np.uint32(32 - r))).astype(np.uint32)
            x1 = x1 ^ x0
        x0 = (x0 + ks[(i + 1) % 3]).astype(np.uint32)
        x1 = (x1 + ks[(i + 2) % 3] + np.uint32(i + 1)).astype(np.uint32)
    return x0, x1


def split_rng(rng):
    """``next_rng, split_rng = jax.random.split(rng)`` (main_procgen.py:130,163) for a raw threefry key ``uint32[2]`` (or a stack
    ``[..., 2]`` of them, the pmapped ``sharded_rng``), computed as jax's original (non-partitionable) split does: counts
    ``0..3`` enciphered pairwise under the key.  Anything else (None, an int seed, an opaque object) is carried through: the
    shipped configuration has dropout 0, so the step never draws from it."""
    a = np.asarray(rng) if isinstance(rng, (np.ndarray, list, tuple)) else None
    if a is None or a.dtype.kind not in "ui" or a.ndim < 1 or a.shape[-1] != 2:
        return rng, rng
    a = a.astype(np.uint32)
    if a.ndim > 1:
        parts = [split_rng(k) for k in a.reshape(-1, 2)]
        return (np.stack([p[0] for p in parts]).reshape(a.shape), np.stack([p[1] for p in parts]).reshape(a.shape))
    with np.errstate(over="ignore"):
        y0, y1 = _threefry2x32(a, np.array([0, 1], np.uint32), np.array([2, 3], np.uint32))
    out = np.concatenate([y0, y1]).reshape(2, 2)
    return out[0], out[1]


class DeviceBatch:
    """A batch resident in (or on its way into) one of the trainer's two device slots: what :func:`prefetch_to_device` yields
    and what ``train_step_fn`` / ``val_step_fn`` accept where the reference passes a device-put batch."""

    def __init__(self, trainer, slot, release):
        self.trainer, self.slot, self._release = trainer, slot, release

    def done(self):
        """The step that read this batch has been synchronised (its aux was read back): the slot may be refilled."""
        if self._release is not None:
            self._release()
            self._release = None


_SLOT_OWNER_LOCK = __import__("threading").Lock()


def prefetch_to_device(iterator, size, trainer, *, rank=0, world=1, device_axis=False):
    """``flax.jax_utils.prefetch_to_device(iterator, size, devices)`` as the reference uses it (main_procgen.py:703,706: size 2):
    host batches (the reference's batch dicts) are uploaded into the trainer's two device slots by a background thread -- the
    ctypes call releases the GIL, the copy runs on the handle's copy stream -- while the main thread is inside ``train_step_fn``
    on the other slot.  Yields :class:`DeviceBatch`.  ``size`` > 2 is clamped: the library keeps two slots.

    The two device slots belong to the TRAINER, and one prefetcher at a time owns them.  The reference opens ``train_iter`` and
    ``val_iter`` side by side on the same state (main_procgen.py:703-708); here the prefetcher that starts first owns the slots and a
    second one opened while it is live yields prepared HOST batches instead -- ``train_step_fn`` / ``val_step_fn`` stage those through the
    handle's third, synchronous slot -- so a validation upload can never land in a slot that holds an uploaded-but-unconsumed training
    batch, and neither iterator can starve the other of slots."""
    import queue
    import threading
    import warnings

    # Ownership is decided HERE, when the prefetcher is created -- not at the first next() (this used to be a generator function: whichever
    # iterator was pulled first won, so a validation iterator pulled before the training one silently demoted the training loop).
    with _SLOT_OWNER_LOCK:
        owner = getattr(trainer, "_prefetch_owner", None) is None
        if owner:
            token = object()
            trainer._prefetch_owner = token
    if not owner:
        warnings.warn("prefetch_to_device: another prefetcher owns this trainer's two device slots; this one yields host batches (staged "
                      "synchronously by the step functions)", RuntimeWarning, stacklevel=2)
        return (batch for batch in iterator)  # no device slots for this one: the host-side preparation only

    free = queue.Queue()
    for s in range(min(max(int(size), 1), 2)):
        free.put(s)
    ready = queue.Queue()
    stop = threading.Event()

    def worker():
        try:
            for batch in iterator:
                slot = free.get()
                if stop.is_set():
                    return
                enc, act, rtg = _batch_arrays(shard_batch(batch, rank, world, device_axis), trainer.cfg.use_symlog)
                images = getattr(trainer, "_encoder", None) is not None and enc.ndim == 5 and enc.shape[-1] == 3
                trainer.upload_async(slot, enc, act, rtg, images=images)
                if images and os.environ.get("ARP_DT_ENCODE_AHEAD", "1") != "0" and os.environ.get("ARP_DT_ENC_EAGER", "1") != "0":
                    trainer.encode_ahead(slot)  # the frozen encoder's pass for this batch runs beside the step that is reading the OTHER slot
                ready.put(DeviceBatch(trainer, slot, lambda s=slot: free.put(s)))
            ready.put(None)
        except BaseException as e:  # surfaces in the consumer
            ready.put(e)

    th = threading.Thread(target=worker, daemon=True)

    def release():
        stop.set()
        free.put(0)  # wake a worker blocked on a free slot
        if th.is_alive():
            th.join(timeout=30)  # an upload in flight finishes before the slots change hands
        if th.is_alive():
            # the worker is still inside an upload: the slots are NOT free -- keep the ownership (a new prefetcher is demoted to host staging) and say so
            raise RuntimeError("prefetch_to_device: the upload thread did not stop within 30 s; the device slots stay owned by this prefetcher")
        with _SLOT_OWNER_LOCK:
            if getattr(trainer, "_prefetch_owner", None) is token:
                trainer._prefetch_owner = None

    def gen():
        th.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            release()

    g = gen()
    # a prefetcher that is created and dropped without ever being pulled must give the slots back too
    import weakref
    weakref.finalize(g, lambda: (None if th.is_alive() else _release_if_unstarted(trainer, token)))
    return g


def _release_if_unstarted(trainer, token):
    with _SLOT_OWNER_LOCK:
        if getattr(trainer, "_prefetch_owner", None) is token:
            trainer._prefetch_owner = None


def _stage(tr, batch, rank, world, device_axis):
    """Put ``batch`` (a host batch dict, or a DeviceBatch from prefetch_to_device) in front of the next step."""
    if isinstance(batch, DeviceBatch):
        tr.select(batch.slot)
        return batch
    enc, act, rtg = _batch_arrays(shard_batch(batch, rank, world, device_axis), tr.cfg.use_symlog)
    tr._set_batch_any(enc, act, rtg) if hasattr(tr, "_set_batch_any") else tr.set_batch(enc, act, rtg)  # frames in with a frozen encoder attached (row N1): set_batch_images
    return None


def create_val_step(model, *, rank=0, world=1, device_axis=False):
    """main_procgen.py:144-169: ``val_step_fn(state, batch, rng) -> (aux, next_rng)`` -- forward only (``deterministic=True``),
    aux = the rank mean of ``loss, trans_loss, return_loss, acc * 100``; the state is NOT consumed (no ``donate_argnums``)."""

    def val_step_fn(state, batch, rng):
        tr = state.trainer
        if world > 1 and getattr(tr, "world", 1) != world:
            raise ValueError("world > 1: wrap the state's trainer in DataParallel (RCCL communicator + state sync) first")
        dev = _stage(tr, batch, rank, world, device_axis)
        aux = tr.val_step()
        if dev is not None:
            dev.done()
        return aux, split_rng(rng)[0]

    return val_step_fn


def create_train_step(model, learning_rate, weight_decay, *, rank=0, world=1, device_axis=False):
    """main_procgen.py:104-141.  ``learning_rate`` is the schedule ``step -> lr`` (``:135``).  With ``world > 1`` (one process
    per GPU, the state's trainer wrapped by :class:`DataParallel` beforehand) every call steps on this rank's shard of the
    global batch it is given."""
    cfg = model if isinstance(model, PolicyConfig) else PolicyConfig(**dict(model))
    if abs(cfg.weight_decay - weight_decay) > 1e-12:
        cfg = PolicyConfig(**{**asdict(cfg), "weight_decay": float(weight_decay)})

    def train_step_fn(state, batch, rng):
        if not state._live:
            raise RuntimeError("this TrainState was donated to a previous train_step_fn call (donate_argnums=0)")
        tr = state.trainer
        if abs(tr.cfg.weight_decay - cfg.weight_decay) > 1e-12:
            raise ValueError("state was created with a different weight_decay than create_train_step")
        if world > 1 and getattr(tr, "world", 1) != world:
            raise ValueError("world > 1: wrap the state's trainer in DataParallel (RCCL communicator + state sync) first")
        dev = _stage(tr, batch, rank, world, device_axis)  # a host batch dict (synchronous upload) or a prefetched DeviceBatch
        aux = tr.train_step(learning_rate(tr.step))
        if dev is not None:
            dev.done()
        state._live = False
        # next_rng of jax.random.split(rng) (main_procgen.py:130) when rng is a raw threefry key; dropout is 0 in the shipped
        # configuration, so the step itself never draws from the split key
        return TrainState(tr), aux, split_rng(rng)[0]

    return train_step_fn
