"""Drop-in mirror of the reference's labelling entry point, with the CLIP forward on MI355X.

Mirrors /root/reference/arp_dt/label_reward.py:
  * ``label_reward(...)`` -- same positional/keyword signature as :44-60 (extra options are
    keyword-only and default to the reference behaviour);
  * the inner seam ``compute_reward(clip_model, images, text=text) -> float32[N]`` (:132-146) is
    :func:`make_compute_reward`;
  * ``discount_cumsum`` (:247-254) and ``stack_outputs`` (:232-245) are vectorised numpy with
    bit-identical results (same f32 summation order);
  * dataset key names ``"{img_key}_{model_type}_reward"`` / ``"..._pos_rtg"`` (+ ``_{inst_type}``),
    gzip, chunks ``(1, num_frames)`` (:256-289).

Differences that are forced by this image (documented in DESIGN.md): no BPE vocabulary offline, so
the prompt is passed as token ids (``tokens=``) or through a ``tokenizer=`` callable
(``clip.tokenize`` when the package exists); no pretrained checkpoint offline, so ``weights=`` (an
openai/CLIP state dict) must be supplied; ``h5py`` is imported lazily and only for ``.hdf5`` paths --
any mapping of numpy arrays (``store=``) works as the data file.

Multi-GPU: labelling shards by contiguous trajectory ranges balanced by frame count, one process per
GPU, no collective on the data path; rank 0 is the only writer.
"""
import os
import time

import numpy as np

from .clip import MODELS, ClipLabeller
from .data import get_clip_instruct, get_clip_special_instruct  # noqa: F401  (re-exported like the reference)


def discount_cumsum(x, gamma=1.0):
    """label_reward.py:247-254.  ``out[t] = x[t] + gamma * out[t+1]`` in x's dtype."""
    x = np.asarray(x)
    if x.ndim == 0:
        x = x[None, ...]
    if gamma == 1.0:
        return np.cumsum(x[::-1], axis=0, dtype=x.dtype)[::-1].copy()
    out = np.zeros_like(x)
    out[-1] = x[-1]
    for t in range(x.shape[0] - 2, -1, -1):
        out[t] = x[t] + gamma * out[t + 1]
    return out


def stack_outputs(pos_outputs, num_frames):
    """label_reward.py:232-245.  Row i = the last ``num_frames`` values up to i, first value left-padded."""
    x = np.asarray(pos_outputs)
    if x.ndim == 0:
        x = x[None, ...]
    idx = np.arange(len(x))[:, None] + np.arange(-num_frames + 1, 1)[None, :]
    return x[np.clip(idx, 0, None)]


def trajectory_bounds(store, done_key=None):
    """label_reward.py:71-87, including the ``time`` fallback of :84-87 when the done-key path raises.
    Returns (len_data, num_frames, [traj start indices..., end])."""
    # label_reward and label_store both ask; on a recorder file the scan is one library read per row of `done` (31 ms for 8192 rows),
    # so the answer is kept on the store object for the lifetime of the handle (nothing here writes `done` / `time`)
    cached = getattr(store, "_arp_bounds", None) if done_key is None else None
    if cached is not None:
        return cached[0], cached[1], list(cached[2])
    auto_key = done_key is None
    if done_key is None:
        for k in ("done", "rewards", "is_terminal"):
            if k in store and store[k] is not None:
                done_key = k
                break
        else:
            raise ValueError
    try:
        d = store[done_key]
        len_data, num_frames = d.shape[:2]
        idx = list(np.nonzero(np.asarray(d[:, -1]))[0] + 1)
        idx.insert(0, 0)
    except Exception:  # label_reward.py:84-87: boundaries from the step counter (1.0 on the first step of a trajectory)
        t = store["time"]
        len_data, num_frames = t.shape[:2]
        idx = list(np.where(np.asarray(t[:, -1, 0]) == 1.0)[0])
        idx.append(len(t))
    if auto_key:
        try:
            store._arp_bounds = (len_data, num_frames, list(idx))
        except (AttributeError, TypeError):  # a plain dict: nothing to hang the cache on
            pass
    return len_data, num_frames, idx


def shard_trajectories(bounds, world):
    """Contiguous trajectory ranges per rank, balanced by frame count.  Returns [(t0, t1)] * world."""
    ntraj = len(bounds) - 1
    lens = np.diff(np.asarray(bounds))
    total = int(lens.sum())
    cuts, acc, r = [0], 0, 1
    for t in range(ntraj):
        acc += int(lens[t])
        while r < world and acc >= total * r / world:
            cuts.append(t + 1)
            r += 1
    while len(cuts) < world:
        cuts.append(ntraj)
    cuts.append(ntraj)
    return [(cuts[i], cuts[i + 1]) for i in range(world)]


def make_compute_reward(model_type="clip"):
    """The reference's ``compute_reward`` closures (label_reward.py:132-146 and :148-163)."""
    if model_type == "clip":

        def compute_reward(clip_model, images, text=None, use_crop=False):
            # the prompt was tokenised + encoded once in clip_model.set_text (the reference re-runs
            # the text tower per trajectory; same value).  Quirk Q1 kept: always prompt 0.
            return clip_model.label(images, use_crop=use_crop)

    elif model_type == "clip_goal_conditioned":

        def compute_reward(clip_model, images, text=None, use_crop=False):
            f = clip_model.encode_image(images, use_crop=use_crop, normalize=False)
            return -1 * np.linalg.norm(f - f[-1], ord=2, axis=1).astype(np.float64)

    elif model_type == "clip_ft":

        def compute_reward(clip_model, images, text=None, use_crop=False):
            # clip_model: arp_amd.finetune.FinetunedClip (towers + fine-tuned head, prompt cached by set_text);
            # label_reward.py:197-228 with a tokenised (tensor) prompt -> logit[0]
            return clip_model.label(images, use_crop=use_crop)

    else:
        raise NotImplementedError(f"model_type {model_type!r} is outside the MI355X hot path (SURVEY.md section 8a, L12)")
    return compute_reward


def center_crop(image, crop_size):
    """label_reward.py:15-36: ``image[:, sh:sh+ch, sw:sw+cw]`` of an [N, H, W, C] stack, ``sh = int((H - ch) / 2)``."""
    _, H, W, _ = image.shape
    ch, cw = crop_size
    sh, sw = int((H - ch) / 2), int((W - cw) / 2)
    return image[:, sh: sh + ch, sw: sw + cw, :]


def _prompt_mode(clip_model, pos_text):
    """The rollout loop's ``isinstance(pos_text, list)`` switch (envs/vl_reward.py:19-22, 56-59): a list of prompts -> the mean over the
    prompts cached by ``set_text`` (their count must match), a single prompt (str / None) -> prompt 0.  A 2-D int array is taken as
    ``clip.tokenize`` output: it is (re)encoded and reduced like a list."""
    if isinstance(pos_text, np.ndarray) and pos_text.dtype.kind in "iu" and pos_text.ndim == 2:
        key = pos_text.tobytes()
        if getattr(clip_model, "_online_tokens", None) != key:
            clip_model.set_text(pos_text)
            clip_model._online_tokens = key
        return pos_text.shape[0] > 1
    if isinstance(pos_text, list):
        n = getattr(clip_model, "_n_prompts", None)
        if n is None and getattr(clip_model, "_text", None) is not None:
            n = len(clip_model._text)
        if n is not None and n != len(pos_text):
            raise ValueError(f"pos_text lists {len(pos_text)} prompts but {n} are cached: call set_text with the tokens of exactly these prompts")
        return True
    return False


def get_torch_clip_reward(clip_model, obs, pos_text=None, use_crop=False):
    """Online single-frame reward of the rollout loop (/root/reference/arp_dt/envs/vl_reward.py:11-23):
    one uint8 frame [H,W,3] (or a stack [N,H,W,3]) -> float32 [N]; the prompt(s) are the ones cached by
    ``clip_model.set_text`` -- ``pos_text`` a list: ``logits_per_text.mean(axis=0)`` over them (:19-20), otherwise prompt 0.
    Same kernels as the offline path; a call of a few frames runs on the latency path."""
    obs = np.asarray(obs)
    if obs.ndim == 3:
        obs = obs[None]
    mean = _prompt_mode(clip_model, pos_text)
    if hasattr(clip_model, "set_prompt_reduce"):
        clip_model.set_prompt_reduce("mean" if mean else "first")
    elif mean:
        raise TypeError("this model object cannot average over prompts")
    try:
        return clip_model.label(obs, use_crop=use_crop)
    finally:
        if mean:
            clip_model.set_prompt_reduce("first")  # the offline pass on the same handle keeps label_reward.py:146 (prompt 0)


def get_torch_clip_goal_conditioned_reward(clip_model, obs, goal_image, use_crop=False):
    """vl_reward.py:26-41: ``-||encode_image(obs) - encode_image(goal)||_2`` on the UN-normalised CLIP image features; a python float.
    ``use_crop``: obs is centre-cropped to half its height, and the goal to half of the CROPPED obs' height -- a quarter of the frame; the
    reference sizes the second crop from the already re-assigned ``obs`` (:29-31) and that is what is reproduced."""
    obs, goal_image = np.asarray(obs), np.asarray(goal_image)
    if use_crop:
        h = obs.shape[0] // 2
        obs = center_crop(obs[None], (h, h))[0]
        goal_image = center_crop(goal_image[None], (h // 2, h // 2))[0]
    if obs.shape == goal_image.shape:  # one call of two frames (the latency path takes up to 1 024 token rows)
        f = clip_model.encode_image(np.stack([obs, goal_image]), use_crop=False, normalize=False)
    else:
        f = np.concatenate([clip_model.encode_image(np.ascontiguousarray(obs[None]), use_crop=False, normalize=False),
                            clip_model.encode_image(np.ascontiguousarray(goal_image[None]), use_crop=False, normalize=False)])
    return -1.0 * float(np.linalg.norm(f[0].astype(np.float64) - f[1].astype(np.float64)))


def get_torch_clip_adapter_reward(clip_model, obs, pos_text=None, use_crop=False):
    """vl_reward.py:44-61 with the fine-tuned model (``arp_amd.finetune.FinetunedClip``, prompt(s) cached by its ``set_text``):
    ``exp(logit_scale) * <adapted image, adapted prompt>``, prompt 0 or the mean over a list of prompts; float32 [1]."""
    return clip_model.online_reward(obs, mean_over_prompts=_prompt_mode(clip_model, pos_text), use_crop=use_crop)


def get_torch_clip_adapter_goal_conditioned_reward(clip_model, obs, goal_image, use_crop=False):
    """vl_reward.py:64-79: ``-||a(obs) - a(goal)||_2`` on the fine-tuned model's adapted, L2-normalised image features; a python float."""
    return clip_model.online_goal_reward(obs, goal_image, use_crop=use_crop)


VL_REWARD_FNS = {  # the dispatch of envs/rollout_procgen.py:133-151 on vl_type
    "clip": get_torch_clip_reward,
    "clip_goal_conditioned": get_torch_clip_goal_conditioned_reward,
    "clip_ft": get_torch_clip_adapter_reward,
    "clip_ft_goal_conditioned": get_torch_clip_adapter_goal_conditioned_reward,
}


def _prefetch(it, depth=4):
    """Runs the iterator in a background thread, ``depth`` items ahead; exceptions surface at the consumer."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    end = object()

    def work():
        try:
            for x in it:
                q.put((x, None))
            q.put((end, None))
        except BaseException as e:  # noqa: BLE001 -- re-raised by the consumer
            q.put((end, e))

    threading.Thread(target=work, daemon=True).start()
    while True:
        x, err = q.get()
        if x is end:
            if err is not None:
                raise err
            return
        yield x


def _open_store(data_path, mode="a"):
    if data_path.endswith((".hdf5", ".h5")):
        from .h5store import H5Store  # ctypes over libhdf5 -- the library h5py wraps; ImportError if absent
        return H5Store(data_path, mode), True
    raise ValueError(f"unsupported data file {data_path!r}: pass store=<mapping of arrays> instead")


def _big_buffer(n_elems, dtype):
    """A frame buffer of ~200 MB that is filled once per batch and dropped at the end of the pass.  ARP_LABEL_HUGEPAGES=1 asks for
    transparent huge pages (anonymous memory + MADV_HUGEPAGE): 49 k page faults going in and an munmap of as many pages coming out cost
    ~9 ms each per buffer with 4 KiB pages (six buffers: 50 ms of a 220 ms file pass).  OFF by default -- measured on the MI355X host
    it is a loss: 0.81 s instead of 0.21 s for the 8192-row pass (2 MiB pages zeroed inside the faults of 32 inflate threads, with
    the host's `defrag = madvise` compaction in the way); on an idle 8-core box the same switch took first touch from 885 to 130 ms."""
    nbytes = int(n_elems) * np.dtype(dtype).itemsize
    try:
        import mmap
        if nbytes >= (8 << 20) and hasattr(mmap, "MADV_HUGEPAGE") and os.environ.get("ARP_LABEL_HUGEPAGES", "0") == "1":
            mm = mmap.mmap(-1, nbytes + (2 << 20))
            mm.madvise(mmap.MADV_HUGEPAGE)
            base = np.frombuffer(mm, np.uint8)
            off = (-base.ctypes.data) % (2 << 20)  # start on a 2 MiB boundary: every huge page of the range is whole
            return base[off : off + nbytes].view(dtype)
    except (OSError, ValueError, ImportError):
        pass
    return np.empty(int(n_elems), dtype)


_FRAME_POOL = []  # frame buffers of earlier label_store passes in this process (dropping one costs ~9 ms of munmap; a later pass --
                  # the next image key, the next demonstration file -- takes them back).  release_frame_buffers() empties it.


def release_frame_buffers():
    _FRAME_POOL.clear()


def _take_frame_buffer(n_elems, dtype):
    for i, b in enumerate(_FRAME_POOL):
        if b.dtype == np.dtype(dtype) and b.size >= n_elems:
            return _FRAME_POOL.pop(i)
    return _big_buffer(n_elems, dtype)


class RowSink:
    """Writes label rows into the demonstration file WHILE the GPU labels the next batch (one process, HDF5; ARP_LABEL_STREAM_WRITE=0
    keeps the reference's order: everything labelled, then everything written).  The datasets are created -- gzip, chunks
    (1, num_frames), maxshape (None, num_frames): label_reward.py:277-283 -- or grown to the file's row count on first use; a writer
    thread places the row blocks (h5store's lock serialises it against the reader's chunk-address queries; the inflate threads never
    enter the library).  The file's final content is the reference's: same datasets, same rows, same filters."""

    def __init__(self, store, num_frames, total_rows, dtype):
        import queue
        import threading
        self.store, self.num_frames, self.total_rows, self.dtype = store, num_frames, int(total_rows), dtype
        self.q = queue.Queue()
        self.err = None
        self.rows = 0
        self.created, self.grown = [], {}  # what abort() has to undo: datasets this pass created / grew (key -> former row count)
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _dataset(self, key):
        ds = self.store.get(key) if hasattr(self.store, "get") else None
        if ds is None:
            ds = self.store.create_dataset(key, shape=(self.total_rows, self.num_frames), dtype=self.dtype, compression="gzip",
                                           chunks=(1, self.num_frames), maxshape=(None, self.num_frames))
            self.created.append(key)
            return ds
        if ds.shape[0] < self.total_rows:
            self.grown[key] = ds.shape[0]
            ds.resize(self.total_rows, axis=0)
        return ds

    def _run(self):
        cache = {}
        t_create = t_write = 0.0
        while True:
            item = self.q.get()
            if item is None:
                if os.environ.get("ARP_LABEL_TIMING") == "1":
                    print(f"[RowSink] dataset create / grow {1e3 * t_create:.1f} ms, row writes {1e3 * t_write:.1f} ms", flush=True)
                return
            if self.err is not None:
                continue  # drain: the error surfaces in close()
            try:
                key, first, rows = item
                t = time.perf_counter()
                if key not in cache:
                    cache[key] = self._dataset(key)
                t1 = time.perf_counter()
                cache[key][first : first + rows.shape[0]] = rows.astype(self.dtype, copy=False)
                t_create += t1 - t
                t_write += time.perf_counter() - t1
            except BaseException as e:  # noqa: BLE001 -- re-raised by close()
                self.err = e

    def __call__(self, key, first, rows):
        self.rows += rows.shape[0]
        self.q.put((key, int(first), rows))

    def close(self):
        self.q.put(None)
        self.t.join()
        if self.err is not None:
            self.abort()
            raise self.err

    def abort(self):
        """The pass failed part-way (a GPU error, a reader exception, a failed write): the reference writes only after everything is
        labelled (label_reward.py:273-289), so a failed run leaves the file as it found it.  Here rows were already streaming in: datasets
        this pass CREATED are deleted and datasets it GREW are cut back to their former length, so that no later reader meets well-formed
        datasets whose tail is fill values.  Rows of a pre-existing dataset that were already overwritten stay overwritten --
        with correct labels of this configuration, which is what the completed pass would have left there."""
        if self.t.is_alive():
            self.q.put(None)
            self.t.join()
        for key in self.created:
            try:
                del self.store[key]
            except Exception:  # noqa: BLE001 -- best effort on the error path; the original error is what the caller sees
                pass
        for key, n in self.grown.items():
            try:
                self.store[key].resize(n, axis=0)
            except Exception:  # noqa: BLE001
                pass
        self.created, self.grown = [], {}


def label_store(store, clip_model, compute_reward, image_keys="ob", model_type="clip", inst_type="none", use_crop=False,
                rank=0, world=1, text=None, batch_frames=1024, sink=None):
    """The per-trajectory loop (label_reward.py:256-289) over any mapping of arrays.

    For the per-frame rewards (``clip``, ``clip_ft``: frame i's reward depends on frame i and the prompt only) consecutive
    trajectories are labelled in ONE ``compute_reward`` call of up to ``batch_frames`` frames and split again afterwards --
    same values, but a 64-frame call is launch-bound (30 k frames/s host-to-host) where a 1024-frame call runs at 70 k.
    ``clip_goal_conditioned`` compares against the trajectory's own last frame and stays one call per trajectory.

    Returns ``{dataset_key: (first_row, float32 [rows, num_frames])}`` for this rank's shard.  ``sink(dataset_key, first_row, rows)``
    (a ``RowSink``): the batched HDF5 path hands every batch's rows over as soon as they exist instead of collecting them; the
    returned row arrays are then empty."""
    len_data, num_frames, bounds = trajectory_bounds(store)
    target_keys = [f"{model_type}_reward", f"{model_type}_pos_rtg"]
    if inst_type != "none":
        target_keys = [f"{k}_{inst_type}" for k in target_keys]
    t0, t1 = shard_trajectories(bounds, world)[rank]
    out = {}
    for img_key in image_keys.split(", "):
        parts = {k: [] for k in target_keys}
        per_frame = model_type in ("clip", "clip_ft") and batch_frames and batch_frames > 0
        pending, pending_frames = [], 0  # [(images, n)] of whole trajectories awaiting one batched call

        def flush():
            nonlocal pending, pending_frames
            if not pending:
                return
            r_all = np.asarray(compute_reward(clip_model, np.concatenate([im for im in pending]) if len(pending) > 1 else pending[0],
                                              text=text, use_crop=use_crop))
            o = 0
            for im in pending:
                r = r_all[o : o + len(im)]
                o += len(im)
                parts[target_keys[0]].append(stack_outputs(r, num_frames))
                parts[target_keys[1]].append(stack_outputs(discount_cumsum(r), num_frames))
            pending, pending_frames = [], 0

        ds = store[img_key]
        spans = [(bounds[idx], min(bounds[idx + 1], len_data)) for idx in range(t0, t1) if bounds[idx] < min(bounds[idx + 1], len_data)]
        if hasattr(ds, "read_last_frames_spans") and per_frame:
            # HDF5: one chunk inflated per num_frames rows on native threads (h5store.py); whole trajectories are grouped into
            # batches of up to batch_frames frames that are inflated straight into one buffer (no concatenate), and the next
            # batch is read while the GPU labels the current one
            # (the FIRST batch is a quarter of the others: the GPU has nothing to do until it has been inflated -- 15 ms for 1024 frames on 32 threads)
            groups, cur, cur_n = [], [], 0
            for a, b in spans:
                cap = batch_frames if groups else max(batch_frames // 4, 1)
                if cur and cur_n + (b - a) > cap:
                    groups.append(cur)
                    cur, cur_n = [], 0
                cur.append((a, b))
                cur_n += b - a
            if cur:
                groups.append(cur)
        if hasattr(ds, "read_last_frames_spans") and per_frame and groups:
            if sink is not None:
                sink.total_rows = spans[-1][1]  # the reference's datasets end with the last LABELLED row (rows after the last `done` are not labelled)
            timing = os.environ.get("ARP_LABEL_TIMING") == "1"
            t_read = t_wait = t_label = 0.0

            import queue
            frame_elems = int(np.prod(ds.shape[2:]))
            free = queue.Queue()  # frame buffers go round: reader fills one, the labeller hands it back (no malloc / munmap per batch)
            # plain CLIP rewards of a labeller with the asynchronous pair (arp_clip_label_submit / _collect): ONE call stays in flight while
            # the next is submitted, so the GPU does not drain between batches (a synchronous call pays the pipeline's fill and drain)
            pipelined = (model_type == "clip" and getattr(clip_model, "label_submit", None) is not None and os.environ.get("ARP_LABEL_PIPELINE", "1") != "0"
                         and max(sum(b - a for a, b in g) for g in groups) <= getattr(clip_model, "max_batch", 0))
            pinned = []
            bufs = []
            for _ in range(min(len(groups) + 1, 5 if pipelined else 4)):  # filling + two read ahead + one in flight + the one being submitted
                buf = _take_frame_buffer(max(sum(b - a for a, b in g) for g in groups) * frame_elems, ds.dtype)
                bufs.append(buf)
                # (ARP_LABEL_PIN=1 pins the buffers -- arp_host_register.  Measured, and therefore OFF by default: registering 200 MB costs
                # ~15 ms per buffer while the staged pageable upload already runs at the link rate, 56.5 against 57.4 GB/s, and overlaps the
                # pass equally well: profiles/r3_seam_probe.txt)
                if getattr(clip_model, "pin_host", None) is not None and os.environ.get("ARP_LABEL_PIN", "0") == "1":
                    try:
                        clip_model.pin_host(buf)
                        pinned.append(buf)
                    except Exception:
                        pass
                free.put(buf)

            def read(g):
                nonlocal t_read
                buf = free.get()
                t = time.perf_counter()
                fr = ds.read_last_frames_spans(g, out=buf)
                t_read += time.perf_counter() - t
                return g, (fr, buf)

            def emit(grp, r_all):
                o = 0
                rs, gs = [], []
                for a, b in grp:
                    r = r_all[o : o + b - a]
                    o += b - a
                    rs.append(stack_outputs(r, num_frames))
                    gs.append(stack_outputs(discount_cumsum(r), num_frames))
                if sink is not None:  # a group is whole consecutive trajectories: one contiguous block of rows from grp[0][0]
                    sink(f"{img_key}_{target_keys[0]}", grp[0][0], np.concatenate(rs, axis=0))
                    sink(f"{img_key}_{target_keys[1]}", grp[0][0], np.concatenate(gs, axis=0))
                else:
                    parts[target_keys[0]].extend(rs)
                    parts[target_keys[1]].extend(gs)

            t_setup = time.perf_counter()
            it = _prefetch((read(g) for g in groups), depth=2)
            inflight, k = None, 0  # (slot, trajectories, buffer) of the submitted, not yet collected call
            while True:
                t = time.perf_counter()
                nxt = next(it, None)
                t_wait += time.perf_counter() - t
                if nxt is None:
                    break
                grp, (frames_all, buf) = nxt
                t = time.perf_counter()
                if pipelined:
                    clip_model.label_submit(k & 1, frames_all, use_crop=use_crop)
                    prev, inflight = inflight, (k & 1, grp, buf)
                    k += 1
                    if prev is not None:
                        emit(prev[1], clip_model.label_collect(prev[0]))
                        free.put(prev[2])
                else:
                    emit(grp, np.array(compute_reward(clip_model, frames_all, text=text, use_crop=use_crop)))
                    free.put(buf)
                t_label += time.perf_counter() - t
                del frames_all
            t_tail = time.perf_counter()
            if inflight is not None:
                emit(inflight[1], clip_model.label_collect(inflight[0]))
                free.put(inflight[2])
            for buf in pinned:
                clip_model.unpin_host(buf)
            del _FRAME_POOL[: max(0, len(_FRAME_POOL) + len(bufs) - 6)]
            _FRAME_POOL.extend(bufs)
            if timing:
                print(f"[label_store] {len(groups)} batches: reader thread busy {t_read:.3f} s, labeller waited for frames {t_wait:.3f} s, "
                      f"labelling {t_label:.3f} s, last collect {time.perf_counter() - t_tail:.3f} s, loop {t_tail - t_setup:.3f} s", flush=True)
            source = ()
        elif hasattr(ds, "read_last_frames"):
            source = _prefetch((ds.read_last_frames(a, b) for a, b in spans), depth=4)
        else:
            source = (np.asarray(ds[a:b, -1]) for a, b in spans)
        for images in source:
            if per_frame:
                if pending and pending_frames + len(images) > batch_frames:
                    flush()
                pending.append(images)
                pending_frames += len(images)
                continue
            r = np.asarray(compute_reward(clip_model, images, text=text, use_crop=use_crop))
            parts[target_keys[0]].append(stack_outputs(r, num_frames))
            parts[target_keys[1]].append(stack_outputs(discount_cumsum(r), num_frames))
        flush()
        first = bounds[t0] if t0 < len(bounds) else len_data
        for k in target_keys:
            # the reference stores what np.concatenate of the stacks gives: float32 for the CLIP logits, float64 for the
            # goal-conditioned distances (label_reward.py:163 casts them to float64).  The dtype follows the MODEL TYPE, not the rows
            # at hand: a rank with an empty shard must not create (or down-cast into) a float32 dataset ahead of float64 rows
            dt = reward_dtype(model_type)
            rows = np.concatenate(parts[k], axis=0).astype(dt, copy=False) if parts[k] else np.zeros((0, num_frames), dt)
            out[f"{img_key}_{k}"] = (first, rows)
    return out


def reward_dtype(model_type):
    """Stored dtype of the label datasets: float64 for the goal-conditioned distances (label_reward.py:163), float32 otherwise."""
    return np.float64 if "goal_conditioned" in str(model_type) else np.float32


def write_results(store, results, is_hdf5, num_frames):
    """label_reward.py:273-289: create (gzip, chunks (1,num_frames)) or overwrite in place."""
    for key, (first, rows) in results.items():
        existing = store.get(key) if hasattr(store, "get") else None
        if existing is not None and getattr(existing, "shape", (0,))[0] >= first + rows.shape[0]:
            existing[first : first + rows.shape[0]] = rows
        elif is_hdf5:
            if existing is None:
                store.create_dataset(key, compression="gzip", chunks=(1, num_frames), maxshape=(None, num_frames), data=rows)
            else:
                existing.resize(first + rows.shape[0], axis=0)
                existing[first:] = rows
        else:
            prev = np.asarray(existing) if existing is not None else np.zeros((0, num_frames), rows.dtype)
            store[key] = np.concatenate([prev[:first], rows], axis=0)


def label_reward(
    env_name,
    distribution_mode,
    num_levels,
    start_level,
    text,
    base_path,
    data_path=None,
    image_keys="ob",
    num_demonstrations=500,
    num_frames=8,
    env_type=None,
    model_type="clip",
    model_ckpt_dir=None,
    use_crop=False,
    inst_type="none",
    *,
    store=None,
    clip_model=None,
    weights=None,
    tokens=None,
    tokenizer=None,
    model_name="ViT-B/16",
    mode="f16",
    device=0,
    rank=0,
    world=1,
    gather=None,
):
    """Same call surface and side effect as the reference (label_reward.py:44-291): labels every
    trajectory of the demonstration file and writes ``{img_key}_{model_type}_reward`` and
    ``..._pos_rtg``.  Returns None.  ``gather(results) -> list of per-rank results`` merges shards
    when ``world > 1`` (rank 0 writes)."""
    # a misconfigured sharded call fails HERE, not after this rank has labelled its whole shard
    if not (0 <= int(rank) < int(world)):
        raise ValueError(f"rank {rank} outside world {world}")
    if world > 1 and gather is None:
        raise ValueError("world > 1 needs gather=<callable returning every rank's results>: rank 0 alone would write a file "
                         "shorter than len_data (the reference always labels the whole file)")
    is_hdf5 = False
    timing = os.environ.get("ARP_LABEL_TIMING") == "1"
    tm = [("start", time.perf_counter())]
    if store is None:
        if data_path is None:
            dirname = f"{env_name}_{distribution_mode}_level{start_level}to{num_levels}_num{num_demonstrations}_frame{num_frames}"
            if env_type != "none":
                dirname += f"_{env_type}"
            data_path = os.path.join(base_path, dirname, "data.hdf5")
        # one process: "a" as the reference (label_reward.py:69).  Sharded (one process per GPU): every rank READS through its own
        # read-only handle -- HDF5 locks a file that is open for writing -- and rank 0 reopens it "a" for the single-writer step
        store, is_hdf5 = _open_store(data_path, "a" if world == 1 else "r")
    tm.append(("open", time.perf_counter()))
    # the scan of `done` (one library read per row of a recorder file: 10 ms per 4 096 rows) runs on a thread of its own beside the prompt's text tower
    # (a GPU call: 4 ms, 14 ms the first time in a process); label_store finds the result cached on the store handle
    import threading
    scan = {}

    def _scan():
        try:
            scan["r"] = trajectory_bounds(store)
        except BaseException as e:  # noqa: BLE001 -- re-raised below
            scan["e"] = e

    scan_thread = threading.Thread(target=_scan, daemon=True)
    scan_thread.start()

    def _bounds():
        scan_thread.join()
        if "e" in scan:
            raise scan["e"]
        return scan["r"]

    compute_reward = make_compute_reward(model_type)
    own_model = clip_model is None
    file_open = is_hdf5
    try:
        if own_model:
            if weights is None:
                raise ValueError("no pretrained CLIP checkpoint is reachable offline: pass weights=<openai/CLIP state dict> "
                                 "or clip_model=<ClipLabeller>")
            if model_type == "clip_ft":  # weights = the fine-tune checkpoint (clip_model.* + head), label_reward.py:166-177
                from .finetune import FinetunedClip
                clip_model = FinetunedClip.from_state_dict(weights, mode=mode, device=device, model=model_name)
            else:
                clip_model = ClipLabeller(MODELS[model_name], weights, mode=mode, device=device)
        if model_type in ("clip", "clip_ft"):
            if tokens is None:
                if tokenizer is None:
                    raise ValueError("no BPE vocabulary offline: pass tokens=<int32 [1,77]> or tokenizer=<callable>")
                tokens = tokenizer([text] if not isinstance(text, list) else text)
            clip_model.set_text(np.asarray(tokens, dtype=np.int32))
        tm.append(("model+text", time.perf_counter()))
        _, num_frames, _ = _bounds()  # quirk Q2: the argument is overwritten from the file
        tm.append(("bounds (rest)", time.perf_counter()))

        # one process on a real file: rows are written while the next batch is labelled (RowSink)
        sink = None
        if is_hdf5 and world == 1 and model_type in ("clip", "clip_ft") and os.environ.get("ARP_LABEL_STREAM_WRITE", "1") != "0":
            sink = RowSink(store, num_frames, 0, reward_dtype(model_type))  # label_store sets the row count once it knows its spans
        try:
            results = label_store(store, clip_model, compute_reward, image_keys=image_keys, model_type=model_type,
                                  inst_type=inst_type, use_crop=use_crop, rank=rank, world=world, text=text, sink=sink)
        except BaseException:
            if sink is not None:
                sink.abort()  # no half-labelled datasets stay behind (the reference writes nothing before the labelling is complete)
            raise
        tm.append(("label_store", time.perf_counter()))
        if sink is not None:
            sink.close()
        tm.append(("sink.close", time.perf_counter()))
        if is_hdf5 and world > 1:
            store.close()  # before the gather: it is the barrier after which no rank holds the file
            file_open = False
        per_rank = gather(results) if world > 1 else [results]
        if rank == 0:
            if is_hdf5 and world > 1:
                store, _ = _open_store(data_path, "a")
                file_open = True
            for res in per_rank:
                write_results(store, res, is_hdf5, num_frames)
        tm.append(("write", time.perf_counter()))
    finally:  # an error on the way (weights, a failed GPU call, the gather) still releases the GPU handle and the file
        scan_thread.join()
        if file_open:
            store.close()
        if timing:
            tm.append(("close", time.perf_counter()))
            print("[label_reward] " + ", ".join(f"{n} {1e3 * (t - tm[i][1]):.1f} ms" for i, (n, t) in enumerate(tm[1:])) + f"; total {1e3 * (tm[-1][1] - tm[0][1]):.1f} ms", flush=True)
        if own_model and clip_model is not None:
            clip_model.close()
    return None
