"""Oracle (test infrastructure): the host-side plumbing of the reference's labelling loop.

Follows /root/reference/arp_dt/label_reward.py:
  * trajectory boundaries from ``done[:, -1]`` ............ :80-83  (``time`` fallback :84-87)
  * ``discount_cumsum`` (reverse cumulative sum, gamma=1) .. :247-254
  * ``stack_outputs`` (sliding window of last num_frames
    values, first value left-padded) ...................... :232-245
  * per-trajectory loop, dataset key names ................ :256-289

Deliberately written the slow way the reference writes it (Python loops, deque) so that the
product's vectorised versions are checked against an independent formulation.
Parity status: pinned -- tests/golden/rtg.npz holds the datasets the reference FILE ITSELF wrote (tests/golden/make_golden.py
executes /root/reference/arp_dt/label_reward.py under I/O stand-ins); tests/test_oracle.py checks this module against them.
"""
from collections import deque

import numpy as np


def trajectory_bounds(done_last):
    """label_reward.py:80-83: ``list(np.nonzero(done[:, -1])[0] + 1)`` with 0 inserted in front."""
    idx = list(np.nonzero(np.asarray(done_last))[0] + 1)
    idx.insert(0, 0)
    return idx


def discount_cumsum(x, gamma=1.0):
    x = np.asarray(x)
    if len(x.shape) == 0:
        x = x[None, ...]
    out = np.zeros_like(x)
    out[-1] = x[-1]
    for t in reversed(range(x.shape[0] - 1)):
        out[t] = x[t] + gamma * out[t + 1]
    return out


def stack_outputs(pos_outputs, num_frames):
    pos_outputs = np.asarray(pos_outputs)
    if len(pos_outputs.shape) == 0:
        pos_outputs = pos_outputs[None, ...]
    stacked = []
    stack = deque([], maxlen=num_frames)
    for i in range(len(pos_outputs)):
        if i == 0:
            stack.extend([pos_outputs[i]] * num_frames)
        else:
            stack.append(pos_outputs[i])
        stacked.append(list(stack))
    return np.asarray(stacked)


def label_file(store, compute_reward, image_keys="ob", model_type="clip", inst_type="none"):
    """The reference loop (label_reward.py:256-289) over an in-memory ``store`` (dict of numpy
    arrays standing in for the HDF5 file).  ``compute_reward(images_u8[N,H,W,3]) -> float32[N]``.
    Returns {dataset_key: float32 [len_data, num_frames]}."""
    try:
        done = store["done"]
        len_data, num_frames = done.shape[:2]
        bounds = trajectory_bounds(done[:, -1])
    except Exception:  # label_reward.py:84-87
        len_data, num_frames = store["time"].shape[:2]
        bounds = list(np.where(store["time"][:, -1, 0] == 1.0)[0])
        bounds.append(len(store["time"]))
    target_keys = [f"{model_type}_reward", f"{model_type}_pos_rtg"]
    if inst_type != "none":
        target_keys = [f"{k}_{inst_type}" for k in target_keys]
    out = {}
    for img_key in image_keys.split(", "):
        chunks = {k: [] for k in target_keys}
        for idx in range(len(bounds) - 1):
            traj = list(range(bounds[idx], min(bounds[idx + 1], len_data)))
            if not traj:
                continue
            r = compute_reward(store[img_key][traj, -1])
            rtg = discount_cumsum(r)
            chunks[target_keys[0]].append(stack_outputs(r, num_frames))
            chunks[target_keys[1]].append(stack_outputs(rtg, num_frames))
        for k in target_keys:
            out[f"{img_key}_{k}"] = np.concatenate(chunks[k], axis=0).astype(np.float32)
    return out
