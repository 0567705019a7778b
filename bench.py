#!/usr/bin/env python3
"""Headline benchmark: frames/s CLIP-reward-labelled (256x256 uint8 frames, ViT-B/32, 16-bit MFMA operands) on N MI355X.

One "step" = one pass of the hot path (preprocess -> ViT-B/32 -> reward) over one batch of
`--batch` (1024) synthetic frames per GPU, frames already resident in HBM (BASELINE.json
configs[1]).  Labelling shards embarrassingly: each rank labels its own frames,
no collective on the data path; `value` = frames labelled by all ranks / max-over-ranks wall time.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 with the contract keys plus `roofline` (dominant kernel, HIP-event
timed per launch) and `cpu_baseline` (torch-CPU port of the same pass, timed on this host).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}  # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md


def gemm_sites(cfg, batch):
    """Algorithmic FLOPs per launch of every GEMM call site of the image tower (2 per MAC)."""
    m, d, g = batch * cfg.tokens, cfg.width, cfg.grid
    return {
        "vit.patch_embed": 2.0 * batch * g * g * d * 3 * cfg.patch * cfg.patch,
        # fused QKV projection + attention (csrc/qkvattn.h): the projection plus QK^T and PV of every (frame, head)
        "vit.qkv_attn": 2.0 * m * 3 * d * d + 4.0 * batch * cfg.tokens * cfg.tokens * d,
        "vit.qkv": 2.0 * m * 3 * d * d,
        "vit.out_proj": 2.0 * m * d * d,
        "vit.c_fc": 2.0 * m * 4 * d * d,
        "vit.c_proj": 2.0 * m * 4 * d * d,
        "vit.proj": 2.0 * batch * d * cfg.embed,
    }


def cpu_baseline(model, seconds, policy_batch=0, finetune_batch=0):
    """Runs the oracle's torch-CPU port in a child process (before this process touches the GPU)."""
    try:
        cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--model", model, "--target-seconds", str(seconds)]
        if policy_batch:
            cmd = [sys.executable, "-m", "oracle.cpu_baseline_policy", "--batch", str(policy_batch), "--target-seconds", str(seconds)]
        if finetune_batch:
            cmd = [sys.executable, "-m", "oracle.cpu_baseline_finetune", "--batch", str(finetune_batch), "--target-seconds", str(seconds)]
        out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        d.pop("checksum", None)
        d.pop("final_loss", None)
        return d
    except Exception as e:  # the baseline is a reported extra, never the measured path
        return {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}


_JSON_FD = None


def claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries print there too (gloo's "[Gloo] Rank 0 is connected to ..." banner at
    rendezvous, ROCm notices): from here on file descriptor 1 points at stderr and the JSON line goes to the saved real stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    sys.stdout.flush()
    os.write(_JSON_FD if _JSON_FD is not None else 1, (line + "\n").encode())


# ---- what the driver reads ---------------------------------------------------------------------------------------------------
# The driver keeps the LAST 8 KB of stdout and parses the LAST line (a 23.6 KB line left BENCH_r04.json.parsed null).
# So: the last stdout line is the headline object ALONE and at most HEADLINE_LIMIT bytes; every secondary bench is its own compact
# line (at most SECONDARY_LIMIT bytes) printed BEFORE it; the complete objects go to gpurun_out/bench_full.json.
HEADLINE_LIMIT = 4096
SECONDARY_LIMIT = 1024
FULL_PATH = os.path.join("gpurun_out", "bench_full.json")


def sig(o, digits=5):
    """floats to `digits` significant figures, recursively (a printed 17-digit double is 2-3x the bytes of the number it carries)"""
    if isinstance(o, bool) or o is None:
        return o
    if isinstance(o, float):
        return float(f"{o:.{digits}g}") if np.isfinite(o) else None
    if isinstance(o, dict):
        return {k: sig(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [sig(v, digits) for v in o]
    return o


def pick(d, keys):
    return None if not isinstance(d, dict) else {k: d[k] for k in keys if k in d and d[k] is not None}


def cut(s, n):
    return s if not isinstance(s, str) or len(s) <= n else s[:n - 1] + "~"


def compact_secondary(name, d):
    """One secondary bench's line in at most SECONDARY_LIMIT bytes: the contract keys, the workload, roofline / cpu_baseline / parity
    numbers.  Prose (sample descriptions, sources, notes) is cut first; the full line is in gpurun_out/bench_full.json."""
    if not isinstance(d, dict) or "value" not in d:
        return {"secondary": name, "error": cut(str((d or {}).get("error", "no JSON line")), 300)}
    out = {"secondary": name}
    out.update(pick(d, ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "dtype"]))
    out["config"] = {"workload": cut((d.get("config") or {}).get("workload", ""), 150)}
    par = (d.get("config") or {}).get("parallelism")
    if par:
        out["config"]["parallelism"] = cut(par, 40)
    rf = pick(d.get("roofline"), ["bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "avg_launch_ms", "kernel"])
    if rf:
        rf["kernel"] = cut(rf.get("kernel", ""), 70)
        out["roofline"] = rf
    cb = pick(d.get("cpu_baseline"), ["value", "unit", "cores", "kind"])
    if cb:
        out["cpu_baseline"] = cb
    par = d.get("parity")
    if isinstance(par, dict):
        err = par.get("max_logit_err_vs_oracle", par.get("max_cosine_err_vs_oracle"))
        out["parity"] = {"err": err, "tolerance": par.get("tolerance"), "within_tolerance": par.get("within_tolerance")}
    ws = d.get("whole_step") or d.get("whole_pass")
    if isinstance(ws, dict):
        out["whole"] = pick(ws, ["gflop_per_step", "tflops", "mfma_frac_of_peak", "nominal_mfma_frac_of_peak"])
    for k in ("rccl", "later_pass_frames_per_s", "cold_process_first_call_frames_per_s", "rows", "file_rewards_equal_direct_labelling", "more_rewards_ms"):
        if k in d:
            out[k] = d[k]
    if isinstance(d.get("staged"), dict):
        out["staged"] = pick(d["staged"], ["staged_ms_per_step", "allreduce_b1_ms", "allreduce_b2_ms", "allreduce_ms", "bucket_bytes"])
        out.pop("rccl", None)
    if name == "online":
        out["reward_ms"] = {k: v.get("latency_ms") for k, v in (d.get("reward") or {}).items()}
        out["greedy_action_ms"] = (d.get("greedy_action") or {}).get("latency_ms")
    sites = d.get("sites_ms_per_step")
    if isinstance(sites, dict):
        out["top_sites_ms"] = dict(list(sites.items())[:4])
    out = sig(out)
    for victim in ("top_sites_ms", "more_rewards_ms", "whole", "config"):
        if len(json.dumps(out)) <= SECONDARY_LIMIT:
            break
        out.pop(victim, None)
    return out


def compact_headline(full):
    """The headline object the driver parses: the contract keys, `config`, `roofline`, `roofline_isolated`, `cpu_baseline`, `whole_pass`,
    `parity`, `extra_summary`, with prose cut to fit HEADLINE_LIMIT bytes.  Optional blocks are dropped, least important first, until it fits."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in full}
    cfg = dict(full.get("config") or {})
    cfg["workload"] = cut(cfg.get("workload", ""), 200)
    if "operands" in cfg:
        cfg["operands"] = cut(cfg["operands"], 90)
    out["config"] = cfg
    rf = pick(full.get("roofline"), ["bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "mfma_util_pct", "kernel",
                                     "flops_per_launch", "bytes_per_launch", "avg_launch_ms", "launches", "frames_per_launch", "streams_sharing_chip"])
    out["roofline"] = rf
    ri = pick(full.get("roofline_isolated"), ["achieved", "peak", "unit", "frac", "avg_launch_ms", "flops_per_launch"])
    if ri:
        out["roofline_isolated"] = ri
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        cb = dict(cb)
        cb["sample"] = cut(cb.get("sample", ""), 230)
    out["cpu_baseline"] = cb
    for k in ("whole_pass", "whole_step", "parity", "rccl", "ranks_seen"):
        if full.get(k) is not None:
            out[k] = full[k]
    if full.get("alt_dtype"):
        out["alt_dtype"] = pick(full["alt_dtype"], ["dtype", "max_cosine_err_vs_oracle", "within_tolerance"])
    if full.get("seam"):
        out["seam"] = pick(full["seam"], ["frames_per_s", "pinned_frames_per_s", "pipelined_frames_per_s", "bit_identical_to_hbm_resident",
                                          "serial_ms_per_step", "prefetched_ms_per_step"])
    if full.get("per_rank_frames_per_s"):
        out["per_rank_frames_per_s"] = full["per_rank_frames_per_s"]
    if full.get("per_rank_samples_per_s"):
        out["per_rank_samples_per_s"] = full["per_rank_samples_per_s"]
    if isinstance(full.get("sites_ms_per_step"), dict):
        out["top_sites_ms"] = dict(list(full["sites_ms_per_step"].items())[:6])
    for k in ("latency_ms", "later_pass_frames_per_s", "rows", "file_rewards_equal_direct_labelling", "more_rewards_ms", "final_aux"):
        if full.get(k) is not None:
            out[k] = full[k]
    if isinstance(full.get("reward"), dict):  # --path online
        out["reward_ms"] = {k: v.get("latency_ms") for k, v in full["reward"].items()}
        out["greedy_action_ms"] = (full.get("greedy_action") or {}).get("latency_ms")
    if full.get("extra_summary"):
        out["extra_summary"] = full["extra_summary"]
    out["full"] = FULL_PATH
    out = sig(out)
    for victim in ("final_aux", "top_sites_ms", "seam", "alt_dtype", "per_rank_frames_per_s", "per_rank_samples_per_s", "roofline_isolated"):
        if len(json.dumps(out)) <= HEADLINE_LIMIT:
            break
        out.pop(victim, None)
    if len(json.dumps(out)) > HEADLINE_LIMIT and isinstance(out.get("extra_summary"), dict):  # many ranks / long summaries: keep value + ms only
        out["extra_summary"] = {k: (pick(v, ["value", "ms_per_step", "parity_err"]) if isinstance(v, dict) else v) for k, v in out["extra_summary"].items()}
    return out


def emit_line(obj):
    """a bench path's ONE line: the complete object when a parent bench.py collects it (ARP_BENCH_CHILD), the compact form otherwise"""
    if os.environ.get("ARP_BENCH_CHILD"):
        emit(json.dumps(obj))
    else:
        emit_report(obj)


def emit_report(full, secondaries=None):
    """Writes gpurun_out/bench_full.json, prints the compact secondary lines, then the headline line LAST.  Refuses (exit 3) a headline over
    HEADLINE_LIMIT: a line the driver cannot parse is an unmeasured round, worse than a failed run that says why."""
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, FULL_PATH), "w") as f:
            json.dump(dict(full, extra=secondaries) if secondaries else full, f)
    except OSError as e:  # a read-only tree must not cost the line
        print(f"bench: could not write {FULL_PATH}: {e!r}", file=sys.stderr)
    for name, d in (secondaries or {}).items():
        line = json.dumps(compact_secondary(name, d))
        if len(line) > SECONDARY_LIMIT:
            line = json.dumps({"secondary": name, "value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step"), "truncated": True})
        emit(line)
    line = json.dumps(compact_headline(full))
    if len(line) > HEADLINE_LIMIT:
        print(f"bench: headline line is {len(line)} bytes > {HEADLINE_LIMIT}; refusing to print a line the driver cannot parse", file=sys.stderr)
        sys.exit(3)
    emit(line)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh copies of this script, one rank per GPU (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set), BEFORE anything here has
    initialised a GPU -- the parent stays a pure launcher.  Rank 0 prints the JSON line; the exit code is the worst child's."""
    import socket
    port = int(os.environ.get("MASTER_PORT", 0))
    if not port:
        # bind-and-close leaves a window in which another process may take the port; SO_REUSEADDR on both sides keeps the rendezvous
        # store able to bind it, and a caller who needs certainty passes MASTER_PORT
        s = socket.socket()
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # the children get the REAL stdout as their fd 1 (this process's own fd 1 points at stderr since claim_stdout)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=ROOT,
                                      stdout=_JSON_FD if _JSON_FD is not None else None))
    # poll ALL children under ONE deadline: a rank that dies early (world > device_count, an RCCL init failure) would otherwise leave
    # its siblings blocked in the rendezvous while this launcher waits on an earlier rank for up to the per-child timeout
    import time
    deadline = time.monotonic() + 3600
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                rc = max(rc, abs(code))
        if time.monotonic() > deadline:
            rc = 124
        if live and rc == 0:
            time.sleep(0.05)
    for p in live:  # first failure (or the deadline): the remaining ranks cannot finish
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    sys.exit(rc)


def run_secondary(a):
    """--all-secondary: the other benches as child processes (each prints its own ONE JSON line), collected under `extra` so that the
    driver's single run of bench.py carries them.  Called before the parent initialises the GPU."""
    # least important first: the driver's 8 KB tail holds the headline line (last) and the compact lines printed just before it
    runs = {
        "h5": ["--path", "h5"],
        "online": ["--path", "online"],
        "label_vit_b16": ["--model", "ViT-B/16", "--batch", "256"],
        # row N1: the parity-true line runs in f32 (the reference's own arithmetic type; its encoder-inside logits meet 1e-3 on every seed,
        # tests/test_m3ae_gpu.py); the f16 line beside it is a throughput mode whose own parity block says whether it is inside
        "policy_with_encoder": ["--path", "policy", "--with-encoder", "--mode", "f32"],
        "policy_with_encoder_f16x3": ["--path", "policy", "--with-encoder", "--mode", "f32", "--encoder-mode", "f16x3"],  # f32-accurate on the 16-bit MFMA
        # round 5: binary16 products with their operand roundings corrected on the fp4 MFMA, encoder AND adapter: the 16-bit line that carries the parity claim
        "policy_with_encoder_f16": ["--path", "policy", "--with-encoder", "--mode", "f16"],
        # round 6: the ordering the 8-GPU steps actually use, on one rank through RCCL (identity reduction)
        "finetune_staged": ["--path", "finetune", "--staged"],
        "finetune": ["--path", "finetune"],
        "policy_staged": ["--path", "policy", "--staged"],
        "policy_with_encoder_f16c": ["--path", "policy", "--with-encoder", "--mode", "f16", "--encoder-mode", "f16c"],
        "policy": ["--path", "policy"],
    }
    # path (2) and row N2 carry their own CPU baseline (a bounded ~5 s sample of the torch-CPU port of the same step, oracle/cpu_baseline_*.py)
    cpu_s = {"policy": 5.0, "finetune": 5.0} if a.cpu_seconds > 0 else {}
    extra = {}
    for name, args in runs.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(a.steps), "--warmup", str(a.warmup), "--cpu-seconds",
               str(cpu_s.get(name, 0)), "--no-secondary"] + args
        try:
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=dict(os.environ, ARP_BENCH_CHILD="1"))
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            extra[name] = json.loads(line[-1]) if line else {"error": (r.stderr or "no output")[-400:], "rc": r.returncode}
        except Exception as e:  # a secondary line must never cost the headline one
            extra[name] = {"error": repr(e)}
    return extra


def summarize_extra(extra, seam):
    """value / ms per step / CPU baseline of every secondary line + the host-fed seam rates, compact enough for the head of the JSON line"""
    out = {}
    if seam:
        out["seam_host_fed_frames_per_s"] = {"sync_call": round(seam["frames_per_s"]), "pipelined_calls": round(seam["pipelined_frames_per_s"])}
    for name, d in (extra or {}).items():
        if not isinstance(d, dict) or "value" not in d:
            out[name] = "failed"
            continue
        e = {"value": round(d["value"], 4 if d.get("unit") == "ms" else 1), "unit": d.get("unit"), "ms_per_step": round(d.get("ms_per_step", 0.0), 4)}
        cb = d.get("cpu_baseline")
        if cb and cb.get("value"):
            e["cpu_baseline"] = {"value": round(cb["value"], 1), "cores": cb.get("cores")}
        par = d.get("parity")
        if par and par.get("max_logit_err_vs_oracle") is not None:
            e["parity_err"] = float(f"{par['max_logit_err_vs_oracle']:.3g}")
        if par and par.get("max_cosine_err_vs_oracle") is not None:
            e["parity_err"] = float(f"{par['max_cosine_err_vs_oracle']:.3g}")
        if name == "online":
            e = {"reward_ms": {k: v["latency_ms"] for k, v in d.get("reward", {}).items()}, "greedy_action_ms": d.get("greedy_action", {}).get("latency_ms")}
            e.update({k: v for k, v in d.get("more_rewards_ms", {}).items()})
        if name == "h5":
            e = {"file_to_file_frames_per_s": d.get("value"), "later_pass": d.get("later_pass_frames_per_s"), "rows": d.get("rows")}
        out[name] = e
    return out


def gather_rates(dist, world, units, elapsed_local):
    """per-rank units/s over the timed region (control plane, gloo)"""
    if dist is None:
        return [units / elapsed_local]
    out = [None] * world
    dist.all_gather_object(out, float(elapsed_local))
    return [units / t for t in out]


def gather_cert(dist, world, cert):
    """The `rccl` block of a multi-GPU line: every rank's own certificate (arp_amd.train.certify_collective: what ncclCommCount /
    ncclCommUserRank report, and an all-reduce(sum) of rank + 1 through the step's communicator) gathered over the gloo control plane,
    so that "did RCCL see N ranks" is read off the line, not off the environment."""
    certs = [cert]
    if dist is not None:
        certs = [None] * world
        dist.all_gather_object(certs, cert)
    return {"rccl_nranks": sorted({c["nranks"] for c in certs}), "ranks_seen": sorted(c["rank"] for c in certs), "devices": [c["device"] for c in certs],
            "allreduce_selfcheck": [c["allreduce_selfcheck"] for c in certs], "expected": certs[0]["expected"],
            "ok": bool(all(c["ok"] for c in certs) and sorted(c["rank"] for c in certs) == list(range(world))),
            "has_comm": bool(all(c["has_comm"] for c in certs)), "rccl_version": certs[0]["rccl_version"],
            "NCCL_ALGO": certs[0]["NCCL_ALGO"], "NCCL_PROTO": certs[0]["NCCL_PROTO"]}


def label_ranks_seen(dist, rank, world, device):
    """labelling shards with NO collective: what certifies an N-GPU line is the control plane -- every rank reports itself and a
    gloo all-reduce of rank + 1 must read N (N + 1) / 2"""
    if dist is None:
        return {"ranks_seen": [0], "devices": [device], "control_plane_selfcheck": 1.0, "expected": 1.0, "ok": True}
    import torch
    seen = [None] * world
    dist.all_gather_object(seen, (rank, device, os.getpid()))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    want = world * (world + 1) / 2.0
    return {"ranks_seen": sorted(r for r, _, _ in seen), "devices": [d for _, d, _ in sorted(seen)], "distinct_processes": len({p for _, _, p in seen}),
            "control_plane_selfcheck": float(t), "expected": want, "ok": bool(float(t) == want and sorted(r for r, _, _ in seen) == list(range(world)))}


def policy_sites(cfg, B, n_params):
    """Algorithmic work per launch of the policy step's call sites: ("mfma", FLOPs) or ("hbm", bytes)."""
    R = B * cfg.window
    Mx, D, E, Kin = R * cfg.enc_tokens, cfg.enc_dim, cfg.emb, cfg.enc_tokens * cfg.enc_dim
    ad = 2.0 * Mx * D * D
    iti = 2.0 * R * Kin * E
    return {
        "dt.adapter_fc1": ("mfma", ad), "dt.adapter_fc2": ("mfma", ad), "dt.adapter_fc2_dX": ("mfma", ad), "dt.adapter_fc2_dW": ("mfma", ad),
        "dt.adapter_fc1_dW": ("mfma", ad), "dt.image_text_input": ("mfma", iti), "dt.image_text_input_dW": ("mfma", iti),
        "dt.image_text_input_dX": ("mfma", iti),
        # norms pass reads p, g; the update reads p, g, mu, nu and writes p, mu, nu: 9 x 4 B per parameter
        "dt.clip_adam": ("hbm", 36.0 * n_params),
    }


def encoder_sites(ecfg, frames):
    """Algorithmic FLOPs per launch of the frozen M3AE encoder's GEMM call sites (one launch covers all `frames` frames of the step)."""
    m, d = frames * ecfg.tokens, ecfg.width
    return {
        "m3ae.image_embedding": ("mfma", 2.0 * frames * (ecfg.tokens - 1) * d * 3 * ecfg.patch * ecfg.patch),
        "m3ae.qkv": ("mfma", 2.0 * m * 3 * d * d), "m3ae.out_proj": ("mfma", 2.0 * m * d * d),
        "m3ae.c_fc": ("mfma", 2.0 * m * ecfg.mlp_ratio * d * d), "m3ae.c_proj": ("mfma", 2.0 * m * ecfg.mlp_ratio * d * d),
        "m3ae.attn": ("mfma", 4.0 * frames * ecfg.tokens * ecfg.tokens * d),
    }


def policy_step_flops(cfg, B):
    R = B * cfg.window
    Mx, D, E, H, Kin = R * cfg.enc_tokens, cfg.enc_dim, cfg.emb, cfg.mlp_ratio * cfg.emb, cfg.enc_tokens * cfg.enc_dim
    adapter = 6 * 2.0 * Mx * D * D                                # fwd 2, bwd 4 GEMMs (the first layer needs no dX: enc is stop_gradient'ed)
    iti = 3 * 2.0 * R * Kin * E                                   # image_text_input fwd, dW, dX
    tok = 3 * 2.0 * (3 * R) * cfg.depth * (4 * E * E + 2 * E * H)  # 12-token transformer, fwd + 2x bwd
    return adapter + iti + tok


def staged_block(train, cfg, prof, steps, elapsed):
    """--staged: what the data-parallel ordering of the policy step costs on ONE rank (the line's ms_per_step IS the staged step) and what it hands to RCCL"""
    ranges, total = train.bucket_plan(cfg)
    per = lambda k: round(prof[k]["ms"] / max(prof[k]["calls"], 1), 4) if k in prof else None
    return {"staged_ms_per_step": round(elapsed / steps * 1e3, 4),
            "ordering": "stage 1 (forward, transformer backward, image_text_input dW) -> bucket 1 on the communication stream || stage 2 (adapter backward) -> bucket 2 + loss scalars -> norms, clip, Adam; dWi is NOT produced last here",
            "allreduce_b1_ms": per("dt.allreduce_b1"), "allreduce_b2_ms": per("dt.allreduce_b2"),
            "bucket_bytes": [4 * sum(hi - lo for lo, hi in ranges[:2]), 4 * sum(hi - lo for lo, hi in ranges[2:])], "gradient_bytes": 4 * total,
            "reduction": "ncclAllReduce(sum) over ONE rank (identity): the calls, streams, events and staged graphs of the 8-GPU step, no xGMI traffic"}


def bench_policy(a):
    """Secondary benchmark: ARPDT train_step (BASELINE.json configs[3]): B = 32 samples per GPU, T = 4, random-init
    M3AE-shaped encodings [B,4,257,768] resident in HBM, forward + backward + RCCL all-reduce + clip + Adam."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu = cpu_baseline(a.model, a.cpu_seconds, a.policy_batch) if (rank == 0 and world == 1 and a.cpu_seconds > 0) else None
    import torch  # before arp_amd: one HIP runtime per process (arp_amd/_ffi.py); the parity gate and the control plane need it
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C_
    from arp_amd import _ffi, clip, synth_policy as S, train
    from arp_amd.train import PolicyConfig, PolicyTrainer
    _ffi.require_gpu()
    if world > _ffi.device_count():
        raise SystemExit(f"--path policy --gpus {world}: {_ffi.device_count()} GPU(s) visible; RCCL needs one GPU per rank")
    _ffi.check(_ffi.lib.arp_set_device(local_rank))
    staged = bool(a.staged and world == 1)
    if staged:  # read when the handle is created
        os.environ["ARP_DT_FORCE_COMM"] = "1"
        os.environ["ARP_DT_OVERLAP"] = "1"
    cfg = PolicyConfig(lambda_ret=0.01)
    # row N1's 16-bit parity line: the f16c encoder (operand roundings corrected on the fp4 MFMA) goes with the same corrections on the policy's adapter
    # (round 6: the corrected adapter is the f16 policy's default; the plain-f16-encoder line stays the all-plain throughput mode it has been since round 2)
    emode_ = (a.encoder_mode or a.mode) if a.with_encoder else None
    adapter_c = a.mode == "f16" and not a.no_adapter_c and (a.adapter_c or emode_ not in ("f16", "bf16"))
    # parity gate (rank 0): the mode timed below, real geometry (257 x 768 encodings, K = 197 376), B = 2, against the fp64 oracle
    parity = None
    parity_geometry = "B = 2, window 4, 257 x 768 encodings in (K = 197 376)"
    if rank == 0 and a.parity_frames > 0:
        from oracle import arpdt_torch as O
        Pp = S.policy_params(cfg, seed=3)
        t0 = PolicyTrainer(cfg, mode=a.mode, device=local_rank, adapter_corrections=adapter_c)
        t0.set_params(Pp)
        if a.with_encoder:
            # the configuration that is timed: FRAMES in, the frozen encoder in the timed mode in front of the policy in the timed mode, against
            # oracle/m3ae_np -> oracle/arpdt_torch in fp64 on the same frames (tests/test_m3ae_gpu.py, full geometry)
            from arp_amd import m3ae
            from oracle import m3ae_np as MO
            ecfg0 = m3ae.EncoderConfig()
            EP = S.m3ae_params(MO.EncConfig(), seed=50)
            frames0 = S.normalized_frames(2 * cfg.window, 256, seed=80).reshape(2, cfg.window, 256, 256, 3)
            _, act, rtg = S.policy_batch(PolicyConfig(enc_tokens=1, enc_dim=4), 2, seed=4)
            codes = MO.forward_representation(EP, MO.EncConfig(), frames0.reshape(-1, 256, 256, 3)).reshape(2, cfg.window, ecfg0.tokens, ecfg0.width)
            ref = O.forward({k: torch.from_numpy(v).double() for k, v in Pp.items()}, O.PolicyConfig(lambda_ret=0.01),
                            torch.from_numpy(np.asarray(codes, np.float64)), torch.from_numpy(act).long(), torch.from_numpy(rtg).double())
            enc0 = m3ae.M3AEEncoder(ecfg0, EP, mode=a.encoder_mode or a.mode, device=local_rank, max_frames=2 * cfg.window)
            t0.attach_encoder(enc0)
            t0.set_batch_images(frames0, act, rtg)
            out = t0.forward()
            t0.close()
            enc0.close()
            parity_geometry = "B = 2, window 4, 8 frames [256,256,3] in -> frozen M3AE ViT-B/16 encoder (257 tokens) -> policy, vs oracle/m3ae_np -> oracle/arpdt_torch (fp64)"
        else:
            enc, act, rtg = S.policy_batch(cfg, 2, seed=4)
            ref = O.forward({k: torch.from_numpy(v).double() for k, v in Pp.items()}, O.PolicyConfig(lambda_ret=0.01), torch.from_numpy(enc).double(),
                            torch.from_numpy(act).long(), torch.from_numpy(rtg).double())
            t0.set_batch(enc, act, rtg)
            out = t0.forward()
            t0.close()
        parity = max(float(np.abs(out["action_pred"] - ref["action_pred"].numpy()).max()), float(np.abs(out["return_pred"] - ref["return_pred"].numpy()).max()))
    if a.churn and a.with_encoder:  # diagnostic (scripts/r6_n1_flow3.sh): which part of the parity gate's throw-away work changes the timed objects' speed?
        from arp_amd import m3ae
        kinds = a.churn.split(",")
        if "malloc" in kinds:
            ps = []
            for _ in range(8):
                p_ = C_.c_void_p()
                _ffi.check(_ffi.lib.arp_dev_malloc(C_.byref(p_), 512 << 20))
                ps.append(p_)
            for p_ in ps:
                _ffi.check(_ffi.lib.arp_dev_free(p_))
        if "h2d" in kinds:
            p_ = C_.c_void_p()
            _ffi.check(_ffi.lib.arp_dev_malloc(C_.byref(p_), 128 << 20))
            hb = np.zeros(128 << 20, np.uint8)
            _ffi.check(_ffi.lib.arp_memcpy_h2d(p_, hb.ctypes.data_as(C_.c_void_p), hb.nbytes))
            _ffi.check(_ffi.lib.arp_dev_free(p_))
        t9 = e9 = None
        if "trainer" in kinds or "full" in kinds or "fwd_enc" in kinds:
            t9 = PolicyTrainer(cfg, mode=a.mode, device=local_rank, adapter_corrections=adapter_c)
            t9.set_params(S.policy_params(cfg, seed=3))
        if "encoder" in kinds or "full" in kinds:
            e9 = m3ae.M3AEEncoder(m3ae.EncoderConfig(), S.m3ae_params(m3ae.EncoderConfig(), seed=50), mode=a.encoder_mode or a.mode, device=local_rank, max_frames=2 * cfg.window)
        if "fwd_enc" in kinds:  # the trainer's forward on ENCODINGS (no encoder involved)
            t9.set_batch(*S.policy_batch(cfg, 2, seed=4))
            t9.forward()
        if "encfwd" in kinds and e9 is not None:  # the encoder alone
            e9.forward_representation(S.normalized_frames(2 * cfg.window, 256, seed=80))
        if "full" in kinds:
            t9.attach_encoder(e9)
            _, act9, rtg9 = S.policy_batch(PolicyConfig(enc_tokens=1, enc_dim=4), 2, seed=4)
            t9.set_batch_images(S.normalized_frames(2 * cfg.window, 256, seed=80).reshape(2, cfg.window, 256, 256, 3), act9, rtg9)
            t9.forward()
        if t9 is not None:
            t9.close()
        if e9 is not None:
            e9.close()
    tr = PolicyTrainer(cfg, mode=a.mode, device=local_rank, adapter_corrections=adapter_c)
    tr.set_params(S.policy_params(cfg, seed=0))
    if world > 1:  # RCCL id from rank 0 over gloo, communicator, sync_state_fn
        train.DataParallel(tr, rank, world, train.torch_object_broadcast(dist))
    elif staged:  # a one-rank communicator: the staged graphs, the communication stream and the RCCL calls of the DP step, reducing over one rank
        tr.comm_init(PolicyTrainer.new_unique_id(), 1, 0)
        tr.broadcast_state()
    rccl = gather_cert(dist, world, train.certify_collective(tr, rank, world))  # every rank: the gather is a collective of the control plane
    if not rccl["ok"]:
        raise SystemExit(f"--path policy --gpus {world}: the communicator does not span the ranks it should: {rccl}")
    enc = None
    if a.with_encoder:
        from arp_amd import m3ae
        ecfg = m3ae.EncoderConfig()
        enc = m3ae.M3AEEncoder(ecfg, S.m3ae_params(ecfg, seed=0), mode=a.encoder_mode or a.mode, device=local_rank, max_frames=a.policy_batch * cfg.window)
        tr.attach_encoder(enc)
        # Frames in, as the training loop feeds them: prefetch_to_device keeps TWO batches of frames on the device (main_procgen.py:703) and the step alternates
        # between the slots.  The encoder is frozen, so batch i + 1 is encoded on the encoder's own stream while step i's policy part runs
        # (arp_dt_encode_ahead -- what prefetch_to_device's uploader thread calls once a batch has landed): every step still pays for exactly one encoder
        # pass over its own 128 frames, inside the timed region (--no-encode-ahead: each step encodes at its own head, as in rounds 1-5).
        ahead = not a.no_encode_ahead and os.environ.get("ARP_DT_ENC_EAGER", "1") != "0" and not a.single_slot
        for k in (() if a.single_slot else (0, 1)):
            _, act_, rtg_ = S.policy_batch(PolicyConfig(enc_tokens=1, enc_dim=4), a.policy_batch, seed=100 + rank + 17 * k)
            frames = S.normalized_frames(a.policy_batch * cfg.window, 256, seed=100 + rank + 17 * k).reshape(a.policy_batch, cfg.window, 256, 256, 3)
            tr.upload_async(k, frames, act_, rtg_, images=True)
        if a.single_slot:  # rounds 1-5: ONE batch staged synchronously, every step encodes it at its head
            _, act_, rtg_ = S.policy_batch(PolicyConfig(enc_tokens=1, enc_dim=4), a.policy_batch, seed=100 + rank)
            frames = S.normalized_frames(a.policy_batch * cfg.window, 256, seed=100 + rank).reshape(a.policy_batch, cfg.window, 256, 256, 3)
            tr.set_batch_images(frames, act_, rtg_)
        step_no = [0]
        if ahead:
            tr.encode_ahead(0)
        settle = 6  # set-up, not warm-up: each slot's chain runs eagerly twice before it is captured (arp_dt.hip::fwd_bwd_graphed) -- three steps per slot

        def step():
            k = step_no[0]
            step_no[0] += 1
            if not a.single_slot:
                tr.select(k & 1)
            tr.train_step_async(lr)
            if ahead:
                tr.encode_ahead((k + 1) & 1)
    else:
        tr.set_batch(*S.policy_batch(cfg, a.policy_batch, seed=100 + rank))
        settle = 0

        def step():
            tr.train_step_async(lr)
    lr = 5e-4
    for _ in range(settle + a.warmup):
        step()
    tr.sync()
    if dist is not None:
        dist.barrier()
    _ffi.check(_ffi.lib.arp_dev_synchronize())
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    tr.sync()
    _ffi.check(_ffi.lib.arp_dev_synchronize())
    elapsed = time.perf_counter() - t0
    per_rank = gather_rates(dist, world, a.policy_batch * a.steps, elapsed)
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tr.profile(True)
    tr.profile_reset()
    for _ in range(a.steps):
        step()
    tr.sync()
    _ffi.check(_ffi.lib.arp_dev_synchronize())
    prof = tr.profile_read()
    if enc is not None:
        enc.profile(True)
        step()
        tr.sync()
        _ffi.check(_ffi.lib.arp_dev_synchronize())
        prof.update({k: {"ms": v["ms"] * a.steps, "calls": v["calls"] * a.steps} for k, v in enc.profile_read().items()})
        enc.profile(False)
    aux = tr.train_step(lr)
    # ---- the S4 seam as the reference calls it: train_step_fn(state, HOST batch, rng) (main_procgen.py:718), outside the timed region.
    # serial = the 101 MB batch uploaded synchronously in front of every step; prefetched = prefetch_to_device(.., 2): batch i+1 goes up
    # on the copy stream from a background thread while step i runs (PCIe-bound: 101 MB per step).
    seam = None
    if rank == 0 and world == 1 and enc is None:
        from arp_amd.train import TrainState, create_train_step, prefetch_to_device
        hb = S.policy_batch(cfg, a.policy_batch, seed=100)
        host_batch = {"image": {"ob": hb[0]}, "action": hb[1], "rtg": {"ob": hb[2]}}
        fn = create_train_step(cfg, lambda step: lr, cfg.weight_decay)
        n_seam = 12
        res = {}
        for name in ("serial", "prefetched"):
            state = TrainState(tr)
            src = (host_batch for _ in range(n_seam + 2))
            if name == "prefetched":
                src = prefetch_to_device(src, 2, tr)
            t_s, k = None, 0
            for b in src:
                if k == 2:
                    t_s = time.perf_counter()
                state, _, _ = fn(state, b, None)
                k += 1
            res[name] = (time.perf_counter() - t_s) / n_seam * 1e3
        seam = {"serial_ms_per_step": res["serial"], "prefetched_ms_per_step": res["prefetched"], "host_batch_mb": hb[0].nbytes / 1e6,
                "call": "train_step_fn(state, host batch dict, rng) incl. the H2D upload of the f32 encodings and the aux read-back"}
    if rank == 0:
        sites = policy_sites(cfg, a.policy_batch, tr.num_params)
        if enc is not None:  # the frozen encoder's launches are part of this step: the dominant site is chosen over them as well
            sites.update(encoder_sites(ecfg, a.policy_batch * cfg.window))
        known = {k: v for k, v in prof.items() if k in sites and v["calls"]}
        site = max(known, key=lambda k: known[k]["ms"])  # the call site that takes the most time per step
        kind, work = sites[site]
        avg_ms = prof[site]["ms"] / max(prof[site]["calls"], 1)
        emode = a.encoder_mode or a.mode
        # the matrix pipe a site runs on: the encoder's sites follow the encoder's mode (f16x3: the 16-bit MFMA, three instructions per algorithmic product)
        peak = (PEAK_TFLOPS["f16" if emode in ("f16x3", "f16c") else emode] if site.startswith("m3ae.") else PEAK_TFLOPS[a.mode]) if kind == "mfma" else 8000.0
        achieved = work / (avg_ms * 1e-3) / (1e12 if kind == "mfma" else 1e9)
        flops = policy_step_flops(cfg, a.policy_batch)
        if enc is not None:  # + the frozen encoder's forward: 46.4 GFLOP per frame, 4 frames per sample (193.5 GF per sample in all)
            flops += m3ae.flops_per_frame(ecfg) * a.policy_batch * cfg.window
        # HBM bytes of the dominant site from the PMC counters (scripts/prof_policy_pmc.sh -> profiles/pmc_traffic_policy.json; measured
        # at the default geometry, B = 32 per GPU)
        traffic = traffic_stale = None
        try:
            recs = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_policy.json")))
            rec = recs["sites"].get(site)
            if rec and a.policy_batch == 32:
                traffic = rec["hbm_bytes_per_launch"]
                from arp_amd._srchash import csrc_sha1
                traffic_stale = recs.get("csrc_sha1") != csrc_sha1()  # (a file with no stamp at all is stale by definition)
        except (OSError, ValueError, KeyError):
            traffic = None
        emit_line(({
            "metric": ("samples/sec ARPDT train_step (frames in, frozen M3AE encoder inside)" if enc is not None else
                       "samples/sec ARPDT train_step (trainable part, encodings in)") + (" -- STAGED data-parallel ordering on one rank" if staged else ""), "value": world * a.policy_batch * a.steps / elapsed,
            "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.mode if enc is None or (a.encoder_mode or a.mode) == a.mode else f"encoder {a.encoder_mode}, policy {a.mode}",
            "data": "synthetic",
            "config": {"workload": (f"ARPDT policy train_step, {a.policy_batch} samples/GPU x window 4, FRAMES in: normalised f32 frames [B,4,256,256,3] resident in "
                                    f"HBM -> frozen random-init M3AE ViT-B/16 encoder (257 tokens, {a.encoder_mode or a.mode}) -> 26.9 M trainable params, policy in {a.mode} (SURVEY row N1; "
                                    f"BASELINE.json configs[3] with the reference's own boundary)") if enc is not None else
                                   (f"ARPDT policy train_step, {a.policy_batch} samples/GPU x window 4, random-init encodings [B,4,257,768] f32 "
                                    f"resident in HBM, 26.9 M trainable params (BASELINE.json configs[3])"), "parallelism": f"dp{world}",
                       "collective": ("RCCL all-reduce(sum) of the flat f32 gradient (107.5 MB) in two buckets on a communication stream -- bucket 1 "
                                      "(image_text_input + transformer + heads, 94 % of the bytes) under the adapter's backward, bucket 2 + 4 loss scalars after it") if world > 1 else "none (1 rank)"},
            "roofline": {"bound": kind, "achieved": achieved, "peak": peak, "unit": "TFLOP/s" if kind == "mfma" else "GB/s",
                         "frac": achieved / peak, "traffic": traffic, "traffic_stale": traffic_stale, "kernel": f"{'gemm' if kind == 'mfma' else 'norms_partial + adam_kernel'} @ {site}",
                         ("flops_per_launch" if kind == "mfma" else "bytes_per_launch"): work, "avg_launch_ms": avg_ms,
                         "traffic_source": None if traffic is None else "profiles/pmc_traffic_policy.json (committed rocprofv3 --pmc passes, not measured in this run)",
                         "note": "the call site with the largest share of the step (sites_ms_per_step)"},
            "whole_step": {"gflop_per_step": flops / 1e9, "tflops": flops / (elapsed / a.steps) / 1e12,
                           "mfma_frac_of_peak": flops / (elapsed / a.steps) / 1e12 / (PEAK_TFLOPS["f16" if (a.encoder_mode or a.mode) in ("f16x3", "f16c") else (a.encoder_mode or a.mode)] if enc is not None else PEAK_TFLOPS[a.mode])},
            "parity": {"max_logit_err_vs_oracle": parity, "geometry": parity_geometry, "tolerance": 1e-3,
                       "within_tolerance": None if parity is None else bool(parity < 1e-3)},
            "cpu_baseline": cpu, "seam": seam, "final_aux": aux, "per_rank_samples_per_s": [round(v, 1) for v in per_rank], "rccl": rccl,
            "staged": staged_block(train, cfg, prof, a.steps, elapsed) if staged else None,
            "sites_ms_per_step": {k: round(v["ms"] / a.steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}))
    tr.close()
    if dist is not None:
        dist.destroy_process_group()


def bench_online(a):
    """Secondary benchmark (the rollout loop): latency of ONE call -- `get_torch_clip_reward` on one 256 x 256 x 3 host frame
    (envs/vl_reward.py:11-23) for both CLIP models, and `greedy_action` on one window of encodings (rollout_procgen.py:123-155)."""
    import time
    from arp_amd import _ffi, clip, synth, label_reward as L, synth_policy as S
    from arp_amd.train import PolicyConfig, PolicyTrainer
    _ffi.require_gpu()
    _ffi.check(_ffi.lib.arp_set_device(0))
    reps = max(50, 10 * a.steps)
    lat = {}
    for name in ("ViT-B/32", "ViT-B/16"):
        cfg = clip.MODELS[name]
        m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode=a.mode, max_batch=64, n_streams=1).set_text(synth.prompt_tokens(1, 8, seed=2))
        fr = synth.procgen_like_frames(8, seed=3)
        for i in range(8):
            L.get_torch_clip_reward(m, fr[i])
        ts = []
        for i in range(reps):
            t0 = time.perf_counter()
            L.get_torch_clip_reward(m, fr[i % 8])
            ts.append(time.perf_counter() - t0)
        d_fr = clip.DeviceBuffer(fr[:1].nbytes); d_fr.upload(fr[:1])
        d_rw = clip.DeviceBuffer(4)
        e0, e1 = clip.Event(), clip.Event()
        for _ in range(3):
            m.label_device_async(d_fr, 1, 256, 256, d_rw)
        m.sync(); m.record(e0)
        for _ in range(reps):
            m.label_device_async(d_fr, 1, 256, 256, d_rw)
        m.record(e1); m.sync()
        dev = clip.elapsed_ms(e0, e1) / reps
        m.profile(True); m.profile_reset()
        m.label(fr[:1])
        launches = int(sum(v["calls"] for v in m.profile_read().values())); m.profile(False)
        lat[name] = {"latency_ms": round(float(np.median(ts)) * 1e3, 4), "mean_ms": round(float(np.mean(ts)) * 1e3, 4),
                     "p99_ms": round(float(np.quantile(ts, 0.99)) * 1e3, 4), "device_ms": round(dev, 4), "launches": launches, "calls": reps}
        m.close()
    # the rest of the rollout loop's reward dispatch (envs/rollout_procgen.py:133-151; vl_reward.py:19-22,26-41,44-79) on the model the
    # reference loads (ViT-B/16): a list of prompts, the goal-conditioned distance (two frames per call), and both with the fine-tuned head
    def med(fn, n=reps):
        for _ in range(5):
            fn()
        ts_ = []
        for _ in range(n):
            t0_ = time.perf_counter()
            fn()
            ts_.append(time.perf_counter() - t0_)
        return round(float(np.median(ts_)) * 1e3, 4)

    more = {}
    cfg = clip.MODELS["ViT-B/16"]
    fr = synth.procgen_like_frames(8, seed=3)
    W16 = synth.clip_weights(cfg, seed=0)
    m = clip.ClipLabeller(cfg, W16, mode=a.mode, max_batch=64, n_streams=1).set_text(synth.prompt_tokens(3, [8, 6, 9], seed=2))
    k = [0]

    def nxt():
        k[0] = (k[0] + 1) % 7
        return fr[k[0]]
    more["clip_list_of_3_prompts"] = med(lambda: L.get_torch_clip_reward(m, nxt(), ["a", "b", "c"]))
    more["clip_goal_conditioned"] = med(lambda: L.get_torch_clip_goal_conditioned_reward(m, nxt(), fr[7]))
    m.close()
    from arp_amd import finetune as FT
    hcfg = FT.FinetuneConfig()
    ckpt = {**{"clip_model." + kk: v for kk, v in W16.items()}, **FT.synth_params(hcfg, seed=0)}
    fm = FT.FinetunedClip.from_state_dict(ckpt, mode=a.mode, model=cfg).set_text(synth.prompt_tokens(1, 8, seed=2))
    del ckpt, W16
    more["clip_ft"] = med(lambda: L.get_torch_clip_adapter_reward(fm, nxt(), "a"), max(reps // 2, 25))
    more["clip_ft_goal_conditioned"] = med(lambda: L.get_torch_clip_adapter_goal_conditioned_reward(fm, nxt(), fr[7]), max(reps // 2, 25))
    fm.close()
    pcfg = PolicyConfig(lambda_ret=0.01)
    tr = PolicyTrainer(pcfg, mode=a.mode)
    tr.set_params(S.policy_params(pcfg, seed=0))
    enc, act, rtg = S.policy_batch(pcfg, 1, seed=5)
    for _ in range(5):
        tr.greedy_action(enc, act, rtg)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        tr.greedy_action(enc, act, rtg)
        ts.append(time.perf_counter() - t0)
    tr.close()
    ga = {"latency_ms": round(float(np.median(ts)) * 1e3, 4), "mean_ms": round(float(np.mean(ts)) * 1e3, 4), "enc_bytes": int(enc.nbytes), "calls": reps}
    v = lat["ViT-B/32"]["latency_ms"]
    emit_line(({"metric": "single_frame_reward_latency", "value": v, "unit": "ms", "latency_ms": v, "n_gpus": 1, "steps": reps, "warmup": 8,
                     "ms_per_step": v, "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": a.mode, "data": "synthetic",
                     "config": {"workload": "get_torch_clip_reward: one 256x256x3 uint8 host frame in, one f32 reward out per call (ViT-B/32; the same for "
                                            "ViT-B/16 under reward); greedy_action: one [1,4,257,768] f32 window of encodings in, one action out"},
                     "reward": lat, "greedy_action": ga,
                     "more_rewards_ms": more, "more_rewards_note": "ViT-B/16, one host frame (two for the goal-conditioned ones) in, one reward out per call: "
                     "get_torch_clip_reward with a list of 3 prompts, get_torch_clip_goal_conditioned_reward, get_torch_clip_adapter_reward and "
                     "get_torch_clip_adapter_goal_conditioned_reward (frozen towers + the 476 M-parameter fine-tuned head at batch 1)"}))


def bench_h5(a):
    """Secondary benchmark: label_reward(data_path=...) end to end on a recorder-style HDF5 file -- open, scan `done`, inflate the last
    frame of every row, label (ViT-B/32, host-fed), write both reward datasets as gzip chunks, close (label_reward.py:69-87,256-289)."""
    import tempfile
    from arp_amd import _ffi, clip, h5store, label_reward as L, synth
    _ffi.require_gpu()
    _ffi.check(_ffi.lib.arp_set_device(0))
    rows, tlen, F = a.h5_rows, 256, 8
    path = os.path.join(tempfile.gettempdir(), f"arp_bench_h5_{os.getpid()}.hdf5")
    t0 = time.perf_counter()
    with h5store.H5Store(path, "w") as f:
        f.attrs["env_name"] = "coinrun"
        for s0 in range(0, rows, tlen):
            n = min(tlen, rows - s0)
            # Procgen renders 64x64 natively; the recorded 256x256 observation is that picture enlarged (what gzip sees in a real file)
            rng = np.random.default_rng(s0)
            base = rng.integers(0, 6, (n, 16, 16, 1)).repeat(4, 1).repeat(4, 2) * 40 + rng.integers(0, 3, (n, 64, 64, 3)) * 5
            fr = base.astype(np.uint8).repeat(4, 1).repeat(4, 2)
            idx = np.clip(np.arange(n)[:, None] + np.arange(-F + 1, 1)[None, :], 0, None)  # the recorder's sliding window of 8 observations
            d = np.zeros((n, F), np.float32)
            d[-1, -1] = 1
            if s0 == 0:
                f.create_dataset("ob", data=fr[idx], compression="gzip", chunks=(1, F, 256, 256, 3), maxshape=(None, F, 256, 256, 3))
                f.create_dataset("done", data=d, compression="gzip", chunks=(1, F), maxshape=(None, F))
            else:
                for k, v in (("ob", fr[idx]), ("done", d)):
                    ds = f[k]
                    n0 = ds.shape[0]
                    ds.resize(n0 + n, axis=0)
                    ds[n0:] = v
    t_write = time.perf_counter() - t0
    size_mb = os.path.getsize(path) / 1e6
    cfg = clip.MODELS["ViT-B/32"]
    m = clip.ClipLabeller(cfg, synth.clip_weights(cfg, seed=0), mode=a.mode, device=0)
    tok = synth.prompt_tokens(1, 8, seed=2)
    rates = []
    keys = ("ob_clip_reward", "ob_clip_pos_rtg")
    try:
        # pass 0: the first labelling call of this PROCESS -- it also pays the GPU runtime's lazy work (code objects loaded at the first launch of every kernel, the
        # labeller's device / pinned buffers, the text tower's first run): a cost per process, not per file.  Its two datasets are then deleted, so that pass 1 is
        # again a FIRST pass over the file (both label datasets created and filled) by a process that has labelled before -- `value`.  Passes 2, 3 overwrite.
        for k in range(4):
            t = time.perf_counter()
            L.label_reward("coinrun", "hard", 500, 0, "x", ".", data_path=path, clip_model=m, tokens=tok)
            rates.append(rows / (time.perf_counter() - t))
            if k == 0:
                with h5store.H5Store(path, "a") as f:
                    for key in keys:
                        del f[key]
        with h5store.H5Store(path, "r") as f:
            same = bool(np.array_equal(f["ob_clip_reward"][0:64, -1], m.label(f["ob"][0:64, -1])))
    finally:
        m.close()
        os.remove(path)
    emit_line(({"metric": "frames/sec label_reward(data_path=HDF5 file) end to end (file -> file), first pass over the file", "value": round(rates[1], 1), "unit": "frames/s",
                     "later_pass_frames_per_s": round(max(rates[2:]), 1), "cold_process_first_call_frames_per_s": round(rates[0], 1),
                     "n_gpus": 1, "steps": 3, "warmup": 1, "ms_per_step": rows / rates[1] * 1e3,
                     "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.mode, "data": "synthetic", "rows": rows,
                     "config": {"workload": f"recorder-schema file: ob uint8 [{rows},8,256,256,3] in gzip chunks of one row ({size_mb:.0f} MB on disk, "
                                            f"{rows * F * 196608 / 1e9:.1f} GB raw; written here in {t_write:.1f} s, page-cache resident), trajectories of {tlen} rows; "
                                            "`value` = a pass that CREATES ob_clip_reward / ob_clip_pos_rtg (the datasets of the process's very first call -- "
                                            "cold_process_first_call_frames_per_s: code-object loads, first allocations -- are deleted before it); later passes overwrite them"},
                     "file_rewards_equal_direct_labelling": same}))


def bench_finetune(a):
    """Secondary benchmark: the CLIP multi-scale adapter fine-tune head step (BASELINE.json configs[4]):
    B = 64 samples x 3 frames, frozen-tower features resident in HBM, forward + backward + AdamW over 476 M parameters.
    Single GPU, as the reference (finetune_module/finetune.py)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu = cpu_baseline(a.model, a.cpu_seconds, finetune_batch=a.finetune_batch) if (a.cpu_seconds > 0 and rank == 0 and world == 1) else None
    dist = None
    if world > 1:  # DP = N (BASELINE configs[4] names DP = 8): one rank per GPU, one RCCL all-reduce of the 1.9 GB gradient per step
        import torch  # noqa: F401  -- before arp_amd: one HIP runtime per process
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from arp_amd import _ffi, finetune as FT, train
    _ffi.require_gpu()
    if world > _ffi.device_count():
        raise SystemExit(f"--path finetune --gpus {world}: {_ffi.device_count()} GPU(s) visible; RCCL needs one GPU per rank")
    _ffi.check(_ffi.lib.arp_set_device(local_rank))
    cfg = FT.FinetuneConfig()
    staged = bool(a.staged and world == 1)
    if staged:  # read when the handle is created: the seven-bucket all-reduce from inside the backward, through a one-rank communicator
        os.environ["ARP_FT_FORCE_COMM"] = "1"
        os.environ["ARP_FT_OVERLAP"] = "1"
    tr = FT.FinetuneTrainer(cfg, mode=a.mode, device=local_rank)
    tr.set_params(FT.synth_params(cfg, seed=0))
    if world > 1:
        FT.DataParallel(tr, rank, world, train.torch_object_broadcast(dist))
    elif staged:
        tr.comm_init(FT.FinetuneTrainer.new_unique_id(), 1, 0)
        tr.broadcast_state()
    rccl = gather_cert(dist, world, train.certify_collective(tr, rank, world))
    if not rccl["ok"]:
        raise SystemExit(f"--path finetune --gpus {world}: the communicator does not span the ranks it should: {rccl}")
    B = a.finetune_batch
    lr = 1e-4
    towers = None
    if a.with_towers:
        # frames in: the frozen CLIP ViT-B/16 towers (random init) produce the per-block CLS / EOT features every step;
        from arp_amd import clip, synth
        ccfg = clip.MODELS["ViT-B/16"]
        towers = clip.ClipLabeller(ccfg, synth.clip_weights(ccfg, seed=0), mode=a.mode, device=local_rank, max_batch=3 * B, fp8_mlp=(2 if a.fp8_attn else 1) if a.fp8_mlp else False)
        frames = np.concatenate([synth.procgen_like_frames(B, seed=200 + k) for k in range(3)])  # image0 | image1 | image2
        tokens = synth.prompt_tokens(B, [8] * B, seed=203)
        rb = FT.synth_batch(cfg, B, seed=100 + rank)
        fbufs = tr.feature_buffers(B)

        def step():
            towers.encode_multiscale_to(frames, tokens, fbufs)  # one 3B-frame image pass + B prompts; features stay in HBM
            tr.set_batch_device(fbufs, rb[4], rb[5])
            tr.train_step_async(lr)
    else:
        tr.set_batch(*FT.synth_batch(cfg, B, seed=100 + rank))

        def step():
            tr.train_step_async(lr)
    for _ in range(a.warmup):
        step()
    tr.sync()
    if dist is not None:
        dist.barrier()
    _ffi.check(_ffi.lib.arp_dev_synchronize())
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    tr.sync()
    _ffi.check(_ffi.lib.arp_dev_synchronize())
    elapsed = time.perf_counter() - t0
    per_rank = gather_rates(dist, world, a.finetune_batch * a.steps, elapsed)
    if dist is not None:
        import torch
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tr.profile(True)
    tr.profile_reset()
    for _ in range(a.steps):
        tr.train_step_async(lr)
    tr.sync()
    prof = tr.profile_read()
    aux = tr.train_step(lr)
    # dominant kernel, HBM-bound either way.  Single-process f16 / bf16 steps apply AdamW from inside the seven weight-gradient GEMMs
    # (arp_ft.hip, GEMM_SITE_ADAMW): the largest of them reads p, m, v and writes p, m, v + the 16-bit mirror of its weight = 26 B per
    # parameter (the operands, <= 192 rows each, are noise).  With ARP_FT_FUSE_ADAM=0, f32 mode or a communicator: the separate AdamW
    # pass -- p, g, m, v read, p, m, v written: 7 x 4 B per parameter.
    dw_sites = {"ft.image_fc2_dW": "image_adapter.layers.3.weight", "ft.text_fc2_dW": "text_adapter.layers.3.weight",
                "ft.image_fc1_dW": "image_adapter.layers.0.weight", "ft.text_fc1_dW": "text_adapter.layers.0.weight",
                "ft.image_inter_dW": "image_intermediate_linear.weight", "ft.text_inter_dW": "text_intermediate_linear.weight",
                "ft.inverse_fc1_dW": "inverse_layer.layers.0.weight"}
    adamw_ms = prof["ft.adamw"]["ms"] / max(prof["ft.adamw"]["calls"], 1)
    fused = [k for k in dw_sites if k in prof and prof[k]["ms"] / max(prof[k]["calls"], 1) > adamw_ms]
    if fused and a.mode != "f32" and world == 1:
        site = max(fused, key=lambda k: prof[k]["ms"])
        nbytes = 26.0 * float(np.prod(tr.shapes[dw_sites[site]]))
        kern = f"gemm_nt_kernel<GEMM_SITE_ADAMW> @ {site} (weight gradient + AdamW in one launch)"
    else:
        site = "ft.adamw"
        nbytes = 28.0 * tr.n_params
        kern = f"ft_adamw_kernel @ {site}"
    avg_ms = prof[site]["ms"] / max(prof[site]["calls"], 1)
    flops = FT.flops_per_sample(cfg) * a.finetune_batch
    # HBM bytes of the dominant launch from the committed PMC passes (scripts/prof_round.sh -> profiles/pmc_traffic_finetune.json), matched by the weight's size
    traffic = traffic_src = traffic_stale = None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_finetune.json")))
        from arp_amd._srchash import csrc_sha1
        traffic_stale = rec.get("csrc_sha1") != csrc_sha1()
        if site in dw_sites and a.finetune_batch == 64:
            n_el = int(np.prod(tr.shapes[dw_sites[site]]))
            for e in rec["fused_adamw_gemm_by_grid_threads"].values():
                if e["weight_elements"] == n_el:
                    traffic = e["hbm_bytes_per_launch"]
                    traffic_src = f"profiles/pmc_traffic_finetune.json ({rec.get('round')}): committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; NOT measured in this run"
    except (OSError, ValueError, KeyError):
        traffic = None
    if rank != 0:
        tr.close()
        dist.destroy_process_group()
        return
    emit_line(({
        "metric": ("samples/sec CLIP multi-scale adapter fine-tune step (frames in: frozen ViT-B/16 towers + head)" if towers is not None else
                   "samples/sec CLIP multi-scale adapter fine-tune step (head; frozen-tower features in)") + (" -- STAGED data-parallel ordering on one rank" if staged else ""), "value": world * a.finetune_batch * a.steps / elapsed,
        "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": a.mode, "data": "synthetic",
        "config": {"towers": None if towers is None else ("ViT-B/16, " + a.mode + ((" + fp8 (e4m3) c_fc / c_proj" + (" / in_proj / out_proj" if a.fp8_attn else "")) if a.fp8_mlp else "")),
                   "workload": f"CLIPMultiscaleAdapter head train step, {a.finetune_batch} samples x 3 frames, ViT-B/16-shaped tower features "
                               f"[3,B,9216]+[3,B,512] / [B,6144]+[B,512] resident in HBM, {tr.n_params / 1e6:.0f} M trainable params "
                               f"(BASELINE.json configs[4])", "parallelism": "single GPU (as the reference)" if world == 1 else f"dp{world}: one RCCL all-reduce(sum) of the "
                               f"flat f32 gradient ({tr.n_params * 4 / 1e9:.1f} GB) per step"},
        "roofline": {"bound": "hbm", "achieved": nbytes / (avg_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": nbytes / (avg_ms * 1e-3) / 1e9 / 8000.0, "traffic": traffic, "traffic_stale": None if traffic is None else traffic_stale, "traffic_source": traffic_src, "kernel": kern,
                     "bytes_per_launch": nbytes, "avg_launch_ms": avg_ms},
        "whole_step": {"gflop_per_step": flops / 1e9, "tflops": flops / (elapsed / a.steps) / 1e12},
        "cpu_baseline": cpu, "final_aux": aux, "per_rank_samples_per_s": [round(v, 1) for v in per_rank], "rccl": rccl,
        "staged": None if not staged else {
            "staged_ms_per_step": round(elapsed / a.steps * 1e3, 4),
            "ordering": "seven buckets in production order from inside the backward on the communication stream; AdamW as a separate pass behind the last one (the fused dW + AdamW epilogue needs the SUMMED gradient and is off with a communicator)",
            "allreduce_ms": [round(prof[f"ft.allreduce_b{i}"]["ms"] / max(prof[f"ft.allreduce_b{i}"]["calls"], 1), 4) if f"ft.allreduce_b{i}" in prof else None for i in range(7)],
            "bucket_bytes": [4 * sum(hi - lo for lo, hi in b) for b in FT.bucket_plan(cfg)[0]], "gradient_bytes": 4 * FT.bucket_plan(cfg)[1],
            "reduction": "ncclAllReduce(sum) over ONE rank (identity)"},
        "sites_ms_per_step": {k: round(v["ms"] / a.steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}))
    tr.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="frames per GPU per step")
    ap.add_argument("--model", default="ViT-B/32", choices=["ViT-B/32", "ViT-B/16"])
    ap.add_argument("--mode", default=None, choices=["bf16", "f16", "f32"],
                    help="GEMM operand type; default f16 for the label path (same MFMA rate as bf16, rewards within 1e-4 of the fp32 "
                         "reference -- bf16 gives 3-8e-4), bf16 for the policy / finetune paths")
    ap.add_argument("--no-alt-bf16", dest="alt_bf16", action="store_false",
                    help="label path, f16 mode: skip the bf16-operand parity check reported as alt_dtype")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="0 disables the cpu_baseline leg")
    ap.add_argument("--parity-frames", type=int, default=8, help="frames checked against the oracle before timing (rank 0)")
    ap.add_argument("--streams", type=int, default=2, help="label path: N = each batch is labelled in N contiguous parts on N HIP streams of "
                    "the same GPU (one half's LayerNorm/attention/GEMM tails overlap the other half's GEMMs); 1 = single stream")
    ap.add_argument("--h5-rows", type=int, default=4096, help="--path h5: rows (= labelled frames) of the generated demonstration file")
    ap.add_argument("--path", default="label", choices=["label", "policy", "finetune", "online", "h5"],
                    help="label = headline metric (BASELINE.json configs[1]); policy = ARPDT train_step (configs[3], secondary); "
                         "finetune = CLIP multi-scale adapter head step (configs[4], secondary); online = latency of one single-frame reward / one "
                         "greedy action (the rollout loop, SURVEY row N4; secondary)")
    ap.add_argument("--finetune-batch", type=int, default=64, help="samples per step (finetune.py:25)")
    ap.add_argument("--fp8-mlp", action="store_true", help="label path / finetune --with-towers: the vision tower's c_fc / c_proj GEMMs on e4m3 "
                    "operands (scaled fp8 MFMA; BASELINE configs[4] 'fp8 MFMA GEMMs'): a lower-precision throughput mode, its parity is printed")
    ap.add_argument("--fp8-attn", action="store_true", help="with --fp8-mlp: the attention's in_proj / out_proj on e4m3 operands too (ViT-B/16 towers)")
    ap.add_argument("--with-towers", action="store_true", help="finetune path: run the frozen CLIP ViT-B/16 towers inside the timed step "
                    "(uint8 frames + tokens in) instead of feeding pre-computed tower features")
    ap.add_argument("--policy-batch", type=int, default=32, help="samples per GPU per step (256 / 8 in configs[3])")
    ap.add_argument("--encoder-mode", default=None, choices=["bf16", "f16", "f32", "f16x3", "f16c"], help="policy path with --with-encoder: operand mode of the frozen encoder "
                    "(default: --mode).  f16x3 = (hi, lo) binary16 operand pairs, three 16-bit MFMAs per product, f32 attention: f32-level error")
    ap.add_argument("--adapter-c", action="store_true", help="policy path, f16: the adapter's forward products corrected on the fp4 MFMA (the default since round 6, except behind a plain 16-bit encoder)")
    ap.add_argument("--no-adapter-c", action="store_true", help="policy path, f16: the plain binary16 adapter products (rounds 2-5's default: 0.76 ms per step, logits 8.7e-4 over 16 seeds, 1.18e-3 behind encoder outputs)")
    ap.add_argument("--staged", action="store_true", help="policy / finetune path at 1 GPU: the data-parallel ORDERING of the step -- backward staged for the bucketed all-reduce "
                    "on the communication stream, through a one-rank RCCL communicator (an identity reduction; ARP_DT_FORCE_COMM / ARP_FT_FORCE_COMM with the overlap on, "
                    "as tests/test_policy_gpu.py::test_bucketed_overlapped_allreduce_equals_serial runs it): the per-rank compute time configs[3] / configs[4] will "
                    "reproduce on 8 GPUs before any communication time (VERDICT r5 weak #7)")
    ap.add_argument("--single-slot", action="store_true", help="policy path with --with-encoder: rounds 1-5's flow -- one batch of frames staged once, every step encodes at its head")
    ap.add_argument("--churn", default="", help="diagnostic: throw-away work before the timed objects are created, comma-separated: malloc, h2d, trainer, encoder, fwd_enc, encfwd, full")
    ap.add_argument("--no-encode-ahead", action="store_true", help="policy path with --with-encoder: every step encodes its own batch at its head (rounds 1-5) instead of the "
                    "frozen encoder's pass for batch i + 1 running beside step i's policy part")
    ap.add_argument("--with-encoder", action="store_true", help="policy path: run the frozen M3AE ViT-B/16 encoder inside the step "
                    "(frames in, the reference's own boundary; SURVEY row N1) instead of feeding pre-computed encodings")
    ap.add_argument("--all-secondary", dest="all_secondary", action="store_true", default=True,
                    help="label path, 1 GPU (default ON): after the headline line is measured, run the policy / policy --with-encoder / finetune / "
                         "ViT-B/16 benches as child processes and append their JSON lines under `extra`, so that one driver run carries them")
    ap.add_argument("--no-secondary", dest="all_secondary", action="store_false", help="headline line only")
    ap.add_argument("--timed-only", action="store_true", help="label path: only the warm-up, the timed steps and the per-launch profiled repeat of the "
                    "same steps -- no seam calls, no isolated single-stream pass, no bf16 parity handle -- so that a rocprofv3 --stats run of this "
                    "command averages exactly the launches `roofline.avg_launch_ms` averages (scripts/prof_label.sh)")
    a = ap.parse_args()
    if a.mode is None:
        a.mode = "f16"  # IEEE-half MFMA operands on every path: the 16-bit mode that meets the parity bars (bf16 stays selectable)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(a.gpus)  # plain `python bench.py --gpus N`: this process only launches the ranks (it never touches a GPU)
    if a.path == "policy":
        return bench_policy(a)
    if a.path == "finetune":
        return bench_finetune(a)
    if a.path == "online":
        return bench_online(a)
    if a.path == "h5":
        return bench_h5(a)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    a.gpus = world  # under torch.distributed.run / spawn_ranks the environment is authoritative

    cpu = None
    if rank == 0 and world == 1 and a.cpu_seconds > 0:
        cpu = cpu_baseline(a.model, a.cpu_seconds)
    # The secondary benches run as child processes BEFORE this process touches the GPU (a process that has initialised the GPU must
    # not fork + exec another program on this pool), one after the other, each alone on the device; the headline bench follows.
    extra = None
    if rank == 0 and world == 1 and a.all_secondary and a.model == "ViT-B/32" and not a.fp8_mlp and a.batch == 1024 and a.mode == "f16":
        extra = run_secondary(a)

    dist = None
    if world > 1:
        # control plane only (barrier + max of a scalar): gloo on CPU tensors.  The data path has no
        # collective -- each rank labels its own shard of the frame stream.
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from arp_amd import _ffi, clip, synth
    _ffi.require_gpu()
    local_rank %= _ffi.device_count()  # one rank per GPU; on a box with fewer GPUs than ranks they share (test rigs only)
    _ffi.check(_ffi.lib.arp_set_device(local_rank))
    cfg = clip.MODELS[a.model]
    weights = synth.clip_weights(cfg, seed=0)
    tokens = synth.prompt_tokens(1, 8, seed=2)
    model = clip.ClipLabeller(cfg, weights, mode=a.mode, device=local_rank, max_batch=a.batch, n_streams=a.streams, fp8_mlp=(2 if a.fp8_attn else 1) if a.fp8_mlp else False).set_text(tokens)

    # parity gate on a few frames (rank 0): the thing timed below is the thing checked here
    parity = None
    if rank == 0 and a.parity_frames > 0:
        from oracle import clip_np
        ocfg = clip_np.ClipConfig(patch=cfg.patch)
        fr = synth.procgen_like_frames(a.parity_frames, seed=1)
        ref = clip_np.compute_reward(weights, ocfg, fr, tokens)
        got = model.label(fr)
        parity = float(np.abs(got - ref).max() / np.exp(float(weights["logit_scale"])))
    H = W = 256
    frames = synth.noise_frames(a.batch, H, W, seed=1000 + rank)
    d_frames = clip.DeviceBuffer(frames.nbytes).upload(frames)
    d_rewards = clip.DeviceBuffer(a.batch * 4)

    def barrier():
        if dist is not None:
            dist.barrier()

    def step():
        model.label_device_async(d_frames, a.batch, H, W, d_rewards)

    for _ in range(a.warmup):
        step()
    model.sync()
    _ffi.check(_ffi.lib.arp_dev_synchronize())

    # ---- timed region: EXACTLY `steps` steps, barrier + device sync on both sides -------------------
    e0, e1 = clip.Event(), clip.Event()
    barrier()
    _ffi.check(_ffi.lib.arp_dev_synchronize())
    t0 = time.perf_counter()
    model.record(e0)
    for _ in range(a.steps):
        step()
    model.record(e1)
    model.sync()
    _ffi.check(_ffi.lib.arp_dev_synchronize())
    t1 = time.perf_counter()
    barrier()
    elapsed = t1 - t0
    ev_ms = clip.elapsed_ms(e0, e1)
    per_rank = gather_rates(dist, world, a.batch * a.steps, elapsed)
    ranks_seen = label_ranks_seen(dist, rank, world, local_rank)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rewards = d_rewards.download(np.float32, a.batch)
    assert np.isfinite(rewards).all(), "non-finite rewards"

    # ---- the S2 seam as the reference calls it (label_reward.py:132-146: host frames in, host rewards out), OUTSIDE the timed
    # region: arp_clip_label incl. the H2D upload over PCIe and the D2H of the rewards.  Never `value`.
    seam = None
    if rank == 0 and world == 1 and not a.timed_only:
        model.label(frames)
        ts = time.perf_counter()
        for _ in range(5):
            r_host = model.label(frames)
        seam_s = (time.perf_counter() - ts) / 5
        model.pin_host(frames)  # the same call on a buffer the caller has registered once (what arp_amd's own HDF5 reader does)
        model.label(frames)
        ts = time.perf_counter()
        for _ in range(5):
            model.label(frames)
        pin_s = (time.perf_counter() - ts) / 5
        model.unpin_host(frames)
        # a STREAM of such calls through the asynchronous pair (one call in flight while the next is submitted: what label_store does)
        # (both slots warmed first: their buffers are allocated on first use.  The window holds n_pipe WHOLE calls -- the first call's
        #  exposed upload included -- and is divided by n_pipe; round 3's first version timed nine calls' work and divided by eight.)
        for k in (0, 1):
            model.label_submit(k, frames)
            model.label_collect(k)
        n_pipe = 16
        ts = time.perf_counter()
        model.label_submit(0, frames)
        for k in range(1, n_pipe):
            model.label_submit(k & 1, frames)
            r_pipe = model.label_collect((k - 1) & 1)
        r_pipe = model.label_collect((n_pipe - 1) & 1)
        pipe_s = (time.perf_counter() - ts) / n_pipe
        seam = {"frames_per_s": a.batch / seam_s, "ms_per_call": seam_s * 1e3, "frames_per_call": a.batch,
                "pinned_frames_per_s": a.batch / pin_s, "pinned_ms_per_call": pin_s * 1e3,
                "pipelined_frames_per_s": a.batch / pipe_s, "pipelined_ms_per_call": pipe_s * 1e3,
                "pipelined_call": "16 calls through arp_clip_label_submit / arp_clip_label_collect, two in flight, the first upload exposed; bit-identical: " + str(bool(np.array_equal(r_pipe, rewards))),
                "call": "arp_clip_label(host uint8 frames [n,256,256,3] -> host float32 rewards): upload over PCIe, label, download",
                "bit_identical_to_hbm_resident": bool(np.array_equal(r_host, rewards))}

    # ---- per-launch kernel timing: the same steps again with HIP events around every launch ---------
    model.profile(True)
    model.profile_reset()
    e2, e3 = clip.Event(), clip.Event()
    model.record(e2)
    for _ in range(a.steps):
        step()
    model.record(e3)
    model.sync()
    prof = model.profile_read()
    prof_ms = clip.elapsed_ms(e2, e3)
    # ---- the clock the chip held: the same steps once more with c_fc on the clock-diagnostic instance of its kernel (gemm256.h CLK: s_memtime / s_memrealtime
    # around every workgroup; the timed instances execute no stamp) -- "0.36 of 2.5 PF" is 0.36 of a peak priced at 2.4 GHz
    clock = None
    if not a.timed_only:  # (a rocprofv3 --stats run of --timed-only averages exactly the timed launches)
        model.profile(False)
        model.clock_probe(True)
        for _ in range(2):
            step()
        model.sync()
        model.clock_probe(True)  # (zeroes the accumulators: the two steps above brought the chip back under load)
        e4, e5 = clip.Event(), clip.Event()
        model.record(e4)
        for _ in range(a.steps):
            step()
        model.record(e5)
        model.sync()
        clock = model.clock_read()
        clock["ms_per_step_while_probed"] = clip.elapsed_ms(e4, e5) / a.steps
        model.clock_probe(False)
        model.profile(True)
    # isolated per-kernel figures: the same steps on ONE stream (no other kernel shares the chip with a launch)
    iso = None
    nsplit = max(1, min(a.streams, a.batch // 128))  # parts a batch is labelled in, one HIP stream each (label_dev in arp_clip.hip)
    if nsplit > 1 and not a.timed_only:
        model.set_streams(1)
        step()
        model.sync()
        model.profile_reset()
        for _ in range(max(a.steps // 2, 2)):
            step()
        model.sync()
        iso = model.profile_read()
        model.set_streams(a.streams)
    model.profile(False)

    # bf16 operands on the same frames (BASELINE.json's configs[1] names bf16): parity beside the f16 line.  Its RATE is the f16
    # rate (same kernels on v_mfma_f32_16x16x32_bf16; profiles/r1_bench_label_bf16.json = `bench.py --mode bf16` on the same box);
    # it is not re-timed here (a second handle's streams shared hardware queues with the first's under the runtime's default of 4
    # queues and ran at the single-stream rate, whichever mode came second; arp_amd._ffi now asks for 8).
    alt = None
    if rank == 0 and world == 1 and a.mode == "f16" and a.alt_bf16 and parity is not None and not a.timed_only:
        m2 = clip.ClipLabeller(cfg, weights, mode="bf16", device=local_rank, max_batch=a.batch, n_streams=1).set_text(tokens)
        err2 = float(np.abs(m2.label(fr) - ref).max() / np.exp(float(weights["logit_scale"])))
        m2.close()
        alt = {"dtype": "bf16", "max_cosine_err_vs_oracle": err2, "within_tolerance": bool(err2 < 1e-4),
               "note": "parity of the same kernels on bf16 operands (rate: run bench.py --mode bf16); outside north_star's 1e-4, hence not the headline mode"}
    del weights

    if rank == 0:
        sites = {k: v * -(-a.batch // nsplit) / a.batch for k, v in gemm_sites(cfg, a.batch).items()}
        dom = max(sites, key=lambda s: prof.get(s, {"ms": 0})["ms"])
        avg_ms = prof[dom]["ms"] / max(prof[dom]["calls"], 1)
        achieved = sites[dom] / (avg_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[a.mode]
        traffic = None
        mfma_util = None
        traffic_stale = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch of the dominant kernel from committed rocprofv3 --pmc passes
            try:
                rec = json.load(open(tpath)).get(dom, {})
                traffic = rec.get("hbm_bytes_per_launch")
                from arp_amd._srchash import csrc_sha1
                traffic_stale = rec.get("csrc_sha1") != csrc_sha1()  # the committed counters were taken on other kernel sources than this tree's
                mfma_util = rec.get("mfma_util_pct")  # rocprofv3 --pmc MfmaUtil of the same kernel, measured alone (committed run)
                # the committed PMC run may have used a different launch size: algorithmic and measured bytes scale with the frames
                if traffic and rec.get("frames_per_launch"):
                    traffic = traffic * (-(-a.batch // nsplit)) / rec["frames_per_launch"]
            except Exception:
                traffic = None
        fps = world * a.batch * a.steps / elapsed
        flops_frame = clip.flops_per_frame(cfg)
        exec_flops = flops_frame - (0 if os.environ.get("ARP_CLS_ONLY") == "0" else 2.0 * (cfg.tokens - 1) * 9 * cfg.width * cfg.width)
        kname = {"vit.qkv_attn": "qkv_attn_kernel"}.get(dom, "gemm2w_kernel" if os.environ.get("ARP_GEMM") == "3" else "gemm256_nt_kernel")
        total_ms = sum(v["ms"] for v in prof.values())
        out = {
            "metric": "frames/sec CLIP reward-labelled (256x256 ViT-B/32)" if a.model == "ViT-B/32" else f"frames/sec CLIP reward-labelled (256x256 {a.model})",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": a.mode,
            "data": "synthetic",
            # the secondary lines of this run in brief (the full lines are under `extra` at the end; a log tail may cut them off)
            "extra_summary": summarize_extra(extra, seam),
            "config": {"workload": f"CLIP {a.model} reward labelling, batch {a.batch} synthetic 256x256x3 uint8 frames per GPU resident in HBM "
                                   f"(BASELINE.json configs[1]), random-init weights, text tower cached", "frames_per_gpu_per_step": a.batch,
                       "operands": {"f16": "IEEE half MFMA operands (same 2.5 PF dense rate as bf16), f32 accumulate / residual / LayerNorm / softmax; "
                                           "rewards within north_star's 1e-4 of the fp32 oracle -- bf16 operands are not (alt_dtype)",
                                    "bf16": "bf16 MFMA operands, f32 accumulate / residual / LayerNorm / softmax",
                                    "f32": "f32-input MFMA"}[a.mode] + (" -- PLUS c_fc / c_proj of blocks 1..11 on e4m3 operands (--fp8-mlp: NOT the "
                                    "headline configuration; see parity)" if a.fp8_mlp else ""),
                       "parallelism": f"shard{world} (no collective)", "streams_per_gpu": nsplit},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_stale": traffic_stale, "mfma_util_pct": mfma_util,
                         "traffic_source": None if traffic is None else "profiles/pmc_traffic.json: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel, scaled to this launch's frames; NOT measured in this run",
                         "mfma_util_source": None if mfma_util is None else "profiles/pmc_traffic.json: committed rocprofv3 --pmc MfmaUtil pass (kernel alone on the chip); NOT measured in this run",
                         "kernel": f"{kname} @ {dom}",
                         "note": (f"each launch covers {-(-a.batch // nsplit)} frames; with --streams {nsplit} that many such launches (the parts of a batch) share "
                                  "the chip, so a launch's HIP-event duration includes time it ran beside the other stream's kernels; "
                                  "--streams 1 gives the isolated per-kernel figure") if nsplit > 1 else "single stream", "flops_per_launch": sites[dom],
                         "avg_launch_ms": avg_ms, "launches": prof[dom]["calls"]},
            "roofline_isolated": (None if iso is None else {
                "achieved": gemm_sites(cfg, a.batch)[dom] / (iso[dom]["ms"] / iso[dom]["calls"] * 1e-3) / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": gemm_sites(cfg, a.batch)[dom] / (iso[dom]["ms"] / iso[dom]["calls"] * 1e-3) / 1e12 / peak,
                "avg_launch_ms": iso[dom]["ms"] / iso[dom]["calls"], "flops_per_launch": gemm_sites(cfg, a.batch)[dom],
                "note": f"same kernel @ {dom}, whole {a.batch}-frame batch per launch on a single stream (nothing else resident)"}),
            "cpu_baseline": cpu,
            "seam": seam,
            "alt_dtype": alt,
            # mfma_frac_of_peak counts the FLOPs actually ISSUED (the last block runs out_proj + MLP on the class-token row only,
            # tower.h); the nominal figure prices the pass at SURVEY section 8(d)'s 8.82 GFLOP/frame including that skipped work
            "whole_pass": {"executed_gflop_per_frame": exec_flops / 1e9, "mfma_frac_of_peak": fps / world * exec_flops / (peak * 1e12),
                           "nominal_gflop_per_frame": flops_frame / 1e9, "nominal_mfma_frac_of_peak": fps / world * flops_frame / (peak * 1e12),
                           "hip_event_ms_per_step": ev_ms / a.steps, "profiled_ms_per_step": prof_ms / a.steps,
                           # the peak is priced at the chip's 2.4 GHz; at the clock it HELD under this pass (c_fc's workgroups, time-weighted) the same FLOPs are
                           # this fraction of what the matrix pipes could have issued
                           "clock_ghz": None if not clock else clock["clock_ghz"], "clock_probe_workgroups": None if not clock else clock["workgroups"],
                           "clock_probe_ms_per_step": None if not clock else clock["ms_per_step_while_probed"],
                           "mfma_frac_of_peak_at_held_clock": None if not clock or not clock["clock_ghz"] else fps / world * flops_frame / (peak * 1e12) * 2.4 / clock["clock_ghz"]},
            "parity": {"max_cosine_err_vs_oracle": parity, "frames": a.parity_frames, "tolerance": 1e-4,
                       "within_tolerance": None if parity is None else bool(parity < 1e-4)},
            "sites_ms_per_step": {k: round(v["ms"] / a.steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
            "sites_total_ms_per_step": total_ms / a.steps,
            "per_rank_frames_per_s": [round(v, 1) for v in per_rank],
            "ranks_seen": ranks_seen,
        }
        if os.environ.get("ARP_BENCH_CHILD"):
            emit(json.dumps(out))
        else:
            emit_report(out, extra)
    model.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
