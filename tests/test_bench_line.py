"""The bench line the driver parses: the LAST stdout line is the headline object alone and at most 4 KB; the secondary benches are
their own compact lines before it (a 23.6 KB line left BENCH_r04.json.parsed null).  No GPU: bench.py's formatting
only, fed with the round-4 objects committed under profiles/ and with a padded worst case."""
import json
import os
import subprocess
import sys

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"]


def r4_objects():
    full = json.load(open(os.path.join(ROOT, "profiles", "r4_bench_default.json")))
    extra = full.pop("extra")
    return full, extra


def test_headline_of_the_round4_object_fits_and_keeps_the_contract():
    full, extra = r4_objects()
    head = bench.compact_headline(full)
    line = json.dumps(head)
    assert len(line) <= bench.HEADLINE_LIMIT, len(line)
    for k in CONTRACT + ["roofline", "cpu_baseline", "whole_pass", "parity", "extra_summary"]:
        assert k in head, k
    assert head["value"] == pytest.approx(full["value"], rel=1e-4)
    assert head["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-4)
    assert head["ms_per_step"] * head["steps"] == pytest.approx(full["ms_per_step"] * full["steps"], rel=1e-4)  # the driver's consistency check
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in head["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in head["cpu_baseline"], k
    assert "workload" in head["config"]


def test_every_secondary_line_fits_and_names_its_numbers():
    _, extra = r4_objects()
    for name, d in extra.items():
        c = bench.compact_secondary(name, d)
        assert len(json.dumps(c)) <= bench.SECONDARY_LIMIT, (name, len(json.dumps(c)))
        assert c["secondary"] == name and c["value"] == pytest.approx(d["value"], rel=1e-4)
        if "roofline" in d:
            assert c["roofline"]["frac"] == pytest.approx(d["roofline"]["frac"], rel=1e-4)
        if d.get("cpu_baseline"):
            assert c["cpu_baseline"]["value"] == pytest.approx(d["cpu_baseline"]["value"], rel=1e-4)
    assert bench.compact_secondary("x", {"error": "boom" * 500})["error"].endswith("~")


def test_a_padded_worst_case_still_fits():
    full, extra = r4_objects()
    full["config"]["workload"] *= 8
    full["cpu_baseline"]["sample"] *= 8
    full["roofline"]["note"] = "n" * 5000
    full["per_rank_frames_per_s"] = [97000.123456] * 8
    full["extra_summary"] = bench.summarize_extra({f"secondary_{i}": dict(extra["policy"]) for i in range(16)}, full["seam"])
    line = json.dumps(bench.compact_headline(full))
    assert len(line) <= bench.HEADLINE_LIMIT, len(line)


def test_emit_report_prints_the_headline_last_and_alone(tmp_path):
    """the real function in a child process (it writes to file descriptor 1): secondaries first, each its own line, headline last"""
    code = ("import json, bench; bench.ROOT = %r; full = json.load(open(%r)); extra = full.pop('extra'); bench.emit_report(full, extra)"
            % (str(tmp_path), os.path.join(ROOT, "profiles", "r4_bench_default.json")))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == 9
    objs = [json.loads(ln) for ln in lines]
    assert all("secondary" in o for o in objs[:-1]) and "secondary" not in objs[-1]
    assert objs[-1]["metric"].startswith("frames/sec CLIP reward-labelled") and len(lines[-1]) <= 4096
    assert all(len(ln) <= bench.SECONDARY_LIMIT for ln in lines[:-1])
    tail = r.stdout[-8192:]  # what the driver keeps: the whole headline line and at least the policy + finetune lines before it
    assert tail.endswith(lines[-1] + "\n") and lines[-2] in tail and lines[-3] in tail
    stored = json.load(open(tmp_path / "gpurun_out" / "bench_full.json"))
    assert set(stored["extra"]) == set(o["secondary"] for o in objs[:-1]) and "sites_ms_per_step" in stored


def test_emit_report_refuses_an_oversized_headline(tmp_path):
    code = ("import bench; bench.ROOT = %r; bench.compact_headline = lambda full: {'pad': 'x' * 5000}; bench.emit_report({'value': 1})" % str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and r.stdout == "" and "refusing" in r.stderr


def _cert_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import test_multi_gpu as M
    from arp_amd import train
    tr = M._DryTrainer(None, device=rank)
    tr.comm_init(b"x" * 128, world, rank)
    rccl = bench.gather_cert(dist, world, train.certify_collective(tr, rank, world))
    seen = bench.label_ranks_seen(dist, rank, world, rank)
    # a communicator that spans fewer ranks than the environment says must not certify
    tr.comm_info = lambda: {"nranks": 1, "rank": 0, "device": rank, "rccl_version": 0, "has_comm": True}
    bad = bench.gather_cert(dist, world, train.certify_collective(tr, rank, world))
    q.put((rank, rccl, seen, bad))
    dist.barrier()
    dist.destroy_process_group()


def test_multi_gpu_lines_certify_their_rank_count_on_gloo():
    """`bench.py --gpus N` prints what the communicator says about itself (rccl_nranks, ranks_seen, an all-reduce of
    rank + 1); dry-run on two gloo ranks with the RCCL calls replaced by tests/test_multi_gpu.py's stand-in"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cert_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, rccl, seen, bad in res:
        assert rccl["ok"] and rccl["rccl_nranks"] == [2] and rccl["ranks_seen"] == [0, 1] and rccl["allreduce_selfcheck"] == [3.0, 3.0] and rccl["expected"] == 3.0
        assert seen["ok"] and seen["ranks_seen"] == [0, 1] and seen["control_plane_selfcheck"] == 3.0 and seen["distinct_processes"] == 2
        assert not bad["ok"] and bad["rccl_nranks"] == [1]
    assert res[0][1] == res[1][1]  # the same block on every rank
    line = json.dumps(bench.compact_headline({"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 2, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
                                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
                                              "config": {"workload": "w"}, "roofline": {}, "cpu_baseline": None, "rccl": res[0][1], "ranks_seen": res[0][2]}))
    assert '"rccl_nranks": [2]' in line and len(line) < 4096
