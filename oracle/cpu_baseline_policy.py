"""Oracle leg of bench.py --path policy: time the torch-CPU port of the ARP-DT train step (fp32 autograd +
clip + Adam, oracle/arpdt_torch.py) on this host's cores.  Child process; prints one JSON object."""
import argparse
import json
import os
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--target-seconds", type=float, default=15.0)
    a = ap.parse_args()
    import numpy as np
    import torch

    from arp_amd import synth_policy as S
    from oracle import arpdt_torch as O

    ncpu = os.cpu_count() or 1
    cfg = O.PolicyConfig(lambda_ret=0.01)
    P = {k: torch.from_numpy(v) for k, v in S.policy_params(cfg, seed=0).items()}
    enc, act, rtg = S.policy_batch(cfg, a.batch, seed=100)
    shard = (torch.from_numpy(enc), torch.from_numpy(act).long(), torch.from_numpy(rtg))
    st = O.init_state(P)
    best = None
    for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        st, _ = O.train_step(st, cfg, [shard], lambda t: 5e-4)
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (th, dt)
    torch.set_num_threads(best[0])
    n = int(max(2, min(50, a.target_seconds / max(best[1], 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        st, aux = O.train_step(st, cfg, [shard], lambda t: 5e-4)
    dt = time.perf_counter() - t0
    print(json.dumps({"value": n * a.batch / dt, "unit": "samples/s", "cores": best[0], "kind": "port",
                      "sample": f"{n} train steps of {a.batch} samples (window 4, encodings [B,4,257,768] in), torch-CPU fp32 autograd + "
                                f"clip + Adam port, {best[0]} torch threads (fastest of 8/16/32/64 on a {ncpu}-cpu host), {dt:.1f} s",
                      "final_loss": float(aux["loss"])}))


if __name__ == "__main__":
    main()
