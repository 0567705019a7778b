"""Oracle leg of bench.py --path finetune: time the torch-CPU port of the fine-tune head step (fp32 autograd + AdamW,
oracle/finetune_torch.py) on this host's cores.  Child process; prints one JSON object."""
import argparse
import json
import os
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--target-seconds", type=float, default=15.0)
    a = ap.parse_args()
    import torch

    from arp_amd import finetune as FT
    from oracle import finetune_torch as O

    ncpu = os.cpu_count() or 1
    cfg = O.HeadConfig()
    fcfg = FT.FinetuneConfig()
    P = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True) for k, v in FT.synth_params(fcfg, seed=0).items()}
    b = FT.synth_batch(fcfg, a.batch, seed=100)
    tb = [torch.from_numpy(x) for x in b[:5]] + [torch.from_numpy(b[5]).long()]
    opt = torch.optim.AdamW(list(P.values()), lr=1e-4, weight_decay=0.001)

    def step():
        opt.zero_grad()
        out = O.forward(P, cfg, *tb)
        out["loss"].backward()
        opt.step()
        return float(out["loss"].detach())

    best = None
    for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        step()
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (th, dt)
    torch.set_num_threads(best[0])
    n = int(max(2, min(30, a.target_seconds / max(best[1], 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n):
        loss = step()
    dt = time.perf_counter() - t0
    print(json.dumps({"value": n * a.batch / dt, "unit": "samples/s", "cores": best[0], "kind": "port",
                      "sample": f"{n} fine-tune head steps of {a.batch} samples (tower features in, 476 M trainable parameters), torch-CPU fp32 "
                                f"autograd + AdamW port, {best[0]} torch threads (fastest of 8/16/32/64 on a {ncpu}-cpu host), {dt:.1f} s",
                      "final_loss": loss}))


if __name__ == "__main__":
    main()
