#!/usr/bin/env python3
"""Host link microbenchmark: pinned hipMemcpyAsync H2D, D2H and both at once, by transfer size and number of streams.
Prints one JSON object.  bench.py runs the same measurement (pcie_link()) next to its H2D/D2H-inclusive figure."""
import json
import sys
import time

import torch


def link_rates(dev, sizes_mb=(1, 4, 16, 64, 160), streams=(1, 2), reps=6):
    out = {}
    big = max(sizes_mb) << 20
    h_in = [torch.empty(big, dtype=torch.uint8).pin_memory() for _ in range(2)]
    h_out = [torch.empty(big, dtype=torch.uint8).pin_memory() for _ in range(2)]
    d_in = [torch.empty(big, dtype=torch.uint8, device=dev) for _ in range(2)]
    d_out = [torch.zeros(big, dtype=torch.uint8, device=dev) for _ in range(2)]
    sts = [torch.cuda.Stream(device=dev) for _ in range(4)]

    def run(mode, nbytes, ns):
        # `ns` streams per direction, each moving nbytes / ns per repetition
        per = nbytes // ns
        def issue():
            for s in range(ns):
                if mode in ("h2d", "both"):
                    with torch.cuda.stream(sts[s]):
                        d_in[s][:per].copy_(h_in[s][:per], non_blocking=True)
                if mode in ("d2h", "both"):
                    with torch.cuda.stream(sts[2 + s]):
                        h_out[s][:per].copy_(d_out[s][:per], non_blocking=True)
        issue()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            issue()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        return per * ns / dt / 1e9   # GB/s per direction

    for mb in sizes_mb:
        for ns in streams:
            for mode in ("h2d", "d2h", "both"):
                out["%s_%dMB_%dstream" % (mode, mb, ns)] = round(run(mode, mb << 20, ns), 2)
    return out


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    r = link_rates(dev)
    r["peak_h2d_GBps"] = max(v for k, v in r.items() if k.startswith("h2d"))
    r["peak_d2h_GBps"] = max(v for k, v in r.items() if k.startswith("d2h"))
    r["peak_h2d_with_d2h_GBps"] = max(v for k, v in r.items() if k.startswith("both"))
    json.dump(r, sys.stdout, indent=1)
    print()
