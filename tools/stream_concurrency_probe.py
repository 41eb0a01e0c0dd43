#!/usr/bin/env python
"""Which HIP streams of this process actually run concurrently?  Pairs of streams each get one single-workgroup spin kernel
(torch.cuda._sleep); a pair that shares a hardware queue (or whatever else serialises it) takes twice as long as a pair that
does not.  Prints the matrix for the first N streams torch hands out (GPU_MAX_HW_QUEUES in the environment changes it).

    python tools/stream_concurrency_probe.py [--streams 12] [--warm K]   # K streams are used once before the probe
"""
import argparse
import time

import torch


def pair_time(a, b, cycles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=12)
    ap.add_argument("--cycles", type=int, default=400000)
    args = ap.parse_args()
    torch.zeros(1, device="cuda")
    sts = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(args.streams)]
    for s in sts:
        with torch.cuda.stream(s):
            torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    one = min(pair_time(sts[1], sts[1], args.cycles) for _ in range(3)) / 2
    print("one kernel: %.1f us" % (one * 1e6))
    n = len(sts)
    print("rows/cols: stream 0 = the default stream, 1.. = torch.cuda.Stream() in creation order; S = serialised pair")
    for i in range(n):
        row = []
        for j in range(n):
            if i == j:
                row.append(".")
                continue
            t = min(pair_time(sts[i], sts[j], args.cycles) for _ in range(2))
            row.append("S" if t > 1.6 * one else "-")
        print("%2d %s" % (i, " ".join(row)))


if __name__ == "__main__":
    main()
