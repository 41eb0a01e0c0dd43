#!/usr/bin/env python
"""Isolated timing of the frame's HBM-bound helper kernels against their byte counts (tuning aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    from faster_rcnn_pytorch_multimodal_amd import ops
    dev = "cuda:0"
    x3 = torch.randn(1, 600, 1000, 3, device=dev)
    us = timed(lambda: ops.pad_channels(x3, 4))
    print("pad_channels 600x1000 3->4: %.1f us, %.0f GB/s" % (us, (x3.numel() * 4 + 600 * 1000 * 16) / us / 1e3))
    c1 = torch.randn(1, 300, 500, 64, device=dev)
    us = timed(lambda: ops.maxpool3x3s2_nhwc(c1))
    print("maxpool 300x500x64: %.1f us, %.0f GB/s" % (us, (c1.numel() * 4 + 150 * 250 * 64 * 4) / us / 1e3))
    t = torch.randn(300, 7, 7, 2048, device=dev)
    us = timed(lambda: ops.spatial_mean(t))
    print("spatial_mean 300x7x7x2048: %.1f us, %.0f GB/s" % (us, (t.numel() * 4 + 300 * 2048 * 4) / us / 1e3))


if __name__ == "__main__":
    main()
