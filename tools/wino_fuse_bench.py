#!/usr/bin/env python
"""A/B of the Winograd forms of one 3x3 layer: two-launch input transform vs the transform fused into the 64x64 GEMM's tile
load (frcnn_conv2d_set_algo flags 16 / 32), 30 back-to-back calls replayed as a hipGraph.

    python tools/wino_fuse_bench.py [n h w c k] ...
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faster_rcnn_pytorch_multimodal_amd import ops          # noqa: E402


def timed(fn, reps=30, rounds=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    e0.record()
    for _ in range(rounds):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (reps * rounds)


def main():
    shapes = [(1, 38, 63, 256, 256), (1, 75, 125, 128, 128), (1, 150, 250, 64, 64), (1, 38, 63, 1024, 512), (300, 7, 7, 512, 512)]
    if len(sys.argv) > 5:
        shapes = [tuple(int(v) for v in sys.argv[1:6])]
    for n, h, w, c, k in shapes:
        x = torch.randn(n, h, w, c, device="cuda")
        wt = torch.randn(k, 3, 3, c, device="cuda") / (3.0 * c ** 0.5)
        u = ops.winograd_filter(wt)
        res = {}
        for name, mode in (("implicit GEMM (analytic plan)", 1), ("winograd, two-launch input transform", 2 | 16),
                           ("winograd, input transform fused into the 64x64 GEMM", 2 | 32)):
            ops.set_conv_algo(mode)
            res[name] = timed(lambda: ops.conv2d_nhwc(x, wt, stride=1, pad=1, relu=True, w_winograd=u if mode != 1 else None))
        ops.set_conv_algo(0)
        print("%dx%dx%d c%d k%d: " % (n, h, w, c, k) + ", ".join("%s %.1f us" % kv for kv in res.items()), flush=True)


if __name__ == "__main__":
    main()
