#!/usr/bin/env python
"""Filter-gradient kernels on the shapes of one res101+FPN 1000x600 training step: the register-staged kernel with its
reduction / accumulation kernels (variant 1) against the LDS-DMA kernel with the fused epilogue (variant 2), each with its
own tuned plan, in the accumulating form the captured step uses (frcnn_conv2d_bwd_weight_acc).

    python tools/wgrad_bench.py [--reps 20] [--out gpurun_out/wgrad.md]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (name, n, h, w, c, k, r, stride, pad, calls per step)
SHAPES = [
    ("l2 conv1 1x1/2 256-128", 1, 150, 250, 256, 128, 1, 2, 0, 1),
    ("l2 conv2 3x3 128", 1, 75, 125, 128, 128, 3, 1, 1, 4),
    ("l2 conv3 1x1 128-512", 1, 75, 125, 128, 512, 1, 1, 0, 4),
    ("l2 conv1 1x1 512-128", 1, 75, 125, 512, 128, 1, 1, 0, 3),
    ("l2 down 1x1/2 256-512", 1, 150, 250, 256, 512, 1, 2, 0, 1),
    ("l3 conv2 3x3 256", 1, 38, 63, 256, 256, 3, 1, 1, 23),
    ("l3 conv3 1x1 256-1024", 1, 38, 63, 256, 1024, 1, 1, 0, 23),
    ("l3 conv1 1x1 1024-256", 1, 38, 63, 1024, 256, 1, 1, 0, 22),
    ("l3 down 1x1/2 512-1024", 1, 75, 125, 512, 1024, 1, 2, 0, 1),
    ("l4 conv2 3x3 512", 1, 19, 32, 512, 512, 3, 1, 1, 3),
    ("l4 conv3 1x1 512-2048", 1, 19, 32, 512, 2048, 1, 1, 0, 3),
    ("l4 conv1 1x1 2048-512", 1, 19, 32, 2048, 512, 1, 1, 0, 2),
    ("fpn lateral p2 1x1 256", 1, 150, 250, 256, 256, 1, 1, 0, 1),
    ("fpn output p2 3x3 256", 1, 150, 250, 256, 256, 3, 1, 1, 1),
    ("fpn output p3 3x3 256", 1, 75, 125, 256, 256, 3, 1, 1, 1),
    ("rpn 3x3 p2 256-256", 1, 150, 250, 256, 256, 3, 1, 1, 1),
    ("tail fc 12544-1024", 256, 1, 1, 12544, 1024, 1, 1, 0, 1),
    ("tail fc 1024-1024", 256, 1, 1, 1024, 1024, 1, 1, 0, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--forced", default="", help="kernel:splits,... e.g. 1:1,3:1,2:8,4:8 - time forced plans instead of the tuned variants")
    args = ap.parse_args()
    from faster_rcnn_pytorch_multimodal_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    rows, tot = [], {1: 0.0, 2: 0.0}
    for name, n, h, w, c, k, r, stride, pad, calls in SHAPES:
        if args.only and args.only not in name:
            continue
        x = torch.randn((n, h, w, c), generator=g).to(dev)
        ho, wo = ops.conv_out_hw(h, w, r, r, stride, pad)
        dy = torch.randn((n, ho, wo, k), generator=g).to(dev)
        grad = torch.zeros(k, c, r, r, device=dev)
        us = {}
        if args.forced:
            cells = []
            for item in args.forced.split(","):
                kern, sp = [int(t) for t in item.split(":")]
                ops.set_wgrad_plan(kern, sp)
                for _ in range(2):
                    ops.conv2d_bwd_weight_acc(x, dy, r, r, grad, None, stride=stride, pad=pad)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.reps):
                    ops.conv2d_bwd_weight_acc(x, dy, r, r, grad, None, stride=stride, pad=pad)
                e1.record()
                torch.cuda.synchronize()
                cells.append("%s %.1f" % (item, 1e3 * e0.elapsed_time(e1) / args.reps))
            ops.set_wgrad_plan(0)
            print("%-26s %s" % (name, " | ".join(cells)), flush=True)
            continue
        for variant in (1, 2):
            ops.set_wgrad_variant(variant)
            ops.set_conv_autotune(True)
            try:
                ops.conv2d_bwd_weight(x, dy, r, r, stride=stride, pad=pad)
            finally:
                torch.cuda.synchronize()
                ops.set_conv_autotune(False)
            for _ in range(2):
                ops.conv2d_bwd_weight_acc(x, dy, r, r, grad, None, stride=stride, pad=pad)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.reps):
                ops.conv2d_bwd_weight_acc(x, dy, r, r, grad, None, stride=stride, pad=pad)
            e1.record()
            torch.cuda.synchronize()
            us[variant] = 1e3 * e0.elapsed_time(e1) / args.reps
            tot[variant] += us[variant] * calls
        fl = 2.0 * n * ho * wo * k * r * r * c
        rows.append("| %s | %d | %.1f | %.1f | %.1f | %.1f | %.2f |" % (name, calls, us[1], fl / us[1] / 1e6, us[2], fl / us[2] / 1e6,
                                                                       us[1] / us[2]))
        print(rows[-1], flush=True)
    ops.set_wgrad_variant(0)
    text = ["# Filter gradient per shape: register-staged + separate kernels (variant 1) vs LDS-DMA with fused epilogue (variant 2)", "",
            "`python tools/wgrad_bench.py --reps %d`; accumulating form (`frcnn_conv2d_bwd_weight_acc`), tuned plan per variant, "
            "back-to-back eager launches on one stream (host-paced below ~15 us)." % args.reps, "",
            "| shape | calls/step | v1 us | v1 TF/s | v2 us | v2 TF/s | v1/v2 |", "|---|---:|---:|---:|---:|---:|---:|"] + rows + [
            "", "sum over calls: variant 1 %.0f us, variant 2 %.0f us per step" % (tot[1], tot[2])]
    print(text[-1])
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(text) + "\n")


if __name__ == "__main__":
    main()
