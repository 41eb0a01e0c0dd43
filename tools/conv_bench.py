#!/usr/bin/env python
"""Micro-benchmark of frcnn_conv2d_fwd on the convolution shapes of one res101 1000x600 frame
(tuning aid; bench.py remains the judged measurement).

    python tools/conv_bench.py [--reps 20] [--tile TM,TN] [--split S] [--only substr]

Each shape is launched `reps` times back to back between two HIP events on the launch stream.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faster_rcnn_pytorch_multimodal_amd.model.frame_graph import capture  # noqa: E402

# (name, n, h, w, c, k, r, stride, pad, residual, calls per frame)
SHAPES = [
    ("stem 7x7/2", 1, 600, 1000, 4, 64, 7, 2, 3, False, 1),
    ("l1 1x1 64-64", 1, 150, 250, 64, 64, 1, 1, 0, False, 1),
    ("l1 3x3 64", 1, 150, 250, 64, 64, 3, 1, 1, False, 3),
    ("l1 1x1 64-256 +res", 1, 150, 250, 64, 256, 1, 1, 0, True, 4),
    ("l1 1x1 256-64", 1, 150, 250, 256, 64, 1, 1, 0, False, 2),
    ("l2 1x1/2 256-128", 1, 150, 250, 256, 128, 1, 2, 0, False, 1),
    ("l2 ds 1x1/2 256-512", 1, 150, 250, 256, 512, 1, 2, 0, False, 1),
    ("l2 3x3 128", 1, 75, 125, 128, 128, 3, 1, 1, False, 4),
    ("l2 1x1 128-512 +res", 1, 75, 125, 128, 512, 1, 1, 0, True, 4),
    ("l2 1x1 512-128", 1, 75, 125, 512, 128, 1, 1, 0, False, 3),
    ("l3 1x1/2 512-256", 1, 75, 125, 512, 256, 1, 2, 0, False, 1),
    ("l3 ds 1x1/2 512-1024", 1, 75, 125, 512, 1024, 1, 2, 0, False, 1),
    ("l3 3x3 256", 1, 38, 63, 256, 256, 3, 1, 1, False, 23),
    ("l3 1x1 256-1024 +res", 1, 38, 63, 256, 1024, 1, 1, 0, True, 23),
    ("l3 1x1 1024-256", 1, 38, 63, 1024, 256, 1, 1, 0, False, 22),
    ("rpn 3x3 1024-512", 1, 38, 63, 1024, 512, 3, 1, 1, False, 1),
    ("rpn 1x1 512-152", 1, 38, 63, 512, 152, 1, 1, 0, False, 1),
    ("l4 1x1 1024-512", 300, 7, 7, 1024, 512, 1, 1, 0, False, 1),
    ("l4 ds 1x1 1024-2048", 300, 7, 7, 1024, 2048, 1, 1, 0, False, 1),
    ("l4 3x3 512", 300, 7, 7, 512, 512, 3, 1, 1, False, 3),
    ("l4 1x1 512-2048 +res", 300, 7, 7, 512, 2048, 1, 1, 0, True, 3),
    ("l4 1x1 2048-512", 300, 7, 7, 2048, 512, 1, 1, 0, False, 2),
]


# trainable convolutions of the res101+FPN train step at 1000x600 (FIXED_BLOCKS = 1)
FPN_TRAIN_SHAPES = [
    ("l2 3x3 128", 1, 75, 125, 128, 128, 3, 1, 1, False, 4),
    ("l2 1x1 128-512", 1, 75, 125, 128, 512, 1, 1, 0, True, 4),
    ("l2 1x1 512-128", 1, 75, 125, 512, 128, 1, 1, 0, False, 3),
    ("l3 3x3 256", 1, 38, 63, 256, 256, 3, 1, 1, False, 23),
    ("l3 1x1 256-1024", 1, 38, 63, 256, 1024, 1, 1, 0, True, 23),
    ("l3 1x1 1024-256", 1, 38, 63, 1024, 256, 1, 1, 0, False, 22),
    ("l4 3x3/2 512", 1, 38, 63, 512, 512, 3, 2, 1, False, 1),
    ("l4 3x3 512", 1, 19, 32, 512, 512, 3, 1, 1, False, 2),
    ("l4 1x1 512-2048", 1, 19, 32, 512, 2048, 1, 1, 0, True, 3),
    ("l4 1x1 2048-512", 1, 19, 32, 2048, 512, 1, 1, 0, False, 2),
    ("fpn lat2 256-256", 1, 150, 250, 256, 256, 1, 1, 0, False, 1),
    ("fpn lat3 512-256", 1, 75, 125, 512, 256, 1, 1, 0, False, 1),
    ("fpn aa2 3x3 256", 1, 150, 250, 256, 256, 3, 1, 1, False, 1),
    ("fpn aa3 3x3 256", 1, 75, 125, 256, 256, 3, 1, 1, False, 1),
    ("rpn 3x3 256-512", 1, 150, 250, 256, 512, 3, 1, 1, False, 1),
    ("rpn 1x1 512-152", 1, 150, 250, 512, 152, 1, 1, 0, False, 1),
    ("t_fc1 12544-2048", 256, 1, 1, 12544, 2048, 1, 1, 0, False, 1),
    ("t_fc2 2048-2048", 256, 1, 1, 2048, 2048, 1, 1, 0, False, 2),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--tile", default="0,0")
    ap.add_argument("--split", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--staging", type=int, default=1, help="1 = LDS-DMA for the 8-wave tiles, 0 = register staging")
    ap.add_argument("--mode", default="fwd", choices=["fwd", "dgrad", "wgrad"])
    ap.add_argument("--fpn", action="store_true", help="use the trainable convolutions of the FPN train step instead")
    ap.add_argument("--batch", type=int, default=1, help="frames per launch (n is multiplied; us/frame is per frame)")
    ap.add_argument("--autotune", action="store_true", help="time every (tile, split-K) candidate first")
    ap.add_argument("--shape", action="append", default=[], help="extra shape n,h,w,c,k,r,stride,pad (replaces the table)")
    ap.add_argument("--algo", type=int, default=0, help="frcnn_conv2d_set_algo: 0 auto, 1 implicit GEMM only, 2 Winograd where it applies")
    ap.add_argument("--nores", action="store_true", help="drop the residual operand (epilogue traffic experiment)")
    ap.add_argument("--graph", action="store_true", help="replay the launches from a hipGraph also with one stream")
    ap.add_argument("--streams", type=int, default=1, help="launch the same convolution on S HIP streams at once (own "
                    "buffers each): aggregate rate of a saturated chip, what the 4-frames-in-flight timed mode sees")
    args = ap.parse_args()
    from faster_rcnn_pytorch_multimodal_amd import _hip, ops
    lib = _hip.load()
    tm, tn = (int(v) for v in args.tile.split(","))
    _hip.check(lib.frcnn_conv2d_set_tile(tm, tn), "set_tile")
    _hip.check(lib.frcnn_conv2d_set_staging(args.staging), "set_staging")
    _hip.check(lib.frcnn_conv2d_set_algo(args.algo), "set_algo")
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(0)
    tot_us = tot_fl = 0.0
    print("%-24s %6s %9s %9s %8s" % ("shape", "calls", "us/call", "TFLOP/s", "us/frame"))
    shapes = FPN_TRAIN_SHAPES if args.fpn else SHAPES
    if args.shape:
        shapes = []
        for sp in args.shape:
            v = [int(t) for t in sp.split(',')]
            shapes.append(("custom " + sp, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], False, 1))
    ops.set_conv_autotune(args.autotune)
    for name, n, h, w, c, k, r, stride, pad, res, calls in shapes:
        if args.only and args.only not in name:
            continue
        n *= args.batch
        x = torch.randn((n, h, w, c), generator=g).to(dev)
        wt = (torch.randn((k, r, r, c), generator=g) * 0.05).to(dev)
        sc = torch.rand((k,), generator=g).to(dev) + 0.5
        sh = torch.randn((k,), generator=g).to(dev)
        ho, wo = ops.conv_out_hw(h, w, r, r, stride, pad)
        rs = torch.randn((n, ho, wo, k), generator=g).to(dev) if (res and not args.nores) else None
        y = torch.empty((n, ho, wo, k), device=dev)
        if args.mode == "fwd":
            run = lambda: ops.conv2d_nhwc(x, wt, sc, sh, rs, stride=stride, pad=pad, relu=True, split_k=args.split, out=y)
        elif args.mode == "dgrad":
            if k % 4:
                continue
            w_t = ops.conv2d_transpose_filter(wt)
            run = lambda: ops.conv2d_bwd_data(y, w_t, tuple(x.shape), stride=stride, pad=pad)
        else:
            if k % 4:
                continue
            run = lambda: ops.conv2d_bwd_weight(x, y, r, r, stride=stride, pad=pad)
        for _ in range(2):
            run()
        if (args.streams > 1 or args.graph) and args.mode == "fwd":
            import time
            sts = [torch.cuda.Stream() for _ in range(args.streams)]
            xs = [x.clone() for _ in sts]
            ys = [torch.empty_like(y) for _ in sts]
            rss = [rs.clone() if rs is not None else None for _ in sts]
            # each stream replays ONE hipGraph of `reps` launches: an eager loop is paced by the host (~20 us per call),
            # which hides what the chip does with four small GEMMs in flight
            graphs = []
            for st, xi, yi, ri in zip(sts, xs, ys, rss):
                gr = torch.cuda.CUDAGraph()
                with capture(gr, stream=st):
                    for _ in range(args.reps):
                        ops.conv2d_nhwc(xi, wt, sc, sh, ri, stride=stride, pad=pad, relu=True, split_k=args.split, out=yi)
                graphs.append(gr)
            for st, gr in zip(sts, graphs):
                with torch.cuda.stream(st):
                    gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for st, gr in zip(sts, graphs):
                with torch.cuda.stream(st):
                    gr.replay()
            torch.cuda.synchronize()
            us = 1e6 * (time.perf_counter() - t0) / (args.reps * args.streams)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / args.reps
        fl = 2.0 * n * ho * wo * k * r * r * c
        tot_us += us * calls / args.batch
        tot_fl += fl * calls / args.batch
        print("%-24s %6d %9.1f %9.1f %8.1f" % (name, calls, us, fl / us / 1e6, us * calls / args.batch))
    print("%-24s %6s %9s %9.1f %8.1f" % ("TOTAL", "", "", tot_fl / tot_us / 1e6, tot_us))


if __name__ == "__main__":
    main()
