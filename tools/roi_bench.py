#!/usr/bin/env python
"""RoIAlign tuning aid: the RoIs of one bench frame (res101, 1000x600), their sampling-grid statistics and the time of
each kernel variant (frcnn_roi_align_set_variant: 1 / 2 generic, 3 / 4 planned with 8 / 4 loads in flight).  python tools/roi_bench.py [--reps 50]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--variants", default="2,3,4")
    ap.add_argument("--typical", action="store_true", help="also time seeded boxes of 32..320 px (what a trained RPN "
                    "proposes) and a plain 60 MB fill (the write floor of the op)")
    ap.add_argument("--heavy", type=int, default=0, help="heavy-RoI threshold (loads per lane, >= 100)")
    args = ap.parse_args()
    import bench
    from faster_rcnn_pytorch_multimodal_amd import _hip, ops
    from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device
    lib = _hip.load()
    if args.heavy:
        lib.frcnn_roi_align_set_variant(args.heavy)
    net, _ = bench.build_net("cuda:0")
    info = np.array([0, bench.W, 0, bench.H, 0, 0, 1.0], np.float32)
    detect_frame_device(net, torch.from_numpy(bench.synthetic_frame(0)).cuda(), info, bench.THRESH, bench.MAX_DETS, bench.MAX_DETS)
    feat, rois = net._act_summaries["conv"], net._predictions["rois"]
    r = rois.cpu().numpy()
    w, h = (r[:, 3] - r[:, 1]) / 16, (r[:, 4] - r[:, 2]) / 16
    gw, gh = np.ceil(np.maximum(w, 1) / 7), np.ceil(np.maximum(h, 1) / 7)
    print("rois %d  feature-map w %.1f+-%.1f h %.1f+-%.1f  grid w %.2f h %.2f  samples/bin %.1f  loads/bin (direct) %.1f"
          % (len(r), w.mean(), w.std(), h.mean(), h.std(), gw.mean(), gh.mean(), (gw * gh).mean(), 4 * (gw * gh).mean()))
    for v in [int(x) for x in args.variants.split(",")]:
        lib.frcnn_roi_align_set_variant(v)
        for _ in range(3):
            ops.roi_align_nhwc(feat, rois, 7, 1.0 / 16.0, 0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            ops.roi_align_nhwc(feat, rois, 7, 1.0 / 16.0, 0)
        e1.record()
        torch.cuda.synchronize()
        print("variant %d: %.1f us" % (v, 1e3 * e0.elapsed_time(e1) / args.reps))
    lib.frcnn_roi_align_set_variant(0)
    if args.typical:
        def timed(fn):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return 1e3 * e0.elapsed_time(e1) / args.reps
        g = torch.Generator().manual_seed(0)
        wh = torch.rand(300, 2, generator=g) * 288 + 32
        xy = torch.rand(300, 2, generator=g) * (torch.tensor([float(bench.W), float(bench.H)]) - wh - 1)
        typ = torch.cat((torch.zeros(300, 1), xy, xy + wh), 1).cuda().contiguous()
        out = torch.empty((300, 7, 7, feat.shape[-1]), device="cuda")
        us_fill = timed(lambda: out.fill_(1.0))
        bytes_ = feat.numel() * 4 + out.numel() * 4 + typ.numel() * 4
        print("60.2 MB fill: %.1f us = %.0f GB/s" % (us_fill, out.numel() * 4 / us_fill / 1e3))
        for v in [0] + [int(x) for x in args.variants.split(",")]:
            lib.frcnn_roi_align_set_variant(v)
            us_typ = timed(lambda: ops.roi_align_nhwc(feat, typ, 7, 1.0 / 16.0, 0))
            print("typical boxes (32..320 px), variant %d: %.1f us = %.0f GB/s (%.1f %% of 8 TB/s)"
                  % (v, us_typ, bytes_ / us_typ / 1e3, bytes_ / us_typ / 1e3 / 80.0))
        lib.frcnn_roi_align_set_variant(0)


if __name__ == "__main__":
    main()
