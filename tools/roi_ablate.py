#!/usr/bin/env python
"""Ablation timing of the map-resident RoIAlign kernel (frcnn_roi_align_set_variant(1000 + mask): 1 no map load, 2 no
tables, 4 no stores, 8 no row loop).  Results of masked runs are wrong by construction; this only locates the time."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from faster_rcnn_pytorch_multimodal_amd import _hip, ops
    from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device
    lib = _hip.load()
    net, _ = bench.build_net("cuda:0")
    info = np.array([0, bench.W, 0, bench.H, 0, 0, 1.0], np.float32)
    detect_frame_device(net, torch.from_numpy(bench.synthetic_frame(0)).cuda(), info, bench.THRESH, bench.MAX_DETS, bench.MAX_DETS)
    feat, rois = net._act_summaries["conv"], net._predictions["rois"]
    g = torch.Generator().manual_seed(0)
    wh = torch.rand(300, 2, generator=g) * 288 + 32
    xy = torch.rand(300, 2, generator=g) * (torch.tensor([float(bench.W), float(bench.H)]) - wh - 1)
    typ = torch.cat((torch.zeros(300, 1), xy, xy + wh), 1).cuda().contiguous()

    def timed(fn, reps=30):
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

    lib.frcnn_roi_align_set_variant(5)
    for mask in [int(a) for a in (sys.argv[1:] or "0 1 2 4 8 6 14 15 10 12".split())]:
        lib.frcnn_roi_align_set_variant(1000 + mask)
        print("mask %2d: bench rois %.1f us, typical %.1f us" % (
            mask, timed(lambda: ops.roi_align_nhwc(feat, rois, 7, 1.0 / 16.0, 0)),
            timed(lambda: ops.roi_align_nhwc(feat, typ, 7, 1.0 / 16.0, 0))))
    lib.frcnn_roi_align_set_variant(1000)
    lib.frcnn_roi_align_set_variant(0)


if __name__ == "__main__":
    main()
