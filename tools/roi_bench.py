#!/usr/bin/env python
"""RoIAlign tuning aid: the RoIs of one bench frame (res101, 1000x600), their sampling-grid statistics and the time of
each kernel variant (frcnn_roi_align_set_variant).  python tools/roi_bench.py [--reps 50]"""
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
    ap.add_argument("--variants", default="1,2,3")
    args = ap.parse_args()
    import bench
    from faster_rcnn_pytorch_multimodal_amd import _hip, ops
    from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device
    lib = _hip.load()
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


if __name__ == "__main__":
    main()
