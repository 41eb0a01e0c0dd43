#!/usr/bin/env python
"""Sweep (tile, split_k) for selected conv shapes and compare with the planner's automatic choice (tuning aid)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.conv_bench import SHAPES  # noqa: E402


def time_conv(ops, x, wt, sc, sh, rs, y, stride, pad, split, reps=20):
    for _ in range(2):
        ops.conv2d_nhwc(x, wt, sc, sh, rs, stride=stride, pad=pad, relu=True, split_k=split, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.conv2d_nhwc(x, wt, sc, sh, rs, stride=stride, pad=pad, relu=True, split_k=split, out=y)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    from faster_rcnn_pytorch_multimodal_amd import _hip, ops
    lib = _hip.load()
    only = [a for a in sys.argv[1:] if not a.startswith("--")] or ["l1", "l2", "l3", "rpn", "stem"]
    dump = {}
    g = torch.Generator().manual_seed(0)
    tiles = [(4, 2), (2, 4), (2, 2), (2, 1), (1, 2), (1, 1)]
    for name, n, h, w, c, k, r, stride, pad, res, calls in SHAPES:
        if not any(name.startswith(o) for o in only):
            continue
        x = torch.randn((n, h, w, c), generator=g).cuda()
        wt = (torch.randn((k, r, r, c), generator=g) * 0.05).cuda()
        sc, sh = torch.rand((k,), generator=g).cuda() + 0.5, torch.randn((k,), generator=g).cuda()
        ho, wo = ops.conv_out_hw(h, w, r, r, stride, pad)
        rs = torch.randn((n, ho, wo, k), generator=g).cuda() if res else None
        y = torch.empty((n, ho, wo, k), device="cuda")
        lib.frcnn_conv2d_set_tile(0, 0)
        auto = time_conv(ops, x, wt, sc, sh, rs, y, stride, pad, 0)
        best = (1e9, None)
        ksteps = (r * r * c + 31) // 32
        for tm, tn in tiles:
            lib.frcnn_conv2d_set_tile(tm, tn)
            for split in (1, 2, 3, 4, 6, 8, 12, 16):
                if split > 1 and ksteps // split < 2:
                    continue
                try:
                    t = time_conv(ops, x, wt, sc, sh, rs, y, stride, pad, split, reps=10)
                except Exception:
                    continue
                dump.setdefault(name, {"shape": [n, h, w, c, k, r, stride, pad], "calls": calls, "t": {}})["t"]["%d,%d,%d" % (tm, tn, split)] = t
                if t < best[0]:
                    best = (t, (tm, tn, split))
        lib.frcnn_conv2d_set_tile(0, 0)
        dump[name]["auto"] = auto
        print("%-24s auto %7.1f us   best %7.1f us  tile %s  (x%d per frame: %.0f us saved)"
              % (name, auto, best[0], best[1], calls, (auto - best[0]) * calls))


    if "--dump" in sys.argv:
        import json
        json.dump(dump, open(os.path.join(ROOT, "gpurun_out", "conv_sweep.json"), "w"))


if __name__ == "__main__":
    main()
