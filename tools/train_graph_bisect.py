#!/usr/bin/env python
"""Which gradients of a single-chain captured training step diverge between replays?

    python tools/train_graph_bisect.py [--h 600 --w 1000] [--side]      (run with and without DEBUG_CLR_GRAPH_PACKET_CAPTURE=0)

Captures the res101+FPN train step as ONE chain (model/train_graph.TrainStepRunner(inline=True)), replays it R times on
the same frame with the same sampling seeds and prints, per parameter, the deviation of each replay's gradient increment
from the first replay's and from an eager step's.  The list of diverging parameters names the node kind at fault
(tools/graph_replay_repro.hip isolates node kinds stand-alone).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=600)
    ap.add_argument("--w", type=int, default=1000)
    ap.add_argument("--replays", type=int, default=4)
    ap.add_argument("--side", action="store_true", help="filter gradients on a side stream (forked graph) instead of in line")
    args = ap.parse_args()
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.model.train_graph import TrainStepRunner
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    print("DEBUG_CLR_GRAPH_PACKET_CAPTURE=%r inline=%s" % (os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), not args.side))
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    C.cfg.USE_FPN = True
    C.cfg.POOLING_MODE = "multiscale"
    C.cfg.ENABLE_CUSTOM_TAIL = True
    net = imagenet(num_layers=101)
    net.create_architecture(2, tag="default", anchor_scales=C.cfg.ANCHOR_SCALES, anchor_ratios=C.cfg.ANCHOR_RATIOS)
    net.load_state_dict(seeded_state_dict(net, 3, bn_mode="tame"), strict=True)
    net._device = "cuda:0"
    net.to("cuda:0")
    net.train()
    rng = np.random.default_rng(0)
    h, w = args.h, args.w
    data = torch.from_numpy((rng.standard_normal((1, h, w, 3)) * 50).astype(np.float32)).cuda()
    info = np.array([0, w, 0, h, 0, 0, 1.0], np.float32)
    wh = rng.uniform(30, min(300, h / 2), (8, 2))
    xy = rng.uniform(0, 1, (8, 2)) * (np.array([w, h]) - wh - 1)
    gt = np.concatenate((xy, xy + wh, np.ones((8, 1))), 1).astype(np.float32)
    blobs = {"data": data, "info": info, "gt_boxes": gt}
    params = [(n, p) for n, p in net.named_parameters() if p.requires_grad]
    grads = [torch.zeros_like(p) for _, p in params]
    runner = TrainStepRunner(net, h, w, 3, 8, info, grads=grads, inline=not args.side)
    incs = []
    for r in range(args.replays):
        for g in grads:
            g.zero_()
        state = torch.random.get_rng_state()
        runner.run(blobs)
        torch.random.set_rng_state(state)
        torch.cuda.synchronize()
        incs.append([g.clone() for g in grads])
    bad = {}
    for r in range(1, args.replays):
        for (name, _), a, b in zip(params, incs[0], incs[r]):
            scale = float(a.abs().max()) or 1.0
            dev = float((a - b).abs().max()) / scale
            if dev > 1e-5:
                bad.setdefault(name, []).append((r + 1, dev))
    print("%d of %d parameters diverge between replays" % (len(bad), len(params)))
    for name, lst in bad.items():
        print("  %-48s %s" % (name, " ".join("replay%d:%.2e" % t for t in lst)))
    C.reset_cfg()
    return 1 if bad else 0


if __name__ == "__main__":
    raise SystemExit(main())
