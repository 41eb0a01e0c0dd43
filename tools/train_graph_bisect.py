#!/usr/bin/env python
"""Which gradients of a single-chain captured training step diverge between replays, and which feature of the step
makes them?

    python tools/train_graph_bisect.py [--h 600 --w 1000] [--configs a,b,...] [--dump DIR]
    (run with and without DEBUG_CLR_GRAPH_PACKET_CAPTURE=0)

Builds the res101+FPN detector once; per configuration captures the train step as ONE chain
(model/train_graph.TrainStepRunner(inline=True)), replays it R times on the same frame with the same sampling seeds and
reports how many parameters' gradient increments differ between the replays and from the EAGER step with the same seeds.
tools/graph_replay_repro.hip isolates node kinds stand-alone.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    "inline": {},
    "inline_nogroup": {"group_wgrad": False},
    "inline_rpn_dense": {"rpn_dense": True},
    "inline_roi_bwd_per_sample": {"roi_per_sample": True},
    "inline_no_fused_act": {"no_fuse_act": True},
    "inline_igemm_only": {"algo": 1},
    "forked": {"inline": False},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=600)
    ap.add_argument("--w", type=int, default=1000)
    ap.add_argument("--replays", type=int, default=4)
    ap.add_argument("--configs", default=",".join(CONFIGS))
    ap.add_argument("--dump", default=None, help="directory for the captured graphs' DOT dumps")
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    from faster_rcnn_pytorch_multimodal_amd import ops
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.model.train_graph import TrainStepRunner
    from faster_rcnn_pytorch_multimodal_amd.nets import autograd_ops, network
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    print("DEBUG_CLR_GRAPH_PACKET_CAPTURE=%r" % os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), flush=True)
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
    rc = 0
    for name in args.configs.split(","):
        conf = CONFIGS[name]
        network.RPN_BACKWARD_ON_LABELLED_PIXELS = not conf.get("rpn_dense", False)
        ops.ROI_ALIGN_BWD_PLANNED = not conf.get("roi_per_sample", False)
        autograd_ops.FUSE_ACT_BWD = not conf.get("no_fuse_act", False)
        ops.set_conv_algo(conf.get("algo", 0))
        # the eager step with fixed seeds: the reference increments
        torch.manual_seed(1234)
        for _, p in params:
            p.grad = torch.zeros_like(p)
        net.train_step(blobs, None, update_weights=False)
        torch.cuda.synchronize()
        eager = [p.grad.clone() for _, p in params]
        grads = [torch.zeros_like(p) for _, p in params]
        dump = os.path.join(args.dump, name + ".dot") if args.dump else None
        runner = TrainStepRunner(net, h, w, 3, 8, info, grads=grads, inline=conf.get("inline", True),
                                 group_wgrad=conf.get("group_wgrad"), debug_dump=dump)
        incs = []
        for r in range(args.replays):
            for g in grads:
                g.zero_()
            torch.manual_seed(1234)
            runner.run(blobs)
            torch.cuda.synchronize()
            incs.append([g.clone() for g in grads])
        line = []
        worst_names = {}
        for r in range(args.replays):
            n_bad, worst = 0, 0.0
            for (pname, _), a, b in zip(params, eager, incs[r]):
                scale = float(a.abs().max()) or 1.0
                dev = float((a - b).abs().max()) / scale
                dev = dev if np.isfinite(dev) else float("inf")
                if dev > 1e-3:
                    n_bad += 1
                    worst_names.setdefault(pname, []).append((r + 1, dev))
                worst = max(worst, dev)
            line.append("replay%d: %d bad (worst %.1e)" % (r + 1, n_bad, worst))
        print("%-28s %s" % (name, "  ".join(line)), flush=True)
        if args.verbose:
            for pname, lst in list(worst_names.items())[:12]:
                print("      %-44s %s" % (pname, " ".join("r%d:%.1e" % t for t in lst)))
        rc |= int(bool(worst_names))
        del runner, incs, grads
        torch.cuda.synchronize()
    C.reset_cfg()
    return rc


if __name__ == "__main__":
    raise SystemExit(main())
