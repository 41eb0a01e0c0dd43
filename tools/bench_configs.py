#!/usr/bin/env python
"""Secondary measurements for BASELINE.json configs[2] (LiDAR-BEV forward) and configs[3] (res101+FPN forward +
backward of one train_step).  bench.py stays the judged headline (configs[1]); this prints one JSON line per config.

    python tools/bench_configs.py [--lidar] [--train] [--steps N]
"""
import argparse
import json
import os
import sys
import time

MFMA_F32_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: dense fp32 matrix peak

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lidar_forward(steps, streams=4):
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.model.frame_graph import FrameRunner
    from faster_rcnn_pytorch_multimodal_amd.nets.lidarnet import lidarnet
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    C.reset_cfg()
    C.cfg.NET_TYPE = "lidar"
    net = lidarnet(num_layers=101)
    net.create_architecture(2, tag="default", anchor_scales=C.cfg.LIDAR.ANCHOR_SCALES[0],
                            anchor_ratios=C.cfg.LIDAR.ANCHOR_ANGLES)
    net.load_state_dict(seeded_state_dict(net, 3, bn_mode="tame"), strict=True)
    net.eval()
    net._device = "cuda:0"
    net.to("cuda:0")
    h, w = 400, 350                                   # --scale 0.5: 0.2 m voxels (minibatch.py:434-438)
    info = np.array([0, w, 0, h, 0, 12, 0.5], np.float32)
    rng = np.random.default_rng(0)
    frames = [torch.from_numpy((rng.random((1, h, w, 15)) * (rng.random((1, h, w, 15)) < 0.05)).astype(np.float32)).cuda()
              for _ in range(4)]
    runners = [FrameRunner(net, h, w, 15, info, 0.5, 100) for _ in range(streams)]
    from faster_rcnn_pytorch_multimodal_amd import ops
    from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device
    ops.flops_begin()                                  # one eager frame with the tuned plans: FLOPs per frame
    detect_frame_device(net, frames[0], info, 0.5, 100, 100)
    torch.cuda.synchronize()
    fl = ops.flops_end()
    from faster_rcnn_pytorch_multimodal_amd.model.streams import concurrent_streams
    sts, distinct = concurrent_streams(streams, "cuda:0")      # streams that overlap by measurement (model/streams.py)
    outs = [None] * streams

    def step(i):
        with torch.cuda.stream(sts[i % streams]):
            outs[i % streams] = runners[i % streams].run(frames[i % 4])

    for i in range(2 * streams):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    C.reset_cfg()
    return {"metric": "frames/sec res101 LiDAR-BEV Faster-RCNN 400x350x15 (--scale 0.5)", "value": steps / dt,
            "unit": "frames/s", "ms_per_step": 1e3 * dt / steps, "n_gpus": 1, "steps": steps, "dtype": "f32",
            "roofline": _roofline(fl, dt / steps, ("fwd",)),
            "config": {"workload": "BASELINE.json configs[2]", "frames_in_flight": streams, "launch": "hipGraph replay",
                       "streams_on_distinct_hardware_queues": distinct,
                       "detections_last_frame": outs[0][1].cpu().tolist()}}


def _roofline(fl, seconds_per_step, kinds):
    """fp32-MFMA roofline of a step from the convolution FLOPs its launches carry (ops.flops_begin / flops_end): `achieved`
    counts the direct-form FLOPs of the convolutions as launched (forward + data gradient + filter gradient), `executed`
    what the matrix pipe multiplies (Winograd F(2x2,3x3) plans do 16/36 of the direct form); both over the WHOLE timed step
    (every other kernel's time included), so they are lower bounds on the convolution kernels' own rate."""
    direct = sum(fl[k] for k in kinds)
    executed = sum(fl[k + "_executed"] for k in kinds)
    return {"bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_F32_PEAK_TFLOPS,
            "achieved": direct / seconds_per_step / 1e12, "frac": direct / seconds_per_step / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "achieved_executed": executed / seconds_per_step / 1e12,
            "frac_executed": executed / seconds_per_step / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "gflop_direct_form": {k: fl[k] / 1e9 for k in kinds}, "gflop_executed": {k: fl[k + "_executed"] / 1e9 for k in kinds},
            "what": "convolution FLOPs of one step (%s) / wall time per step of the timed mode / fp32 MFMA peak" % " + ".join(kinds)}


def _timed_train_windows(net, blobs, opt, steps, windows=3):
    """8 untimed steps (allocator / clocks settle after the plan tuning), then `windows` timed windows of `steps`
    train steps each (pseudo batch of 16, train_val.py:379); returns the losses of the last window and the MEDIAN
    window time, so that a one-off stall does not decide the number."""
    for i in range(8):
        net.train_step(blobs, opt, update_weights=False)
    times, losses = [], []
    for _ in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = [net.train_step(blobs, opt, update_weights=(i % 16 == 15)) for i in range(steps)]
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    return losses, sorted(times)[len(times) // 2]


def _timed_pipeline_windows(net, blobs, opt, steps, inflight, windows=3):
    """The same windows with `inflight` frames of a pseudo batch in flight (model/train_graph.TrainPipeline): a frame's loss
    is collected when its slot comes round again; every 16th frame the slots' gradients are merged and the optimizer
    steps."""
    from faster_rcnn_pytorch_multimodal_amd.model.train_graph import TrainPipeline
    opt.zero_grad(set_to_none=False)
    pipe = TrainPipeline(net, slots=inflight)

    def run(n):
        losses = []
        for i in range(n):
            if pipe.in_flight() >= pipe.slots:
                losses.append(pipe.collect()[0])
            pipe.submit(blobs)
            if i % 16 == 15:
                while pipe.in_flight():
                    losses.append(pipe.collect()[0])
                pipe.flush()
                net.apply_update(opt, in_place=True)
        while pipe.in_flight():
            losses.append(pipe.collect()[0])
        return losses

    run(8)
    times, losses = [], []
    for _ in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = run(steps)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    pipe.flush()
    return losses, sorted(times)[len(times) // 2]


def _build_fpn_train():
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
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
    data = torch.from_numpy((rng.standard_normal((1, 600, 1000, 3)) * 50).astype(np.float32)).cuda()
    info = np.array([0, 1000, 0, 600, 0, 0, 1.0], np.float32)
    wh = rng.uniform(30, 300, (8, 2))
    xy = rng.uniform(0, 1, (8, 2)) * (np.array([1000, 600]) - wh - 1)
    gt = np.concatenate((xy, xy + wh, np.ones((8, 1))), 1).astype(np.float32)          # 8 gt boxes
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-4, momentum=C.cfg.TRAIN.MOMENTUM,
                          weight_decay=C.cfg.TRAIN.WEIGHT_DECAY)
    blobs = {"data": data, "info": info, "gt_boxes": gt, "gt_boxes_dc": np.zeros((0, 4), np.float32)}
    return net, opt, blobs


def _snapshot(net):
    return {k: v.detach().clone() for k, v in net.state_dict().items()}


def _restore(net, opt, state):
    """Weights, buffers (BatchNorm statistics) and optimizer back to the snapshot, IN PLACE (captured graphs read parameters
    and the derived filters by address), so that the modes of one run train the same net from the same point and their losses
    are comparable."""
    from faster_rcnn_pytorch_multimodal_amd.nets.hip_modules import refresh_derived_weights
    with torch.no_grad():
        own = net.state_dict()
        for k, v in state.items():
            own[k].copy_(v)
        refresh_derived_weights(net)
    opt.state.clear()
    opt.zero_grad(set_to_none=False)
    torch.manual_seed(7)
    torch.cuda.synchronize()


def _tune_and_count(net, opt, blobs, autotune=True):
    """Two eager steps with the plan autotuner on, then one eager step whose convolution calls are counted."""
    from faster_rcnn_pytorch_multimodal_amd import ops
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    torch.manual_seed(C.cfg.RNG_SEED)
    ops.set_conv_autotune(autotune)      # time the (tile, split-K) candidates of every forward / data-gradient shape once
    try:
        for _ in range(2):
            net.train_step(blobs, opt, update_weights=False)
    finally:
        torch.cuda.synchronize()
        ops.set_conv_autotune(False)
    opt.zero_grad()
    ops.flops_begin()
    net.train_step(blobs, opt, update_weights=False)
    torch.cuda.synchronize()
    fl = ops.flops_end()
    opt.zero_grad()
    return fl


def fpn_train(steps, autotune=True, graph=False, inflight=1, modes=None):
    """BASELINE.json configs[3].  ``modes``: list out of 'eager', 'graph', 'pipeline' measured on ONE net in this order
    (bench.py's extra_configs); default: the single mode the flags select."""
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    net, opt, blobs = _build_fpn_train()
    fl = _tune_and_count(net, opt, blobs, autotune)
    if modes is None:
        modes = ["pipeline"] if inflight > 1 else ["graph"] if graph else ["eager"]
    out = []
    start = _snapshot(net)
    for mode in modes:
        _restore(net, opt, start)          # every mode starts from the SAME weights and an empty optimizer state
        if mode == "pipeline":
            n_in = int(os.environ.get("FRCNN_PIPE_SLOTS", inflight if inflight > 1 else 4))
            losses, dt = _timed_pipeline_windows(net, blobs, opt, steps, n_in)
            launch = "hipGraph replay of the whole step, %d frames of a pseudo batch in flight (TrainPipeline, single-chain graphs)" % n_in
            wg = "in line, grouped per ResNet stage"
        elif mode == "graph":
            net.enable_train_graphs(True)        # model/train_graph.py: the step replayed as one hipGraph
            losses, dt = _timed_train_windows(net, blobs, opt, steps)
            launch, wg = "hipGraph replay of the whole step as one chain, one frame at a time", "in line, grouped per ResNet stage"
        else:
            losses, dt = _timed_train_windows(net, blobs, opt, steps)
            launch, wg = "eager (autograd)", "per layer inside autograd's backward (synchronous)"
        out.append({"metric": "train steps/sec res101+FPN Faster-RCNN 1000x600 forward+backward", "value": steps / dt,
                    "unit": "steps/s", "ms_per_step": 1e3 * dt / steps, "n_gpus": 1, "steps": steps, "dtype": "f32",
                    "mode": mode, "roofline": _roofline(fl, dt / steps, ("fwd", "dgrad", "wgrad")),
                    "config": {"workload": "BASELINE.json configs[3]: 8 random gt boxes, 12000/2000 proposals, 256 sampled RoIs, "
                                           "FIXED_BLOCKS=1, SGD update every 16 steps; every mode of a run starts from the same weights "
                                           "and an empty optimizer state, median of 3 windows of `steps` steps",
                               "launch": launch, "filter_gradients": wg, "loss_first": losses[0], "loss_last": losses[-1],
                               "packet_capture_env": os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE")}})
    C.reset_cfg()
    return out if len(out) > 1 else out[0]


def lidar_train(steps, modes=("eager",)):
    """LiDAR-BEV train_step (not a BASELINE config; the training counterpart of configs[2]): 400x350x15 blob, 8 gt
    boxes, FIXED_BLOCKS=1 -> layer2/layer3 BatchNorm with batch statistics.  modes out of 'eager', 'graph'."""
    from faster_rcnn_pytorch_multimodal_amd.layer_utils.generate_3d_anchors import generate_anchors_3d
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.nets.lidarnet import lidarnet
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    C.reset_cfg()
    C.cfg.NET_TYPE = "lidar"
    net = lidarnet(num_layers=101)
    net.create_architecture(2, tag="default", anchor_scales=C.cfg.LIDAR.ANCHOR_SCALES[0],
                            anchor_ratios=C.cfg.LIDAR.ANCHOR_ANGLES)
    net.load_state_dict(seeded_state_dict(net, 3, bn_mode="tame"), strict=True)
    net._device = "cuda:0"
    net.to("cuda:0")
    net.train()
    h, w, scale = 400, 350, 0.5
    rng = np.random.default_rng(0)
    data = torch.from_numpy((rng.random((1, h, w, 15)) * (rng.random((1, h, w, 15)) < 0.05)).astype(np.float32)).cuda()
    info = np.array([0, w, 0, h, 0, 12, scale], np.float32)
    _, a3, a2 = generate_anchors_3d((h + 15) // 16, (w + 15) // 16, 16, C.cfg.LIDAR.ANCHOR_SCALES[0],
                                    C.cfg.LIDAR.ANCHOR_ANGLES, scale, device="cuda:0")
    a3, a2 = a3.cpu().numpy(), a2.cpu().numpy()
    inside = np.where((a2[:, 0] >= 0) & (a2[:, 1] >= 0) & (a2[:, 2] < w) & (a2[:, 3] < h))[0]
    gt = a3[inside[rng.choice(len(inside), 8, replace=False)]].copy()
    gt[:, 0:2] += rng.uniform(-2, 2, (8, 2))
    gt[:, 6] += rng.uniform(-0.2, 0.2, 8)
    gt = np.concatenate((gt, np.ones((8, 1))), 1).astype(np.float32)
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=1e-5, momentum=C.cfg.TRAIN.MOMENTUM,
                          weight_decay=C.cfg.TRAIN.WEIGHT_DECAY)
    blobs = {"data": data, "info": info, "gt_boxes": gt, "gt_boxes_dc": np.zeros((0, 4), np.float32)}
    fl = _tune_and_count(net, opt, blobs)
    out = []
    start = _snapshot(net)
    for mode in modes:
        _restore(net, opt, start)
        if mode == "graph":
            net.enable_train_graphs(True)
        if mode == "pipeline":
            losses, dt = _timed_pipeline_windows(net, blobs, opt, steps, 4)
        else:
            losses, dt = _timed_train_windows(net, blobs, opt, steps)
        out.append({"metric": "train steps/sec res101 LiDAR-BEV Faster-RCNN 400x350x15 forward+backward", "value": steps / dt,
                    "unit": "steps/s", "ms_per_step": 1e3 * dt / steps, "n_gpus": 1, "steps": steps, "dtype": "f32", "mode": mode,
                    "roofline": _roofline(fl, dt / steps, ("fwd", "dgrad", "wgrad")),
                    "config": {"workload": "training counterpart of BASELINE.json configs[2]: 8 gt boxes, 256 sampled RoIs, "
                                           "FIXED_BLOCKS=1 (layer2/3 BatchNorm on batch statistics)",
                               "launch": {"graph": "hipGraph replay of the whole step, one frame at a time",
                                          "pipeline": "hipGraph replay, 4 frames of a pseudo batch in flight (TrainPipeline, single-chain graphs)",
                                          "eager": "eager (autograd)"}[mode],
                               "loss_first": losses[0], "loss_last": losses[-1]}})
    C.reset_cfg()
    return out if len(out) > 1 else out[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lidar", action="store_true")
    ap.add_argument("--train", action="store_true")
    ap.add_argument("--lidar-train", action="store_true")
    ap.add_argument("--steps", type=int, default=0)
    ap.add_argument("--no-autotune", action="store_true", help="heuristic conv plans in the training step")
    ap.add_argument("--graph", action="store_true", help="--train: replay the step as a hipGraph (Network.enable_train_graphs)")
    ap.add_argument("--inflight", type=int, default=1, help="--train: frames of a pseudo batch in flight (TrainPipeline)")
    ap.add_argument("--modes", default=None, help="--train / --lidar-train: comma list out of eager,graph,pipeline measured on one net")
    ap.add_argument("--roi-bwd-per-sample", action="store_true", help="A/B: sample-by-sample RoIAlign backward instead of the planned one")
    ap.add_argument("--rpn-dense-backward", action="store_true", help="A/B: dense backward through the RPN head")
    ap.add_argument("--no-dgrad-winograd-cache", action="store_true", help="A/B: transform the data-gradient filter per call")
    ap.add_argument("--no-fuse-act-bwd", action="store_true", help="A/B: separate frcnn_act_bwd passes inside the Bottleneck backward")
    ap.add_argument("--group-wgrad", action="store_true", help="A/B: grouped filter-gradient launches per ResNet stage (autograd_ops.GROUP_WGRAD)")
    ap.add_argument("--group-wgrad-size", type=int, default=0, help="A/B: layers per grouped filter-gradient launch")
    ap.add_argument("--wgrad-variant", type=int, default=0, help="A/B: frcnn_conv2d_wgrad_set_variant (1 register-staged kernels, 2 LDS-DMA)")
    ap.add_argument("--plans", default="", help="import a convolution plan table (bench.py --plans / profiles/r05_plans.json) before running")
    ap.add_argument("--export-plans", default="", help="write the convolution plan table after the run")
    ap.add_argument("--wgrad-streams", type=int, default=0, help="A/B: side streams of the filter gradients (autograd_ops.WGRAD_SIDE_STREAMS)")
    args = ap.parse_args()
    if args.plans:
        from faster_rcnn_pytorch_multimodal_amd import ops as _o3
        with open(args.plans) as f:
            _o3.import_conv_plans(json.load(f))
    if args.wgrad_streams:
        from faster_rcnn_pytorch_multimodal_amd.nets import autograd_ops as _a5
        _a5.WGRAD_SIDE_STREAMS = args.wgrad_streams
    if args.wgrad_variant:
        from faster_rcnn_pytorch_multimodal_amd import ops as _o2
        _o2.set_wgrad_variant(args.wgrad_variant)
    if args.roi_bwd_per_sample:
        from faster_rcnn_pytorch_multimodal_amd import ops as _o
        _o.ROI_ALIGN_BWD_PLANNED = False
    if args.group_wgrad_size:
        from faster_rcnn_pytorch_multimodal_amd.nets import autograd_ops as _a4
        _a4.GROUP_WGRAD_SIZE = args.group_wgrad_size
    if args.group_wgrad:
        from faster_rcnn_pytorch_multimodal_amd.nets import autograd_ops as _a3
        _a3.GROUP_WGRAD = True
    if args.no_fuse_act_bwd:
        from faster_rcnn_pytorch_multimodal_amd.nets import autograd_ops as _a2
        _a2.FUSE_ACT_BWD = False
    if args.no_dgrad_winograd_cache:
        from faster_rcnn_pytorch_multimodal_amd.nets import autograd_ops as _a
        _a.DGRAD_WINOGRAD_CACHE = False
    if args.rpn_dense_backward:
        from faster_rcnn_pytorch_multimodal_amd.nets import network as _n
        _n.RPN_BACKWARD_ON_LABELLED_PIXELS = False
    both = not (args.lidar or args.train or args.lidar_train)
    if args.lidar or both:
        print(json.dumps(lidar_forward(args.steps or 80)))
    if args.train or both:
        res = fpn_train(args.steps or 64, not args.no_autotune, args.graph, args.inflight,
                        modes=args.modes.split(",") if args.modes else None)
        for r in (res if isinstance(res, list) else [res]):
            print(json.dumps(r))
    if args.lidar_train or both:
        res = lidar_train(args.steps or 64, modes=tuple(args.modes.split(",")) if args.modes else ("eager", "graph", "pipeline"))
        for r in (res if isinstance(res, list) else [res]):
            print(json.dumps(r))
    if args.export_plans:
        from faster_rcnn_pytorch_multimodal_amd import ops as _o4
        with open(args.export_plans, "w") as f:
            json.dump(_o4.export_conv_plans(), f)


if __name__ == "__main__":
    main()
