#!/usr/bin/env python
"""Headline benchmark: frames/s of the res101 image Faster R-CNN forward at 1000x600 (BASELINE.json
configs[1]) on N MI355X GPUs, one process per GPU, one frame per GPU per step.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one synthetic frame per rank through the whole hot path (channel pad -> ResNet-101 head -> RPN ->
proposal sort/NMS -> RoIAlign -> layer4 on 300 RoIs -> detection tail -> per-class filter with the
max_dets cut), replayed as one hipGraph, followed (N > 1) by the RCCL all-gather of the fixed-size
detection records (the eval collate).  Frames are resident in HBM before the timed region starts.
Rank 0 prints ONE JSON line; `roofline` is for the dominant kernel (the fp32-MFMA implicit-GEMM conv),
`cpu_baseline` times the CPU oracle on a bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, C = 600, 1000, 3
NUM_CLASSES = 2
THRESH, MAX_DETS = 0.5, 100          # tools/test_net.py:290
WEIGHT_SEED, BN_MODE = 3, "tame"     # cfg.RNG_SEED; see DESIGN.md "workload" for why BN is damped
MFMA_F32_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0
REFERENCE_ORDER_FLOPS = 628.4e9          # the frame's convolutions in the reference's order of operations
MIN_TIMED_S = 1.0                    # repeat the --steps region until about this much timed work exists
# Committed rocprofv3 PMC passes (counters cannot be collected inside the timed run: one counter per pass, serialised
# dispatches).  They were collected with the plan table PLANS_FILE; by default this run LOADS that table instead of re-tuning,
# so the kernels it times are the kernels the passes measured (roofline.from_profiles.plans_match says so in the line).
PMC_TIMED_FILE = "r05_pmc_timed.json"         # PMC pass over the TIMED mode (hipGraph replays on 4 streams), tools/pmc_timed.py
PMC_FILE = "r05_pmc_traffic.json"             # FETCH_SIZE / WRITE_SIZE / MfmaUtil passes over the eager one-stream mode
PLANS_FILE = "r05_plans.json"                 # the tuned convolution plans those passes replayed


def synthetic_frame(seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal((1, H, W, C)) * 50).astype(np.float32)


def build_net(device):
    from faster_rcnn_pytorch_multimodal_amd.model import config as Cfg
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    Cfg.reset_cfg()
    Cfg.cfg.NET_TYPE = "image"
    net = imagenet(num_layers=101)
    net.create_architecture(NUM_CLASSES, tag="default", anchor_scales=Cfg.cfg.ANCHOR_SCALES,
                            anchor_ratios=Cfg.cfg.ANCHOR_RATIOS)
    sd = seeded_state_dict(net, WEIGHT_SEED, bn_mode=BN_MODE)
    net.load_state_dict(sd, strict=True)
    net.eval()
    net._device = device
    net.to(device)
    return net, sd


def conv_roofline(net, frame, info, steps):
    """Per-dispatch durations of every convolution kernel over `steps` eager frames: each launch carries its own
    start / stop HIP events on the launch stream (frcnn_conv2d_profile_begin / _end -> hipExtLaunchKernelGGL), i.e. the
    begin -> end time of that dispatch, the quantity rocprofv3 --kernel-trace reports.  Returns the aggregate over all
    conv launches of a frame (main implicit-GEMM kernels + the split-K second passes)."""
    from faster_rcnn_pytorch_multimodal_amd import ops
    from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device
    detect_frame_device(net, frame, info, THRESH, MAX_DETS, MAX_DETS)
    torch.cuda.synchronize()
    ops.PROFILE = []
    ops.conv_profile_begin()
    try:
        for _ in range(steps):
            detect_frame_device(net, frame, info, THRESH, MAX_DETS, MAX_DETS)
        torch.cuda.synchronize()
    finally:
        recs = ops.conv_profile_end()
        shapes, ops.PROFILE = ops.PROFILE, None
    per_call = [[0.0, 0.0] for _ in shapes]               # [GEMM kernel us, other passes us] per conv2d call
    wino_calls = set()
    for us, call, kind in recs:                           # kind 1 = split-K second pass, 2 = Winograd transform
        per_call[call][min(kind, 1)] += us
        if kind == 2:
            wino_calls.add(call)
    per_layer = {}
    total_us = total_flops = main_us = executed_flops = 0.0
    for ci, (shp, (m_us, e_us)) in enumerate(zip(shapes, per_call)):
        fl = shp["flops"] * (3.0 / 4.0 if (shp["r"] == 7 and shp["c"] == 4) else 1.0)  # stem: 3 real channels
        total_us += m_us + e_us
        main_us += m_us
        total_flops += fl
        # Winograd F(2x2,3x3) calls multiply 16 values per 2x2 output tile instead of 36: what the MFMA pipe executed
        executed_flops += (2.0 * 16 * shp["n"] * ((shp["h"] + 1) // 2) * ((shp["w"] + 1) // 2) * shp["c"] * shp["k"]
                           if ci in wino_calls else fl)
        key = "%dx%dx%d c%d k%d r%d s%d" % (shp["n"], shp["h"], shp["w"], shp["c"], shp["k"], shp["r"], shp["stride"])
        ent = per_layer.setdefault(key, [0, 0.0, 0.0])
        ent[0] += 1
        ent[1] += m_us + e_us
        ent[2] += fl
    n_main = sum(1 for r in recs if r[2] == 0)
    n_second = sum(1 for r in recs if r[2] == 1)
    n_wino = sum(1 for r in recs if r[2] == 2)
    wino_us = sum(r[0] for r in recs if r[2] == 2)
    return {"ms_per_frame": 1e-3 * total_us / steps, "flops_per_frame": total_flops / steps,
            "executed_flops_per_frame": executed_flops / steps, "winograd_calls_per_frame": len(wino_calls) / steps,
            "launches_per_frame": len(shapes) / steps, "main_kernel_avg_us": main_us / max(n_main, 1),
            "second_pass_launches_per_frame": n_second / steps,
            "second_pass_avg_us": (total_us - main_us - wino_us) / max(n_second, 1),
            "winograd_transform_launches_per_frame": n_wino / steps, "winograd_transform_us_per_frame": wino_us / steps,
            "per_layer": {k: {"calls_per_frame": v[0] / steps, "us_per_call": v[1] / v[0],
                              "tflops": v[2] / v[1] / 1e6} for k, v in per_layer.items()}}


def roi_align_timing(net, steps, prof=None):
    """HIP-event timing of the RoIAlign launch on the last frame's feature map / rois."""
    from faster_rcnn_pytorch_multimodal_amd import ops
    feat = net._act_summaries["conv"]
    rois = net._predictions["rois"]
    ops.roi_align_nhwc(feat, rois, 7, 1.0 / 16.0, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        ops.roi_align_nhwc(feat, rois, 7, 1.0 / 16.0, 0)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / steps
    n, h, w, c = feat.shape
    bytes_ = h * w * c * 4 + rois.shape[0] * 7 * 7 * c * 4 + rois.numel() * 4
    out = {"bound": "hbm", "kernel": "roi_plan_kernel + roi_align_fwd_planned (one frcnn_roi_align_fwd call)", "achieved": bytes_ / us / 1e3, "peak": HBM_PEAK_GBS,
           "unit": "GB/s", "frac": bytes_ / us / 1e3 / HBM_PEAK_GBS, "traffic": (prof or {}).get("roi_align_traffic_bytes_per_call"),
           "traffic_source": "roofline.from_profiles (committed rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE passes; the RoIAlign kernels do not "
                             "depend on the convolution plans)",
           "us_per_launch": us,
           "algorithmic_bytes": bytes_,
           "what": "the reference's operation: RoIAlign of the %d-channel feature map (Network._crop_pool_layer)" % c}
    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / steps

    # a second, trained-like RoI set: 300 seeded boxes with sides of 32..320 px inside the frame (an untrained RPN proposes
    # frame-wide strips, the heaviest windows the op can get)
    g = torch.Generator().manual_seed(11)
    wh = torch.rand(rois.shape[0], 2, generator=g) * 288 + 32
    xy = torch.rand(rois.shape[0], 2, generator=g) * (torch.tensor([float(W), float(H)]) - wh - 1)
    typical = torch.cat((torch.zeros(rois.shape[0], 1), xy, xy + wh), 1).to(feat.device)
    u_typ = timed(lambda: ops.roi_align_nhwc(feat, typical, 7, 1.0 / 16.0, 0))
    out["typical_boxes"] = {"what": "300 seeded boxes with sides 32..320 px", "us_per_launch": u_typ,
                            "achieved": bytes_ / u_typ / 1e3, "frac": bytes_ / u_typ / 1e3 / HBM_PEAK_GBS}
    out["bound_note"] = ("every RoI reads its whole window through the vector L1 (the bench's RoIs: ~326 MB per launch against "
                         "70 MB algorithmic); measured L2->L1 rate for this access pattern 31-33 TB/s L2-resident, 15 TB/s from "
                         "the Infinity Cache (tools/l2_bw.hip): a per-RoI kernel bottoms out at 11-13 us + the plan launch "
                         "(profiles/r03_roi_align.md), i.e. below the 0.60 target whatever the schedule")
    from faster_rcnn_pytorch_multimodal_amd.nets import network as N
    if N.PROJECT_BEFORE_POOLING:
        # the timed frame pools the projected map instead (Network._layer4_projected): layer4[0].conv1 | downsample[0] as ONE
        # 1024 -> 2560 convolution, its 512 + 2048 channels pooled through one plan (frcnn_roi_align_fwd_split)
        ck = 2560
        gmap = torch.randn((1, h, w, ck), device=feat.device)
        sc, sh = torch.rand(ck, device=feat.device) + 0.5, torch.randn(ck, device=feat.device)
        b = h * w * ck * 4 + rois.shape[0] * 7 * 7 * ck * 4 + rois.numel() * 4 + 8 * ck
        calls = {}
        for name, rset in (("bench_rois", rois), ("typical_boxes", typical)):
            u = timed(lambda: ops.roi_align_split(gmap, rset, 7, 1.0 / 16.0, 512, 0, scale=sc, shift=sh, relu1=True))
            calls[name] = {"us_per_call": u, "achieved": b / u / 1e3, "frac": b / u / 1e3 / HBM_PEAK_GBS}
        out["in_timed_frame"] = {"what": "frcnn_roi_align_fwd_split on the 2560-channel map conv1|downsample[0](F): one plan "
                                         "launch + two pooling launches (512 channels BatchNorm + ReLU, 2048 channels BatchNorm)",
                                 "algorithmic_bytes": b, "calls": calls}
    return out


def nms_timing(net, steps):
    """HIP-event timing of the RPN NMS (bit-matrix + scan launches) on the last frame's sorted proposals: achieved
    bytes = boxes read + the suppression bit-matrix written by the mask kernel and read back by the scan."""
    from faster_rcnn_pytorch_multimodal_amd import ops
    p = net._predictions
    order = p["rpn_order"]
    n = int(order.numel())
    boxes = ops.gather_rows(p["rpn_proposals"], order, None)
    ops.nms_sorted(boxes, 0.7, 300)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        ops.nms_sorted(boxes, 0.7, 300)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / steps
    words = (n + 63) // 64
    bytes_ = n * 16 + 2 * n * words * 8
    return {"bound": "hbm", "kernel": "nms_mask_kernel + nms_scan_kernel (%d boxes)" % n, "achieved": bytes_ / us / 1e3,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_ / us / 1e3 / HBM_PEAK_GBS, "us_per_launch": us,
            "algorithmic_bytes": bytes_, "note": "latency-bound: chunks of 64 boxes are resolved in order (ballot rounds per chunk, row fetches by helper waves)"}


def cpu_baseline(sd, frames, info):
    """The CPU oracle (PyTorch-CPU restatement of the reference path) on this host: 1 warm-up + len(frames)
    timed frames of the same workload."""
    from oracle import frcnn_oracle as O
    net = O.ImageNetOracle(num_classes=NUM_CLASSES)
    net.load_state_dict(sd, strict=True)
    O.frame_detect(net, frames[0], info, NUM_CLASSES, THRESH, MAX_DETS)
    t0 = time.perf_counter()
    dets = [O.frame_detect(net, f, info, NUM_CLASSES, THRESH, MAX_DETS) for f in frames]
    dt = time.perf_counter() - t0
    return {"value": len(frames) / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d frames of the 1000x600 res101 workload (CPU oracle: torch-CPU convs + numpy NMS/RoIAlign), "
                      "%.1f s, os.cpu_count()=%d" % (len(frames), dt, os.cpu_count())}, dets


def structured_rpn(seed, h=38, w=63, a=25):
    """SURVEY 8d cfg-2 "structured" RPN output: logits ~ N(0,1)*2, deltas ~ N(0,0.3).  A random-init RPN gives
    near-identical scores for all 59 850 anchors, so its ranking is decided by rounding noise and differs between ANY
    two implementations; with these injected values the proposal stage is well-conditioned."""
    g = torch.Generator().manual_seed(1000 + seed)
    cls = torch.randn(1, 2 * a, h, w, generator=g) * 2.0          # (1,2A,H,W) like rpn_cls_score_net's output
    box = torch.randn(1, h, w, 4 * a, generator=g) * 0.3          # (1,H,W,4A)
    return cls, box


def fuse_rpn(cls, box, ld=152):
    """(1,2A,H,W) logits + (1,H,W,4A) deltas -> the fused RPN head layout (1,H,W,ld) [bg | fg | deltas | pad]."""
    a = cls.shape[1] // 2
    fused = torch.zeros((1, cls.shape[2], cls.shape[3], ld))
    fused[..., :2 * a] = cls.permute(0, 2, 3, 1)
    fused[..., 2 * a:6 * a] = box
    return fused


def map_delta(net, sd, frames_host, info):
    """"mAP delta vs CPU ref" of the metric, on the structured-RPN variant of the frames (same injected RPN logits /
    deltas on both paths; backbone, RoIAlign, layer4, heads and the per-class filter are each path's own).
    Ground truth := the CPU oracle's detections of a frame (AP_cpu = 1 by construction); AP_gpu = VOC
    continuous-area AP (lib/datasets/voc_eval.py:53-69) at IoU 0.7 of the device detections against it."""
    from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device
    from oracle import frcnn_oracle as O
    # a random-init head scores every RoI 0.5 +- 1e-3, so the rank of two RoIs (hence who suppresses whom in the
    # per-class NMS) hangs on the 7th digit in ANY implementation; spread the scores like a trained head does
    sd = dict(sd)
    sd["cls_score_net.weight"] = sd["cls_score_net.weight"] * 8.0
    cpu = O.ImageNetOracle(num_classes=NUM_CLASSES)
    cpu.load_state_dict(sd, strict=True)
    net.load_state_dict(sd, strict=True)
    aps, same_rois = [], []
    for i, f in enumerate(frames_host):
        cls, box = structured_rpn(i)
        ref = O.frame_detect(cpu, f, info, NUM_CLASSES, THRESH, MAX_DETS, structured=(cls, box))
        net._rpn_override = fuse_rpn(cls, box).to(net._device)
        try:
            dets, counts = detect_frame_device(net, f, info, THRESH, MAX_DETS, MAX_DETS)
            n = int(net._predictions["rois_count"].item())
            rois = net._predictions["rois"][:n].cpu()
        finally:
            net._rpn_override = None
        same_rois.append(bool(n == cpu._dbg["keep"].shape[0] and
                              torch.equal(net._predictions["rpn_order"][net._predictions["rpn_keep"][:n]].cpu(),
                                          cpu._dbg["order"][cpu._dbg["keep"]])))
        dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
        valid = lambda b: b[(b[:, 2] > b[:, 0]) & (b[:, 3] > b[:, 1])]
        for j in range(1, NUM_CLASSES):
            gt = valid(ref[j]) if len(ref[j]) else ref[j]
            if len(gt) == 0:
                continue
            aps.append(O.average_precision(valid(dets[j, :counts[j]]), gt[:, :4], iou_thresh=0.7))
    if not aps:
        return None
    ap_gpu = float(np.mean(aps))
    return {"ap_cpu": 1.0, "ap_gpu": ap_gpu, "delta": abs(1.0 - ap_gpu), "iou": 0.7, "frames": len(frames_host),
            "proposal_indices_bit_exact": all(same_rois),
            "weights": "bench weights with cls_score_net.weight x8 (well-conditioned scores); boxes left with "
                       "x2 < x1 or y2 < y1 by the reference's one-sided frame clamp are dropped on both sides",
            "ground_truth": "the CPU oracle's detections (thresh %.1f, max_dets %d) on the structured-RPN frames"
                            % (THRESH, MAX_DETS)}


class ResidentFrames:
    """Frame source of model.test.test_net (the ``blobs_at`` protocol of its docstring) over frames resident in HBM: frame
    i of the sequence is resident frame i % len(frames), exactly what the timed steps of this file replay."""
    name = "bench_resident_frames"

    def __init__(self, frames, info, n, num_classes=NUM_CLASSES):
        self.frames, self.info, self.n, self.num_classes = frames, info, int(n), num_classes

    def num_frames(self, mode):
        return self.n

    def blobs_at(self, i, mode):
        return {"data": self.frames[i % len(self.frames)], "info": self.info}


def drop_in_timing(net, frames, info, n_frames, value, calls=3):
    """frames/s of the DROP-IN entry point: ``model.test.test_net(net, db, out_dir, max_dets, thresh, mode)`` as
    tools/test_net.py:290 calls it (lib/model/test.py:138-257), over `n_frames` frames resident in HBM.  The whole call is
    timed - frame loop, collate, unpacking into all_boxes, detections.pkl and the per-class text files.  test_net replays
    captured frames, cfg.TEST.FRAMES_IN_FLIGHT in flight (model/frame_graph.FramePool); every all_boxes[cls][frame] of
    the timed calls is compared with the eager single-stream record of that frame, bit for bit."""
    import shutil
    import tempfile
    from faster_rcnn_pytorch_multimodal_amd.model import collate
    from faster_rcnn_pytorch_multimodal_amd.model.config import cfg
    from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device, test_net
    out_dir = tempfile.mkdtemp(prefix="frcnn_bench_dropin_")
    try:
        lanes = int(cfg.TEST.FRAMES_IN_FLIGHT)
        test_net(net, ResidentFrames(frames, info, 3 * lanes), out_dir, max_dets=MAX_DETS, thresh=THRESH, mode="val")  # captures
        db = ResidentFrames(frames, info, n_frames)
        runs = []
        for _ in range(calls):
            timers = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            boxes = test_net(net, db, out_dir, max_dets=MAX_DETS, thresh=THRESH, mode="val", timers=timers)
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0, timers, boxes))
        max_out = max(MAX_DETS, int(cfg.TEST.RPN_POST_NMS_TOP_N))
        expected = []
        for f in frames:
            dets, counts = detect_frame_device(net, f, info, THRESH, MAX_DETS, max_out)
            expected.append(collate.unpack_records(collate.pack_record(dets, counts).unsqueeze(0), NUM_CLASSES, max_out)[0])
        bad = sum(1 for _, _, boxes in runs for i in range(n_frames) for j in range(1, NUM_CLASSES)
                  if not np.array_equal(np.asarray(boxes[j][i]).reshape(-1, 5), expected[i % len(frames)][j]))
        runs.sort(key=lambda r: r[0])
        dt, timers, _ = runs[len(runs) // 2]
        fps = n_frames / dt
        return {"frames_s": fps, "ms_per_frame": 1e3 * dt / n_frames, "frames": n_frames, "calls": calls,
                "loop_only_frames_s": n_frames / timers["loop_s"], "ratio_to_value": fps / value,
                "seconds_each_call": [r[0] for r in runs], "pool": timers["pool"],
                "records_equal_to_eager_path": bad == 0, "mismatching_entries": bad,
                "entry_point": "model.test.test_net(net, db, out_dir, max_dets=%d, thresh=%.1f, mode='val') - the call of "
                               "tools/test_net.py:290 / lib/model/test.py:138-257; whole call timed (frame loop + collate + "
                               "all_boxes + detections.pkl + result text files), median of %d calls; frames resident in "
                               "HBM; cfg.TEST.FRAME_GRAPHS=%s FRAMES_IN_FLIGHT=%d (the defaults)"
                               % (MAX_DETS, THRESH, calls, bool(cfg.TEST.FRAME_GRAPHS), lanes)}
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)


def drop_in_uncertainty(device, n_frames, frames_host, info, calls=3):
    """The same two measurements with the uncertainty heads on (cfg.UC.EN_{BBOX,CLS}_{ALEATORIC,EPISTEMIC}, E_NUM_SAMPLE =
    10 Monte-Carlo passes per frame: lib/model/test.py:73-77, lib/model/config.py:46): FrameRunner x 4 streams driven
    like the timed steps of this file (`runner_frames_s`) and the drop-in call test_net (`frames_s`)."""
    import shutil
    import tempfile
    from faster_rcnn_pytorch_multimodal_amd.model import config as Cfg
    from faster_rcnn_pytorch_multimodal_amd.model.frame_graph import FrameRunner
    from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device, test_net
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    from faster_rcnn_pytorch_multimodal_amd.nets.uncertainty import num_uncertainty_pos
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    from faster_rcnn_pytorch_multimodal_amd.model import collate
    Cfg.reset_cfg()
    out_dir = tempfile.mkdtemp(prefix="frcnn_bench_dropin_uc_")
    try:
        cfg = Cfg.cfg
        cfg.NET_TYPE = "image"
        for k in ("EN_BBOX_ALEATORIC", "EN_CLS_ALEATORIC", "EN_BBOX_EPISTEMIC", "EN_CLS_EPISTEMIC"):
            cfg.UC[k] = True
        net = imagenet(num_layers=101)
        net.create_architecture(NUM_CLASSES, tag="default", anchor_scales=cfg.ANCHOR_SCALES, anchor_ratios=cfg.ANCHOR_RATIOS)
        net.load_state_dict(seeded_state_dict(net, WEIGHT_SEED, bn_mode=BN_MODE), strict=True)
        net.eval()
        net._device = device
        net.to(device)
        frames = [torch.from_numpy(f).to(device) for f in frames_host]
        lanes = int(cfg.TEST.FRAMES_IN_FLIGHT)
        max_out = max(MAX_DETS, int(cfg.TEST.RPN_POST_NMS_TOP_N))
        runners = [FrameRunner(net, H, W, C, info, THRESH, MAX_DETS, max_out=max_out) for _ in range(lanes)]
        streams = [torch.cuda.Stream(device=device) for _ in range(lanes)]

        def loop(n):
            for i in range(n):
                with torch.cuda.stream(streams[i % lanes]):
                    runners[i % lanes].run(frames[i % len(frames)])
            torch.cuda.synchronize()

        loop(2 * lanes)
        t = []
        for _ in range(calls):
            t0 = time.perf_counter()
            loop(n_frames)
            t.append(time.perf_counter() - t0)
        runner_fps = n_frames / sorted(t)[len(t) // 2]
        del runners
        net.set_uc_seed(1)
        test_net(net, ResidentFrames(frames, info, 3 * lanes), out_dir, max_dets=MAX_DETS, thresh=THRESH, mode="val")
        db = ResidentFrames(frames, info, n_frames)
        runs = []
        for _ in range(calls):
            net.set_uc_seed(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            boxes = test_net(net, db, out_dir, max_dets=MAX_DETS, thresh=THRESH, mode="val")
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0, boxes))
        # the eager path draws the same masks for the same forward count: replay the first frames eagerly and compare
        width = 5 + num_uncertainty_pos(NUM_CLASSES, 4)
        net.set_uc_seed(1)
        bad, checked = 0, min(n_frames, 2 * len(frames))
        for i in range(checked):
            dets, counts = detect_frame_device(net, frames[i % len(frames)], info, THRESH, MAX_DETS, max_out)
            exp = collate.unpack_records(collate.pack_record(dets, counts).unsqueeze(0), NUM_CLASSES, max_out, width)[0]
            bad += sum(1 for _, boxes in runs for j in range(1, NUM_CLASSES)
                       if not np.array_equal(np.asarray(boxes[j][i]).reshape(-1, width), exp[j]))
        runs.sort(key=lambda r: r[0])
        fps = n_frames / runs[len(runs) // 2][0]
        return {"frames_s": fps, "ms_per_frame": 1e3 / fps, "runner_frames_s": runner_fps, "ratio_to_runner": fps / runner_fps,
                "frames": n_frames,
                "e_num_sample": int(cfg.UC.E_NUM_SAMPLE), "a_num_ce_sample": int(cfg.UC.A_NUM_CE_SAMPLE),
                "row_width": width, "records_equal_to_eager_path": bad == 0, "frames_checked_against_eager": checked,
                "what": "cfg.UC.EN_BBOX/CLS_ALEATORIC + EN_BBOX/CLS_EPISTEMIC: 10 Monte-Carlo passes of the heads per frame as "
                        "batched launches, uncertainty columns per detection; frames_s through model.test.test_net (whole "
                        "call), runner_frames_s = FrameRunner x %d streams driven directly" % lanes}
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
        Cfg.reset_cfg()


def plans_sha(rows):
    """Hash of a convolution plan table (ops.export_conv_plans rows), independent of row order."""
    import hashlib
    return hashlib.sha1(json.dumps(sorted([int(v) for v in r] for r in rows)).encode()).hexdigest()[:16]


def _profile(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def from_profiles(run_plans_sha):
    """Everything this line takes from COMMITTED profile files (another process, another box, another day), in one object,
    with what it was collected on: file names, the git head recorded at collection, the hash of the plan table the
    profiled command replayed and whether it equals this run's.  Values of a profile whose plans differ from this run's are
    reported as null (they would describe other kernels)."""
    traffic, timed = _profile(PMC_FILE), _profile(PMC_TIMED_FILE)
    out = {"what": "rocprofv3 PMC passes kept under profiles/ - NOT measured by this run (counters are collected one per pass "
                   "with serialised dispatches); valid for this run only where plans_match is true",
           "plans_sha_run": run_plans_sha, "files": {}}
    ok_all = True
    for key, name, prof in (("traffic", PMC_FILE, traffic), ("timed", PMC_TIMED_FILE, timed)):
        if prof is None:
            out["files"][key] = {"file": "profiles/" + name, "present": False}
            ok_all = False
            continue
        sha = prof.get("plans_sha")
        out["files"][key] = {"file": "profiles/" + name, "present": True, "collected_at_head": prof.get("collected_at_head"),
                             "plans_sha": sha, "plans_match": sha == run_plans_sha}
        ok_all = ok_all and sha == run_plans_sha
    out["plans_match"] = ok_all
    conv = (traffic or {}).get("kernels", {}).get("conv_igemm", {}) if out["files"]["traffic"].get("plans_match") else {}
    roi = (traffic or {}).get("kernels", {}) if traffic else {}
    out["conv_traffic_bytes_per_call"] = conv.get("traffic_bytes_per_launch")
    out["conv_mfma_util_percent_isolated"] = (conv.get("mfma") or {}).get("mfma_util_percent")
    out["conv_mfma_util_by_kernel_isolated"] = {k: round(v["mfma_util_percent"], 1)
                                                for k, v in ((conv.get("mfma") or {}).get("by_kernel") or {}).items()}
    # the RoIAlign kernels do not depend on the conv plans
    out["roi_align_traffic_bytes_per_call"] = (roi.get("roi_align_fwd") or {}).get("traffic_bytes_per_launch")
    out["roi_align_affine_traffic_bytes_per_call"] = (roi.get("roi_align_fwd_affine") or {}).get("traffic_bytes_per_launch")
    out["mfma_busy_ms_per_frame_timed_mode"] = (timed.get("mfma_busy_ms_per_frame")
                                                if timed and out["files"]["timed"].get("plans_match") else None)
    return out


def Cfg_post_nms():
    from faster_rcnn_pytorch_multimodal_amd.model.config import cfg
    return cfg.TEST.RPN_POST_NMS_TOP_N


def extra_configs(steps):
    """BASELINE.json configs[2] (LiDAR-BEV forward) and configs[3] (res101+FPN forward+backward: eager, one captured step at
    a time, three captured steps of a pseudo batch in flight) inside the driver's run, each with the fp32-MFMA roofline of
    ITS timed mode (tools/bench_configs.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_configs as BC
    out = []
    t0 = time.perf_counter()
    out.append(BC.lidar_forward(max(steps, 80)))
    res = BC.fpn_train(64, modes=["eager", "graph", "pipeline"])
    out += res if isinstance(res, list) else [res]
    res = BC.lidar_train(64, modes=("eager", "graph", "pipeline"))
    out += res if isinstance(res, list) else [res]
    return {"seconds": time.perf_counter() - t0, "runs": out}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--conv-algo", type=int, default=0, help="frcnn_conv2d_set_algo: 0 = the autotuner may pick Winograd F(2x2,3x3) "
                    "for the eligible 3x3 layers (default), 1 = implicit GEMM only")
    ap.add_argument("--retune", action="store_true", help="tune the convolution plans in this run instead of loading the committed "
                    "table profiles/%s" % PLANS_FILE)
    ap.add_argument("--plans", default=None, help="JSON file of tuned conv plans: loaded when it exists (profiler runs "
                    "then use exactly the kernels of the timed run), written after warm-up otherwise")
    ap.add_argument("--streams", type=int, default=4,
                    help="frames in flight per GPU: frame i is replayed on HIP stream i %% streams, so the small "
                         "layers of one frame fill the CUs the other leaves idle")
    ap.add_argument("--frames", type=int, default=0, help="distinct resident frames per rank (default streams + 1; raised until "
                    "coprime with --streams)")
    ap.add_argument("--gather-every", type=int, default=8, help="frames per eval-collate block: one all-gather + one "
                    "device->host copy per block, on a dedicated stream")
    ap.add_argument("--regions", type=int, default=0, help="timed repetitions of the --steps region (0 = until ~1 s is timed)")
    ap.add_argument("--no-upload", action="store_true", help="skip the extra region that uploads every frame inside the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cu-mask", default="none", choices=["none", "xcd", "rows", "xcd2"],
                    help="experiment: give every frame stream its own share of the CUs (hipExtStreamCreateWithCUMask): 'xcd' = "
                         "whole XCDs per stream (mask bit i = CU i / 8 of XCD i %% 8), 'rows' = the same CU rows of every XCD")
    ap.add_argument("--only-timed", action="store_true", help="profiling aid (tools/pmc_timed.py): the runners' warm-up frames and "
                    "exactly --steps graph replays, nothing else; prints the frame counts")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip BASELINE configs[2] / configs[3] (extra_configs)")
    ap.add_argument("--no-drop-in", action="store_true", help="skip the legs that time model.test.test_net (drop_in, drop_in_uncertainty)")
    ap.add_argument("--drop-in-frames", type=int, default=240, help="frames per timed test_net call")
    ap.add_argument("--cpu-frames", type=int, default=3)
    ap.add_argument("--map-frames", type=int, default=8, help="frames (seeds 0..n-1, BASELINE configs[4]'s set) of the mAP-delta leg")
    ap.add_argument("--layers", action="store_true", help="print the per-layer conv table to stderr")
    ap.add_argument("--rehearse-collate", action="store_true",
                    help="control-flow rehearsal WITHOUT a GPU (CPU test of the N > 1 launcher): ranks exchange synthetic "
                         "records over gloo through the same step / gather / verify / report code; measures nothing")
    return ap.parse_args(argv)


def self_launch(args, argv):
    """``python bench.py --gpus N`` with N > 1 and no launcher around it: start the N ranks through
    ``torch.distributed.run`` as a CHILD process (one rank per GPU, RCCL rendezvous on 127.0.0.1) and hand its exit
    code back.  This parent never touches the GPU (no HIP call happens before this point) and is never replaced by
    exec; rank 0 of the child prints the JSON line on the inherited stdout."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
    env["FRCNN_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def device_for_rank(local_rank, backend, device_count):
    """The GPU a rank binds: cuda:LOCAL_RANK under RCCL (one process per GPU; the launcher hands out LOCAL_RANK 0..N-1);
    the gloo rehearsal on a box with fewer GPUs than ranks wraps around."""
    if backend == "nccl":
        return int(local_rank)
    return int(local_rank) % max(int(device_count), 1)


_MASKED_STREAMS = []


def masked_stream(device, k, n, mode, num_cu=256, num_xcd=8):
    """A HIP stream whose kernels only run on share k of n of the CUs (hipExtStreamCreateWithCUMask), as a torch stream."""
    if mode == "none":
        return torch.cuda.Stream(device=device)
    import ctypes
    from faster_rcnn_pytorch_multimodal_amd.model.train_graph import _hip_runtime
    hip = _hip_runtime()
    words = (num_cu + 31) // 32
    mask = [0] * words
    for i in range(num_cu):
        xcd, row = i % num_xcd, i // num_xcd          # mask bit i = CU `row` of XCD `xcd` (bits interleave the XCDs)
        if mode == "xcd2":                              # two halves of the chip, streams alternate between them
            mine = (xcd * 2 // num_xcd) == (k % 2)
        else:
            mine = (xcd * n // num_xcd == k) if mode == "xcd" else (row % n == k)
        if mine:
            mask[i // 32] |= 1 << (i % 32)
    arr = (ctypes.c_uint32 * words)(*mask)
    handle = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), ctypes.c_uint32(words), arr)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    _MASKED_STREAMS.append(handle)
    return torch.cuda.ExternalStream(handle.value, device=device)


def rehearsal_record(frame_id, numel):
    """Deterministic stand-in for a detection record (rehearsal mode only)."""
    return torch.arange(numel, dtype=torch.float32) * 0.5 + float(frame_id)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args, argv))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rehearsal = args.rehearse_collate
    if not rehearsal and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; this package has no CPU execution path")
    # FRCNN_BENCH_BACKEND=gloo is a REHEARSAL of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share
    # devices, the collated record goes through the host); the judged multi-GPU run uses the default: RCCL.
    backend = "gloo" if rehearsal else os.environ.get("FRCNN_BENCH_BACKEND", "nccl")
    # FRCNN_BENCH_FORCE_DIST=1: go through the process group / all-gather path even with one rank (RCCL smoke test)
    use_dist = world > 1 or rehearsal or os.environ.get("FRCNN_BENCH_FORCE_DIST") == "1"
    device = "cpu"
    if not rehearsal:
        dev_index = device_for_rank(local_rank, backend, torch.cuda.device_count())
        torch.cuda.set_device(dev_index)
        device = "cuda:%d" % dev_index
    dist = None
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:               # FORCE_DIST without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29531"),
                              RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
    gather_dev = device if backend == "nccl" else "cpu"

    from faster_rcnn_pytorch_multimodal_amd.model import collate
    numel = collate.record_numel(NUM_CLASSES, MAX_DETS)
    info = np.array([0, W, 0, H, 0, 0, 1.0], np.float32)
    n_streams = max(1, args.streams)
    # distinct resident frames, their number coprime with the number of runners: runner k then sees EVERY frame over the
    # steps, so a replay that silently did nothing would leave another frame's record behind and fail the verification
    n_resident = args.frames if args.frames > 0 else n_streams + 1
    while np.gcd(n_resident, n_streams) != 1:
        n_resident += 1
    gather_every = max(1, args.gather_every)
    net = sd = None
    distinct_queues = None
    if not rehearsal:
        from faster_rcnn_pytorch_multimodal_amd.model.frame_graph import FrameRunner
        from faster_rcnn_pytorch_multimodal_amd import ops as _ops
        net, sd = build_net(device)
        # rank r gets frames r, r+world, ... (BASELINE configs[4]: seeds 0..7 on 8 GPUs); a pinned host copy (what a loader
        # hands over, lib/model/test.py:197-206) and a copy resident in HBM
        frames_host = [synthetic_frame(rank + world * i) for i in range(n_resident)]
        frames_pinned = [torch.from_numpy(f).pin_memory() for f in frames_host]
        frames = [f.to(device) for f in frames_pinned]
        _ops.set_conv_algo(args.conv_algo)
        plans_loaded = False
        plans_path = args.plans
        if not plans_path and not args.retune and os.path.exists(os.path.join(ROOT, "profiles", PLANS_FILE)):
            # default: the committed plan table - the kernels the profiles under profiles/ measured (roofline.from_profiles)
            plans_path = os.path.join(ROOT, "profiles", PLANS_FILE)
        if plans_path and os.path.exists(plans_path):
            with open(plans_path) as f:
                _ops.import_conv_plans(json.load(f))
            plans_loaded = True
        runners = [FrameRunner(net, H, W, C, info, THRESH, MAX_DETS, use_graph=not args.no_graph) for _ in range(n_streams)]
        if args.cu_mask != "none" or os.environ.get("FRCNN_STREAM_POLICY", "measured") == "pool":
            streams = [masked_stream(device, k, n_streams, args.cu_mask) for k in range(n_streams)]
            distinct_queues = None
        else:
            # streams chosen by measurement: a HIP stream is not a hardware queue (model/streams.py)
            from faster_rcnn_pytorch_multimodal_amd.model.streams import concurrent_streams
            streams, distinct_queues = concurrent_streams(n_streams, device)
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())

    def new_ring(steps):
        return collate.RecordRing(numel, steps, every=gather_every, device=device, distributed=use_dist,
                                  gather_device=gather_dev, pin=not rehearsal)

    def step(i, ring, source):
        """One frame of this rank: upload (host source) or HBM-resident input -> hipGraph replay on stream i % S ->
        record packed into the device ring.  The collate (all-gather of a block of records + device->host copy) runs on
        the ring's own stream every `gather_every` frames."""
        if rehearsal:
            ring.slot(i).copy_(rehearsal_record(rank + world * (i % n_resident), numel))
            ring.commit(i)
            return
        k = i % n_streams
        with torch.cuda.stream(streams[k]):
            dets, counts = runners[k].run(source[i % n_resident], poison=True)
            collate.pack_record(dets, counts, ring.slot(i))
            ring.commit(i)

    def fence():
        if use_dist:
            dist.barrier()
        if not rehearsal:
            torch.cuda.synchronize()

    def region(steps, source):
        """EXACTLY `steps` steps between two fences (barrier + device synchronise).  Returns (elapsed MAX over ranks,
        this rank's own time to drain, host records (steps, ranks, numel))."""
        ring = new_ring(steps)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, ring, source)
        if not rehearsal:
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
        host = ring.drain()
        if not rehearsal:
            torch.cuda.synchronize()
        own = time.perf_counter() - t0
        fence()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, own, host, ring.gathers

    if args.only_timed:
        region(args.steps, frames)
        per_runner = 3 if not args.no_graph else 0      # FrameRunner: two warm-up frames + one with the tuned plans, eager
        print(json.dumps({"frames_total": args.steps + per_runner * n_streams, "frames_replayed": args.steps,
                          "streams": n_streams}))
        return
    source = None if rehearsal else frames
    warm_elapsed = None
    if args.warmup > 0:
        warm_elapsed, _, _, _ = region(args.warmup, source)
    if not rehearsal and args.plans and not plans_loaded and rank == 0:
        with open(args.plans, "w") as f:
            json.dump(_ops.export_conv_plans(), f)
    # the timed region is EXACTLY --steps steps; when that is short (the driver's 20 steps are ~0.1 s) it is repeated, every
    # repetition fenced on both sides, until about MIN_TIMED_S of timed work exists, and the MEDIAN repetition is reported
    n_regions = args.regions
    if n_regions <= 0:
        est = (warm_elapsed / args.warmup * args.steps) if warm_elapsed else 0.0
        n_regions = 1 if (rehearsal or est <= 0) else int(min(15, max(1, np.ceil(MIN_TIMED_S / est))))
        n_regions += 1 - n_regions % 2                       # odd: the median is a measured repetition
        if use_dist:
            t = torch.tensor([n_regions], dtype=torch.int64, device=gather_dev)
            dist.broadcast(t, src=0)
            n_regions = int(t.item())
    regions = [region(args.steps, source) for _ in range(n_regions)]
    order = sorted(range(n_regions), key=lambda r: regions[r][0])
    elapsed, own_elapsed, _, gathers = regions[order[n_regions // 2]]
    upload = None
    if not rehearsal and not args.no_upload:
        # same steps, every frame uploaded host -> device inside the step; median of (up to) three regions
        ups = sorted((region(args.steps, frames_pinned) for _ in range(min(3, n_regions))), key=lambda r: r[0])
        upload = ups[len(ups) // 2]
        upload_hosts = [u[2] for u in ups]
    per_rank = [own_elapsed]
    # what each rank is bound to (the rehearsal reports the device the RCCL run WOULD bind: cuda:LOCAL_RANK)
    rank_devices = [device if not rehearsal else "cuda:%d" % device_for_rank(local_rank, "nccl", 0)]
    rank_frames = [[rank + world * j for j in range(n_resident)]]
    if use_dist:
        gathered_dev, gathered_frames = [None] * world, [None] * world
        dist.all_gather_object(gathered_dev, rank_devices[0])
        dist.all_gather_object(gathered_frames, rank_frames[0])
        rank_devices, rank_frames = gathered_dev, gathered_frames
    if use_dist:
        own = torch.tensor([own_elapsed], dtype=torch.float64, device=gather_dev)
        allr = torch.zeros(world, dtype=torch.float64, device=gather_dev)
        dist.all_gather_into_tensor(allr, own)
        per_rank = [float(v) for v in allr.cpu()]

    # ---- verification of the timed records (untimed): every timed step's record must equal, bit for bit, what the
    # eager single-stream path returns for the same frame; with N > 1 the expected records of all ranks are exchanged
    # once more so that every rank checks every row the timed all-gathers delivered
    if rehearsal:
        expected = torch.stack([torch.stack([rehearsal_record(r + world * j, numel) for j in range(n_resident)])
                                for r in range(world)])
    else:
        from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device
        own_expected = torch.zeros((n_resident, numel), device=device)
        for j in range(n_resident):
            dets, counts = detect_frame_device(net, frames[j], info, THRESH, MAX_DETS, MAX_DETS)
            collate.pack_record(dets, counts, own_expected[j])
        torch.cuda.synchronize()
        if use_dist:
            expected = collate.gather_records((own_expected if backend == "nccl" else own_expected.cpu()).view(-1))
            expected = expected.view(world, n_resident, numel).cpu()
        else:
            expected = own_expected.cpu().unsqueeze(0)
    distinct = len({expected[0, j].numpy().tobytes() for j in range(n_resident)})
    checked = [r[2] for r in regions] + (upload_hosts if upload else [])
    bad = [(ri, i) for ri, host in enumerate(checked) for i in range(args.steps)
           if not torch.equal(host[i], expected[:, i % n_resident, :])]
    if use_dist:
        flag = torch.tensor([len(bad)], dtype=torch.int64, device=gather_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.SUM)
        n_bad = int(flag.item())
    else:
        n_bad = len(bad)
    if n_bad or (distinct < 2 and n_resident > 1):
        print("bench.py: rank %d: %d timed records differ from the eager path (first at (region, step) %s); %d distinct "
              "expected records" % (rank, len(bad), bad[:8], distinct), file=sys.stderr)
        if use_dist:
            dist.destroy_process_group()
        raise SystemExit(3)
    import hashlib
    host_rec = regions[order[n_regions // 2]][2]
    checksum = hashlib.sha1(host_rec.numpy().tobytes()).hexdigest()[:16]
    dets_last = host_rec[args.steps - 1, 0 if not use_dist else rank]
    verification = {"timed_steps_checked": args.steps, "regions_checked": len(checked), "equal_to_eager_path": True,
                    "records_sha1_16": checksum,
                    "distinct_frames_per_rank": n_resident, "distinct_expected_records": distinct,
                    "frame_rotation": "step i replays frame i %% %d on runner i %% %d (coprime): every runner sees every frame, "
                                      "and its output buffers are overwritten with NaN / -1 before each replay, so a replay that "
                                      "did not execute cannot pass" % (n_resident, n_streams),
                    "detections_per_class_last_step": [int(v) for v in dets_last[NUM_CLASSES * MAX_DETS * 5:]],
                    "what": "every timed step's record of every timed region (collated in blocks of %d frames and copied "
                            "device->host inside the timed region) equals the eager single-stream record of the same frame, "
                            "bit for bit; tests/test_timed_path.py compares the same arrangement with the CPU oracle"
                            % gather_every}

    # ---- the collective on its own (N > 1): HIP-event timing of the all-gather of one record on one stream
    allgather_us = None
    if use_dist and not rehearsal and backend == "nccl":
        reps = 50
        blk = torch.zeros(gather_every * numel, device=device)
        out_blk = torch.zeros((world, gather_every * numel), device=device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(5):
            collate.gather_records(blk, out_blk)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            collate.gather_records(blk, out_blk)
        e1.record()
        torch.cuda.synchronize()
        allgather_us = 1e3 * e0.elapsed_time(e1) / reps

    out = None
    if rank == 0:
        out = {
            "metric": "frames/sec res101 Faster-RCNN 1000x600", "value": world * args.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: res101 image Faster-RCNN forward, 1x(1000x600) synthetic "
                                   "frame per GPU per step, 6000 pre-NMS / 300 proposals, RoIAlign 7x7, per-class NMS, "
                                   "thresh %.1f max_dets %d; weights seeded random init (BN tame)" % (THRESH, MAX_DETS),
                       "frames_per_step": world, "parallelism": "frame-sharded x%d, all-gather of detections" % world,
                       "launch": "eager" if args.no_graph else "hipGraph replay", "frames_in_flight": n_streams,
                       "streams_on_distinct_hardware_queues": distinct_queues,
                       "input": "frames resident in HBM before the timed region (with_host_upload: the same steps with every "
                                "frame uploaded from pinned host memory inside the step)"},
            "timed_regions": {"count": n_regions, "reported": "median", "steps_each": args.steps,
                              "ms_each": [1e3 * regions[r][0] for r in range(n_regions)],
                              "why": "a region is EXACTLY --steps steps between two fences; it is repeated until ~%.1f s "
                                     "of timed work exists so that a 20-step run is not a 0.1 s sample" % MIN_TIMED_S},
            "verification": verification,
        }
        if upload is not None:
            out["with_host_upload"] = {"value": world * args.steps / upload[0], "unit": "frames/s",
                                       "ms_per_step": 1e3 * upload[0] / args.steps,
                                       "bytes_per_frame": int(frames_pinned[0].numel() * 4),
                                       "what": "median of up to 3 regions of --steps steps; each step copies its frame from pinned host memory "
                                               "on the frame's stream (lib/model/test.py:75 hands test_frame a host blob)"}
        if use_dist:
            out["collective"] = {"backend": dist.get_backend(), "is_rccl": dist.get_backend() == "nccl",
                                 "rccl_ranks": dist.get_world_size() if dist.get_backend() == "nccl" else 0,
                                 "ranks": dist.get_world_size(),
                                 "per_rank_frames_per_s": [args.steps / v for v in per_rank],
                                 "rank_devices": rank_devices, "rank_frame_seeds": rank_frames,
                                 "gather_every_frames": gather_every, "allgathers_per_region": gathers,
                                 "allgather_us_per_block": allgather_us,
                                 "allgather_us_per_step": (allgather_us / gather_every) if allgather_us else None,
                                 "record_bytes_per_rank": 4 * numel, "block_bytes_per_rank": 4 * numel * gather_every,
                                 "collate": "records accumulate in a device ring; ONE all_gather_into_tensor per block of "
                                            "frames on a dedicated stream (no frame stream waits for it), then one "
                                            "device->host copy of the collated block",
                                 "self_launched": os.environ.get("FRCNN_BENCH_SELF_LAUNCHED") == "1"}
        if rehearsal:
            out["data"] = "rehearsal: synthetic records over gloo, no device work - NOT a measurement"
            out["value"] = None
        else:
            conv = conv_roofline(net, frames[0], info, steps=min(args.steps, 5))
            run_plans = _ops.export_conv_plans()
            run_sha = plans_sha(run_plans)
            prof = from_profiles(run_sha)
            # ALGORITHMIC FLOPs of a frame's convolutions = SURVEY.md section 8(d) / BASELINE.md section 3: 628.4 GFLOP, the
            # reference's formulation (direct-form 3x3 convolutions, layer4[0].conv1 / downsample[0] on the pooled RoIs).  The
            # launches of this build do less arithmetic for the same function (Winograd; those two convolutions moved in front of
            # the pooling): *_launched counts the direct-form FLOPs of the convolutions as launched, frac_executed what the matrix
            # pipe really multiplied.
            from faster_rcnn_pytorch_multimodal_amd.nets import network as _N
            # reference-order FLOPs derived from the layer table: what was launched + the two 1x1 convolutions of layer4[0]
            # evaluated on the 300 x 7 x 7 pooled pixels instead of the feature map (BASELINE.md section 3 gives 628.4e9)
            algorithmic = conv["flops_per_frame"]
            if _N.PROJECT_BEFORE_POOLING:
                blk = net.resnet.layer4[0]
                macs = blk.conv1.in_channels * (blk.conv1.out_channels + blk.downsample[0].out_channels)
                fh, fw = net._act_summaries["conv"].shape[1:3]
                algorithmic += 2.0 * macs * (int(Cfg_post_nms()) * 49 - fh * fw)
            assert abs(algorithmic - REFERENCE_ORDER_FLOPS) < 0.005 * REFERENCE_ORDER_FLOPS, (algorithmic, REFERENCE_ORDER_FLOPS)
            achieved = algorithmic / (conv["ms_per_frame"] * 1e-3) / 1e12
            achieved_launched = conv["flops_per_frame"] / (conv["ms_per_frame"] * 1e-3) / 1e12
            timed_tflops = algorithmic / (1e-3 * 1e3 * elapsed / args.steps) / 1e12
            timed_launched = conv["flops_per_frame"] / (1e-3 * 1e3 * elapsed / args.steps) / 1e12
            out["roofline"] = {
                "bound": "mfma", "kernel": "frcnn_conv2d_fwd: conv_igemm_* (all instantiations) + split-K second passes + Winograd "
                "transforms, %d calls/frame" % round(conv["launches_per_frame"]), "achieved": achieved,
                "peak": MFMA_F32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": prof["conv_traffic_bytes_per_call"],
                "traffic_source": "from_profiles (NOT measured by this run; null when the profile's plan table differs from this run's)",
                "from_profiles": prof,
                "plans": {"source": ("profiles/" + PLANS_FILE if (plans_loaded and not args.plans) else
                                     (args.plans if plans_loaded else "tuned in this run (autotune level %d)" % _ops.AUTOTUNE_LEVEL)),
                          "sha": run_sha, "entries": len(run_plans)},
                "mode": "isolated kernels: eager launches on one stream, every dispatch bracketed by its own start / stop HIP "
                        "events (hipExtLaunchKernelGGL) = rocprofv3's per-dispatch duration; avg_launch_us is per "
                        "frcnn_conv2d_fwd call (every launch it makes); the timed run overlaps %d frames, see "
                        "frac_timed" % n_streams,
                "achieved_launched": achieved_launched, "frac_launched": achieved_launched / MFMA_F32_PEAK_TFLOPS,
                "algorithmic_flops_per_frame": algorithmic,
                "achieved_timed_algorithmic": timed_tflops,
                "frac_timed_algorithmic_not_pipe_utilisation": timed_tflops / MFMA_F32_PEAK_TFLOPS,
                "frac_timed_launched": timed_launched / MFMA_F32_PEAK_TFLOPS,
                "frac_timed_what": "ALGORITHMIC conv FLOPs of a frame (628.4 GFLOP, the reference's formulation) / ms_per_step of "
                                   "the TIMED run (hipGraph x %d streams) / peak.  NOT the matrix pipe's utilisation: Winograd and "
                                   "the head reorder make the pipe execute fewer FLOPs - that figure is frac_executed_timed" % n_streams,
                "frac_executed_timed": conv["executed_flops_per_frame"] / (1e-3 * 1e3 * elapsed / args.steps) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                "frac_executed_timed_what": "FLOPs the matrix pipe EXECUTES per frame (Winograd layers: 16/36 of the direct form) / "
                                            "ms_per_step of the timed run / peak: the MFMA pipe's utilisation in the timed mode; "
                                            "achieved / frac / frac_timed above are ALGORITHMIC (reference-order, direct-form) rates",
                "conv_calls_per_frame": int(round(conv["launches_per_frame"])),
                "flops_per_frame": conv["flops_per_frame"], "kernel_ms_per_frame": conv["ms_per_frame"],
                "avg_launch_us": 1e3 * conv["ms_per_frame"] / conv["launches_per_frame"],
                "main_kernel_avg_us": conv["main_kernel_avg_us"],
                "second_pass_launches_per_frame": conv["second_pass_launches_per_frame"],
                "second_pass_avg_us": conv["second_pass_avg_us"],
                "flops_what": "achieved / frac / frac_timed count the ALGORITHMIC FLOPs of the frame's convolutions (628.4 GFLOP: direct "
                              "form, the reference's order of operations - SURVEY.md section 8(d), BASELINE.md section 3); "
                              "*_launched count flops_per_frame, the direct-form FLOPs of the convolutions as launched; "
                              "%d of the %d calls per frame ran as Winograd F(2x2,3x3) (autotuned; same fp32 arithmetic, "
                              "2.25x fewer multiplications), so the MFMA pipe executed executed_flops_per_frame: "
                              "frac_executed is its utilisation" % (round(conv["winograd_calls_per_frame"]),
                                                                    round(conv["launches_per_frame"])),
                "reference_order_flops_per_frame": REFERENCE_ORDER_FLOPS,
                "head_order": "layer4[0].conv1 and layer4[0].downsample[0] (bias-free 1x1) run on the 38x63 feature map BEFORE "
                              "the RoIAlign instead of on 300x7x7 pooled pixels after it (pooling and a 1x1 convolution "
                              "commute; BatchNorm + ReLU stay behind the pooling, in the RoIAlign epilogue): flops_per_frame "
                              "counts the convolutions as LAUNCHED, i.e. %.1f GFLOP less than the reference's order of "
                              "operations (reference_order_flops_per_frame = algorithmic_flops_per_frame)"
                              % ((REFERENCE_ORDER_FLOPS - conv["flops_per_frame"]) / 1e9),
                "executed_flops_per_frame": conv["executed_flops_per_frame"],
                "frac_executed": conv["executed_flops_per_frame"] / (1e-3 * conv["ms_per_frame"]) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                "winograd_calls_per_frame": conv["winograd_calls_per_frame"],
                "winograd_transform_launches_per_frame": conv["winograd_transform_launches_per_frame"],
                "winograd_transform_us_per_frame": conv["winograd_transform_us_per_frame"]}
            busy = prof["mfma_busy_ms_per_frame_timed_mode"]
            if busy is not None:
                prof["mfma_util_timed_percent"] = 100.0 * busy / (1e3 * elapsed / args.steps)
                prof["mfma_util_timed_what"] = (
                    "matrix-pipe busy time per frame (%.3f ms: sum over the frame's dispatches of MfmaUtil x duration, rocprofv3 "
                    "PMC pass over the hipGraph x %d-stream mode, profiles/%s) / ms_per_step of THIS run" % (busy, n_streams, PMC_TIMED_FILE))
            # the per-layer table of the isolated pass (what DESIGN.md's kernel section cites), largest share first
            out["roofline"]["per_layer"] = [
                {"shape": k, "calls_per_frame": v["calls_per_frame"], "us_per_call": round(v["us_per_call"], 2),
                 "tflops": round(v["tflops"], 1)}
                for k, v in sorted(conv["per_layer"].items(), key=lambda kv: -kv[1]["us_per_call"] * kv[1]["calls_per_frame"])]
            if args.layers:
                for k, v in sorted(conv["per_layer"].items(), key=lambda kv: -kv[1]["us_per_call"] * kv[1]["calls_per_frame"]):
                    print("%-40s x%-4.0f %9.1f us/call %7.1f TFLOP/s" % (k, v["calls_per_frame"], v["us_per_call"], v["tflops"]),
                          file=sys.stderr)
            out["roofline_roi_align"] = roi_align_timing(net, 20, prof)
            out["roofline_nms"] = nms_timing(net, 20)
            if world == 1 and not args.no_drop_in:
                out["drop_in"] = drop_in_timing(net, frames, info, max(args.steps, args.drop_in_frames), out["value"])
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"], _ = cpu_baseline(sd, frames_host[:args.cpu_frames], info)
                out["map_delta_vs_cpu"] = map_delta(net, sd, [synthetic_frame(i) for i in range(args.map_frames)], info)
            if world == 1 and not args.no_drop_in:
                del runners, net
                torch.cuda.empty_cache()
                out["drop_in_uncertainty"] = drop_in_uncertainty(device, max(args.steps, args.drop_in_frames),
                                                                 frames_host, info)
    if rank == 0 and world == 1 and not rehearsal and not args.no_extra_configs:
        torch.cuda.empty_cache()
        out["extra_configs"] = extra_configs(args.steps)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
