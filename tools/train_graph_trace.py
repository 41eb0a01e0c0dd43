#!/usr/bin/env python
"""Find the FIRST kernel call of a single-chain captured training step whose result differs between two replays.

    python tools/train_graph_trace.py [--h 600 --w 1000] [--side]

Every function of ``faster_rcnn_pytorch_multimodal_amd.ops`` is wrapped for the duration of the capture: after the call a
checksum (sum of the returned / accumulated float tensors) is written into slot i of a device-side trace buffer, i = call
number.  The checksum launches are part of the captured chain, so each replay leaves the checksums of ITS OWN
intermediate tensors; comparing the trace of replay 1 with that of replay 2 names the first call that went wrong.
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
    ap.add_argument("--replays", type=int, default=3)
    ap.add_argument("--side", action="store_true")
    ap.add_argument("--memops", type=int, default=0, help="frcnn_set_memops_mode: 1 = hipMemsetAsync / hipMemcpyAsync nodes")
    ap.add_argument("--no-trace", action="store_true", help="no checksum launches in the chain: gradients only")
    args = ap.parse_args()
    from faster_rcnn_pytorch_multimodal_amd import ops
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.model import train_graph
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    from faster_rcnn_pytorch_multimodal_amd import _hip
    _hip.check(_hip.load().frcnn_set_memops_mode(args.memops), "frcnn_set_memops_mode")
    print("DEBUG_CLR_GRAPH_PACKET_CAPTURE=%r inline=%s memops_mode=%d trace=%s" % (
        os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), not args.side, args.memops, not args.no_trace), flush=True)
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
    params = [p for p in net.parameters() if p.requires_grad]
    grads = [torch.zeros_like(p) for p in params]

    trace = torch.zeros(8192, dtype=torch.float32, device="cuda:0")
    names = []
    state = {"on": False, "i": 0}
    originals = {}

    def checksum(t):
        # 4096 samples strided over the tensor: a one-workgroup reduction (a global reduction would add a memset node for
        # its semaphores to the chain under test)
        v = t.detach().reshape(-1)
        if v.numel() == 0:
            return torch.zeros((1,), device="cuda:0")
        step = max(1, v.numel() // 4096)
        return v[::step][:4096].float().abs().sum().view(1)

    def wrap(name, fn):
        def inner(*a, **kw):
            out = fn(*a, **kw)
            if state["on"]:
                tensors = []
                if isinstance(out, torch.Tensor):
                    tensors = [out]
                elif isinstance(out, (tuple, list)):
                    tensors = [t for t in out if isinstance(t, torch.Tensor)]
                elif isinstance(out, dict):
                    tensors = [t for t in out.values() if isinstance(t, torch.Tensor)]
                elif hasattr(out, "__dict__"):
                    tensors = [t for t in vars(out).values() if isinstance(t, torch.Tensor)]
                if name.startswith("conv2d_bwd_weight_acc"):
                    g = a[4]
                    tensors = list(g) if isinstance(g, (list, tuple)) else [g]
                tensors = [t for t in tensors if t.is_floating_point() and t.is_cuda]
                if tensors and state["i"] < trace.numel():
                    with torch.no_grad():
                        s = checksum(tensors[0])
                        for t in tensors[1:]:
                            s = s + checksum(t)
                        trace[state["i"]:state["i"] + 1].mul_(0.0).add_(s)        # kernels only (copy_ would be a memcpy node)
                    if len(names) <= state["i"]:
                        names.append("%s %s" % (name, [tuple(t.shape) for t in tensors][:2]))
                    state["i"] += 1
            return out
        return inner

    for name, fn in list(vars(ops).items()):
        if callable(fn) and not name.startswith("_") and getattr(fn, "__module__", "") == ops.__name__ and name not in (
                "set_conv_autotune", "set_conv_algo", "conv_plan_algo", "winograd_filter_wanted", "conv_out_hw",
                "winograd_eligible", "dgrad_winograd_wanted", "flops_begin", "flops_end", "nms_suppress_at_equal",
                "export_conv_plans", "import_conv_plans", "conv_profile_begin", "conv_profile_end"):
            originals[name] = fn
            setattr(ops, name, wrap(name, fn))

    real_step = train_graph.TrainStepRunner._step
    capture_calls = {"n": 0}

    def traced_step(self):
        # trace only while capturing (the warm-up steps run untraced)
        state["on"] = torch.cuda.is_current_stream_capturing() and not args.no_trace
        state["i"] = 0
        try:
            return real_step(self)
        finally:
            if state["on"]:
                capture_calls["n"] = state["i"]
            state["on"] = False

    train_graph.TrainStepRunner._step = traced_step
    runner = train_graph.TrainStepRunner(net, h, w, 3, 8, info, grads=grads, inline=not args.side)
    n = capture_calls["n"]
    print("traced %d kernel calls in the captured step; graph: %d nodes, %d edges, kinds %s" % (
        n, runner.nodes, runner.edges, runner.node_kinds), flush=True)
    hist = {}
    for nm in names:
        hist[nm.split()[0]] = hist.get(nm.split()[0], 0) + 1
    print("   " + ", ".join("%s x%d" % kv for kv in sorted(hist.items())), flush=True)
    traces, incs = [], []
    for r in range(args.replays):
        for g in grads:
            g.zero_()
        trace.zero_()
        torch.manual_seed(1234)
        runner.run(blobs)
        torch.cuda.synchronize()
        traces.append(trace[:n].cpu().numpy().copy())
        incs.append([g.clone() for g in grads])
    rc = 0
    pnames = [k for k, p in net.named_parameters() if p.requires_grad]
    for r in range(1, args.replays):
        bad_p = [(k, float((a - b).abs().max()) / (float(a.abs().max()) or 1.0)) for k, a, b in zip(pnames, incs[0], incs[r])]
        bad_p = [(k, d) for k, d in bad_p if not d <= 1e-3]
        print("replay %d: %d of %d gradient increments differ from replay 1's%s" % (
            r + 1, len(bad_p), len(pnames), (": first " + ", ".join("%s %.1e" % t for t in bad_p[:3])) if bad_p else ""))
    for r in range(1, args.replays):
        a, b = traces[0], traces[r]
        rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-20)
        rel[~np.isfinite(rel)] = np.inf
        bad = np.where(rel > 1e-3)[0]
        if len(bad) == 0:
            print("replay %d: all %d checksums equal replay 1's" % (r + 1, n))
            continue
        rc = 1
        print("replay %d: %d of %d checksums differ; first at call %d" % (r + 1, len(bad), n, bad[0]))
        lo = max(0, bad[0] - 6)
        for i in range(lo, min(n, bad[0] + 10)):
            print("   %s call %4d  %-70s replay1 %.6e  replay%d %.6e" % ("*" if i in set(bad) else " ", i, names[i][:70], a[i], r + 1, b[i]))
    C.reset_cfg()
    return rc


if __name__ == "__main__":
    raise SystemExit(main())
