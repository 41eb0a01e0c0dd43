#!/usr/bin/env python
"""Per-shape table of the convolutions of TODAY's frame (the shapes `Network.forward` launches for one 1000x600 frame of
bench.py's net: head reorder, fused projections, fused RPN head included), per candidate plan, ALONE and with FOUR copies in
flight on four HIP streams (every stream replays a hipGraph of `reps` launches on its own buffers: the saturated chip the
four-frames-in-flight schedule of model/frame_graph.FramePool sees).  Tuning aid and evidence; bench.py stays the judged
measurement.

    python tools/conv_shape_table.py [--reps 12] [--only substr] [--out profiles/r05_conv_per_shape.md] [--json file]

Columns per candidate: us per call alone, us per call with 4 in flight (wall time of the 4 x reps launches / (4 x reps)),
TFLOP/s (direct-form FLOPs) of both, and `load/alone`: 1.0 = the chip was already full with one copy, 0.25 = four copies
ran in the time of one.  Per shape the winner of each column is marked; the frame totals at the end compare three plan
tables: best-alone (what frcnn_conv2d_set_autotune(1) picks), best-under-load (set_autotune(2)), and per fixed tile.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from faster_rcnn_pytorch_multimodal_amd.model.frame_graph import capture  # noqa: E402

TILES = ["256x128", "128x256", "128x128", "128x64", "64x128", "64x64", "128x128d2", "64x64buf", "128x64buf", "64x128buf", "128x128buf", "256x128buf", "128x256buf", "64x64pers"]   # kTiles order (conv_igemm.hip)
BK = 32


def frame_shapes(dev):
    """(shape dict incl. residual flag) -> calls per frame, in first-seen order, from one eager frame of bench.py's net."""
    import bench
    from faster_rcnn_pytorch_multimodal_amd import ops
    from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device
    net, _ = bench.build_net(dev)
    info = np.array([0, bench.W, 0, bench.H, 0, 0, 1.0], np.float32)
    frame = torch.from_numpy(bench.synthetic_frame(0)).to(dev)
    detect_frame_device(net, frame, info, bench.THRESH, bench.MAX_DETS, bench.MAX_DETS)
    torch.cuda.synchronize()
    ops.PROFILE = []
    detect_frame_device(net, frame, info, bench.THRESH, bench.MAX_DETS, bench.MAX_DETS)
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    out = {}
    for p in prof:
        key = (p["n"], p["h"], p["w"], p["c"], p["k"], p["r"], p["stride"], p["pad"], p["residual"], p["relu"])
        out[key] = out.get(key, 0) + 1
    del net
    torch.cuda.empty_cache()
    return out


def candidates(M, c, k, r, wino_ok):
    ksteps = (r * r * c + BK - 1) // BK
    cands = []
    for cfg in range(len(TILES)):
        for sp in (1, 2, 4, 8):
            if sp > 1 and ksteps // sp < 2:
                continue
            sps = (ksteps + sp - 1) // sp
            if (ksteps + sps - 1) // sps != sp:
                continue
            if sp > 1 and sp * M * k * 4 > (256 << 20):
                continue
            cands.append(("%s%s" % (TILES[cfg], "/k%d" % sp if sp > 1 else ""), cfg, sp, sps))
    if wino_ok:
        ws = (c + BK - 1) // BK
        for cfg in range(len(TILES)):
            cands.append(("wino+" + TILES[cfg], cfg + 16, 1, ws))
        if c % BK == 0:
            cands.append(("wino-fused 64x64", 5 + 32, 1, c // BK))
    return cands


def time_graphs(graphs, streams, reps):
    best = 1e30
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st, gr in zip(streams, graphs):
            with torch.cuda.stream(st):
                gr.replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return 1e6 * best / (reps * len(graphs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--plan", default="", help="only candidate plans whose label contains one of these comma-separated substrings")
    args = ap.parse_args()
    from faster_rcnn_pytorch_multimodal_amd import _hip, ops
    lib = _hip.load()
    dev = "cuda:0"
    shapes = frame_shapes(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    S = args.streams
    streams = [torch.cuda.Stream() for _ in range(S)]
    lines, table = [], []
    tot = {"alone": 0.0, "load": 0.0}
    per_tile_tot = {}
    flops_tot = 0.0
    lines.append("| shape (n x h x w, c -> k, r/stride, +res) | calls | plan | alone us | 4-in-flight us | alone TF/s | loaded TF/s | load/alone |")
    lines.append("|---|---|---|---|---|---|---|---|")
    for (n, h, w, c, k, r, stride, pad, res, relu), calls in shapes.items():
        name = "%dx%dx%d c%d k%d r%d/%d%s" % (n, h, w, c, k, r, stride, " +res" if res else "")
        if args.only and args.only not in name:
            continue
        ho, wo = ops.conv_out_hw(h, w, r, r, stride, pad)
        M = n * ho * wo
        fl = 2.0 * M * k * r * r * c
        wino_ok = (not res) and ops.winograd_eligible(k, r, r, c, stride, pad)
        wt = (torch.randn((k, r, r, c), generator=g) * 0.05).to(dev)
        u = ops.winograd_filter(wt) if wino_ok else None
        sc = (torch.rand((k,), generator=g) + 0.5).to(dev)
        sh = torch.randn((k,), generator=g).to(dev)
        xs = [torch.randn((n, h, w, c), generator=g).to(dev) for _ in range(S)]
        ys = [torch.empty((n, ho, wo, k), device=dev) for _ in range(S)]
        rss = [torch.randn((n, ho, wo, k), generator=g).to(dev) if res else None for _ in range(S)]
        key = [n, h, w, c, k, r, r, stride, pad, 1 + (256 if res else 0)]
        rows = []
        for label, code, sp, sps in candidates(M, c, k, r, wino_ok):
            if args.plan and not any(t == label or (t.endswith("*") and t[:-1] in label) for t in args.plan.split(",")):
                continue
            try:
                ops.import_conv_plans([key + [code, sp, sps]])
            except Exception as e:          # a plan the library refuses for this shape
                continue
            try:
                graphs = []
                for st, xi, yi, ri in zip(streams, xs, ys, rss):
                    st.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(st):
                        ops.conv2d_nhwc(xi, wt, sc, sh, ri, stride=stride, pad=pad, relu=relu, out=yi, w_winograd=u if code >= 16 else None)
                    torch.cuda.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with capture(gr, stream=st):
                        for _ in range(args.reps):
                            ops.conv2d_nhwc(xi, wt, sc, sh, ri, stride=stride, pad=pad, relu=relu, out=yi,
                                            w_winograd=u if code >= 16 else None)
                    graphs.append(gr)
                alone = time_graphs(graphs[:1], streams[:1], args.reps)
                load = time_graphs(graphs, streams, args.reps)
            except Exception as e:
                print("skip %s %s: %s" % (name, label, e), file=sys.stderr)
                continue
            rows.append((label, code, sp, alone, load))
            del graphs
        if not rows:
            continue
        ba = min(rows, key=lambda t: t[3])
        bl = min(rows, key=lambda t: t[4])
        for label, code, sp, alone, load in rows:
            mark = ("**A**" if label == ba[0] else "") + ("**L**" if label == bl[0] else "")
            lines.append("| %s | %d | %s %s | %.1f | %.1f | %.1f | %.1f | %.2f |" % (
                name, calls, label, mark, alone, load, fl / alone / 1e6, fl / load / 1e6, load / alone))
            table.append({"shape": name, "calls": calls, "plan": label, "code": code, "splits": sp, "alone_us": alone,
                          "load_us": load, "flops": fl})
            if sp == 1 and code < 16:
                t = per_tile_tot.setdefault(label, [0.0, 0.0, 0])
                t[0] += alone * calls
                t[1] += load * calls
                t[2] += calls
        # frame totals: loaded time of the plan that wins ALONE vs of the plan that wins UNDER LOAD
        tot["alone"] += ba[4] * calls
        tot["load"] += bl[4] * calls
        tot.setdefault("alone_alone", 0.0)
        tot["alone_alone"] += ba[3] * calls
        tot.setdefault("load_alone", 0.0)
        tot["load_alone"] += bl[3] * calls
        flops_tot += fl * calls
        print("%-44s x%-3d best alone %-22s %7.1f us (loaded %7.1f) | best loaded %-22s %7.1f us (alone %7.1f)" % (
            name, calls, ba[0], ba[3], ba[4], bl[0], bl[4], bl[3]), flush=True)
    lines.append("")
    lines.append("Frame totals over these shapes (direct-form %.1f GFLOP):" % (flops_tot / 1e9))
    lines.append("")
    lines.append("| plan table | conv us per frame, one copy alone | conv us per frame, 4 in flight | loaded TFLOP/s |")
    lines.append("|---|---|---|---|")
    lines.append("| per shape the plan fastest ALONE (autotune level 1) | %.0f | %.0f | %.1f |" % (
        tot.get("alone_alone", 0), tot["alone"], flops_tot / max(tot["alone"], 1e-9) / 1e6))
    lines.append("| per shape the plan fastest UNDER LOAD (autotune level 2) | %.0f | %.0f | %.1f |" % (
        tot.get("load_alone", 0), tot["load"], flops_tot / max(tot["load"], 1e-9) / 1e6))
    all_calls = sum(shapes.values())
    for label, (a, l, cnt) in per_tile_tot.items():
        if cnt == all_calls or not args.only:
            lines.append("| tile %s everywhere, implicit GEMM, no split (%d of %d calls) | %.0f | %.0f | - |" % (label, cnt, all_calls, a, l))
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write("# Convolution shapes of one frame: every candidate plan alone and with 4 copies in flight\n\n"
                    "`python tools/conv_shape_table.py --reps %d` on one MI355X; **A** = fastest alone, **L** = fastest with four copies "
                    "in flight (hipGraph replay on four streams, own buffers each).  us are per call; `load/alone` = 0.25 means four "
                    "copies ran in the time of one (idle CUs were available), 1.0 means one copy already filled the chip.\n\n" % args.reps)
            f.write(text + "\n")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(table, f)
    lib.frcnn_conv2d_clear_plans()


if __name__ == "__main__":
    main()
