#!/usr/bin/env python
"""Launch a fixed list of convolution cases (shape x forced plan x frames per launch) a few times each, eagerly, so that
`rocprofv3 --kernel-trace --pmc <counters>` sees one dispatch per launch; tools/conv_pmc_summary.py folds the counter CSVs.
A batch of 4 frames in ONE launch stands for the saturated chip of the four-frames-in-flight schedule (the counters are per
dispatch, so four concurrent launches could not be told apart).

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES ... --output-format csv -d out -o pmc -- python3 tools/conv_pmc_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name, n, h, w, c, k, r, stride, pad, residual, plan code (tile index, +16 Winograd, +32 fused Winograd), splits
CASES = [
    ("l3 1x1 256-1024+res b4 64x64", 4, 38, 63, 256, 1024, 1, 1, 0, True, 5, 1),
    ("l3 1x1 256-1024+res b4 128x64", 4, 38, 63, 256, 1024, 1, 1, 0, True, 3, 1),
    ("l3 1x1 256-1024+res b4 128x128", 4, 38, 63, 256, 1024, 1, 1, 0, True, 2, 1),
    ("l3 1x1 256-1024+res b1 64x64", 1, 38, 63, 256, 1024, 1, 1, 0, True, 5, 1),
    ("l3 1x1 1024-256 b4 64x64", 4, 38, 63, 1024, 256, 1, 1, 0, False, 5, 1),
    ("l3 1x1 1024-256 b4 128x64", 4, 38, 63, 1024, 256, 1, 1, 0, False, 3, 1),
    ("l3 1x1 1024-256 b1 64x64", 1, 38, 63, 1024, 256, 1, 1, 0, False, 5, 1),
    ("l3 3x3 256 b4 wino-fused", 4, 38, 63, 256, 256, 3, 1, 1, False, 5 + 32, 1),
    ("l3 3x3 256 b4 wino+64x64", 4, 38, 63, 256, 256, 3, 1, 1, False, 5 + 16, 1),
    ("l3 3x3 256 b4 direct 64x64", 4, 38, 63, 256, 256, 3, 1, 1, False, 5, 1),
    ("l4 1x1 512-2048+res 64x64", 300, 7, 7, 512, 2048, 1, 1, 0, True, 5, 1),
    ("l4 1x1 512-2048+res 128x128d2", 300, 7, 7, 512, 2048, 1, 1, 0, True, 6, 1),
    ("l4 1x1 512-2048+res 256x128", 300, 7, 7, 512, 2048, 1, 1, 0, True, 0, 1),
    ("l4 1x1 2048-512 128x256", 300, 7, 7, 2048, 512, 1, 1, 0, False, 1, 1),
    ("l4 1x1 2048-512 64x64", 300, 7, 7, 2048, 512, 1, 1, 0, False, 5, 1),
    ("l4 3x3 512 wino+128x128d2", 300, 7, 7, 512, 512, 3, 1, 1, False, 6 + 16, 1),
    ("l4 3x3 512 wino+256x128", 300, 7, 7, 512, 512, 3, 1, 1, False, 0 + 16, 1),
]
REPS = 4


def main():
    from faster_rcnn_pytorch_multimodal_amd import ops
    dev = "cuda:0"
    g = torch.Generator(device="cpu").manual_seed(0)
    BK = 32
    order = []
    for name, n, h, w, c, k, r, stride, pad, res, code, sp in CASES:
        ho, wo = ops.conv_out_hw(h, w, r, r, stride, pad)
        x = torch.randn((n, h, w, c), generator=g).to(dev)
        wt = (torch.randn((k, r, r, c), generator=g) * 0.05).to(dev)
        sc = (torch.rand((k,), generator=g) + 0.5).to(dev)
        sh = torch.randn((k,), generator=g).to(dev)
        rs = torch.randn((n, ho, wo, k), generator=g).to(dev) if res else None
        y = torch.empty((n, ho, wo, k), device=dev)
        u = ops.winograd_filter(wt) if code >= 16 else None
        ksteps = (c + BK - 1) // BK if code >= 16 else (r * r * c + BK - 1) // BK
        sps = (ksteps + sp - 1) // sp
        ops.import_conv_plans([[n, h, w, c, k, r, r, stride, pad, 1 + (256 if res else 0), code, sp, sps]])
        for _ in range(REPS):
            ops.conv2d_nhwc(x, wt, sc, sh, rs, stride=stride, pad=pad, relu=True, out=y, w_winograd=u)
        torch.cuda.synchronize()
        order.append((name, 2.0 * n * ho * wo * k * r * r * c))
    import json
    print(json.dumps({"cases": order, "reps": REPS}))


if __name__ == "__main__":
    main()
