"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (read-only, /root/reference).

Run only in the build container (the reference never travels to the GPU box):

    python -B tests/golden/make_golden.py

What can be imported: layer_utils.generate_anchors / snippets, model.bbox_transform,
utils.bbox, nets.resnet (full ResNet-101), layer_utils.generate_3d_anchors, anchor/proposal target
layers and utils.loss_utils.  ``easydict`` is absent here, so a 15-line attribute-dict stand-in is
registered before ``model.config`` is imported.  Not importable: nets.network (file missing),
proposal_layer / torchpoolers / filter_predictions (need torchvision / cv2) — no vectors for those.

Only inputs/outputs are stored (small .npz files); weights are regenerated from seeds by
oracle.frcnn_oracle.seeded_state_dict on both sides.
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF_LIB = "/root/reference/lib"


def _install_easydict_stand_in():
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                setattr(self, k, v)

        def __setattr__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setattr__(k, v)
            super().__setitem__(k, v)

        __setitem__ = __setattr__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    mod = types.ModuleType("easydict")
    mod.EasyDict = EasyDict
    sys.modules["easydict"] = mod


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    _install_easydict_stand_in()
    sys.path.insert(0, REF_LIB)
    sys.path.insert(0, ROOT)
    torch.set_num_threads(8)
    from layer_utils.generate_anchors import generate_anchors            # reference
    from layer_utils.snippets import generate_anchors_pre                # reference
    from model.bbox_transform import bbox_transform, bbox_transform_inv, clip_boxes  # reference
    from utils.bbox import bbox_overlaps                                 # reference
    from model.config import cfg as ref_cfg                              # reference
    import nets.resnet as ref_resnet                                     # reference
    from oracle.frcnn_oracle import seeded_state_dict                    # weights only

    # ---- anchors ---------------------------------------------------------------------------------
    scales, ratios = (2, 4, 8, 16, 32), (0.5, 0.75, 1, 1.25, 2)
    out = {"default9": generate_anchors(), "waymo25": generate_anchors(ratios=np.array(ratios), scales=np.array(scales))}
    a_img, n_img = generate_anchors_pre(38, 63, 16, scales, ratios)
    out["pre_38x63_s16"] = a_img
    a_half, _ = generate_anchors_pre(19, 32, 16, scales, ratios, 0.5)
    out["pre_19x32_s16_fs0.5"] = a_half
    a_odd, _ = generate_anchors_pre(5, 7, 16, scales, ratios, 0.3)       # non-dyadic frame scale
    out["pre_5x7_s16_fs0.3"] = a_odd
    a_fpn, _ = generate_anchors_pre(150, 250, 4, scales, ratios)
    out["pre_150x250_s4_sha256"] = np.frombuffer(sha(a_fpn).encode(), dtype=np.uint8)
    out["pre_150x250_s4_probe"] = a_fpn[::9973]
    np.savez_compressed(os.path.join(HERE, "anchors.npz"), **out)

    # ---- box codec -------------------------------------------------------------------------------
    g = torch.Generator().manual_seed(1234)
    n = 512
    xy = torch.rand(n, 2, generator=g) * 900
    wh = torch.rand(n, 2, generator=g) * 300 + 1
    boxes = torch.cat((xy, xy + wh), 1)
    boxes[:8] = torch.tensor([[0., 0, 15, 15], [10, 20, 49, 39], [5, 5, 5, 5], [0, 0, 999, 599],
                              [100, 100, 99, 99], [3.5, 7.25, 80.125, 90.5], [0, 0, 0, 0], [998, 598, 999, 599]])
    d1 = torch.randn(n, 4, generator=g) * 0.3
    d1[0] = torch.tensor([0.1, -0.2, 0.3, 0.0])
    d1[1] = 0.0
    d2 = torch.randn(n, 8, generator=g) * 0.5
    info = np.array([0, 1000, 0, 600, 0, 0, 1.0], dtype=np.float32)
    info2 = np.array([5, 700, 10, 400, 0, 0, 0.5], dtype=np.float32)
    gxy = torch.rand(n, 2, generator=g) * 900
    gwh = torch.rand(n, 2, generator=g) * 300 + 1
    gt = torch.cat((gxy, gxy + gwh), 1)
    codec = {
        "boxes": boxes.numpy(), "deltas1": d1.numpy(), "deltas2": d2.numpy(), "gt": gt.numpy(),
        "info": info, "info2": info2,
        "inv1": bbox_transform_inv(boxes, d1).numpy(),
        "inv2": bbox_transform_inv(boxes, d2).numpy(),
        "inv2_scale0.5": bbox_transform_inv(boxes.clone(), d2, 0.5).numpy(),
        "clip1": clip_boxes(bbox_transform_inv(boxes, d1), info).numpy(),
        "clip2_info2": clip_boxes(bbox_transform_inv(boxes, d2), info2).numpy(),
        "fwd": bbox_transform(boxes, gt).numpy(),
        "overlaps": np.asarray(bbox_overlaps(boxes[:64], gt[:48])),
    }
    np.savez_compressed(os.path.join(HERE, "box_codec.npz"), **codec)

    # ---- ResNet-101 stage outputs (reference nets/resnet.py, weights from names+seed) -------------
    ref_cfg.USE_FPN = False
    net = ref_resnet.resnet101()
    net.eval()
    res = {}
    for bn_mode, seed in (("random", 11), ("tame", 12)):
        sd = seeded_state_dict(net, seed, bn_mode=bn_mode, all_backbone=True)
        net.load_state_dict(sd, strict=True)
        gi = torch.Generator().manual_seed(77)
        x = torch.randn(1, 3, 64, 96, generator=gi) * 50
        with torch.no_grad():
            stem = net.maxpool(net.relu(net.bn1(net.conv1(x))))
            l1 = net.layer1(stem)
            l2 = net.layer2(l1)
            l3 = net.layer3(l2)
            pooled = torch.randn(3, 1024, 7, 7, generator=gi) * l3.abs().mean()
            l4 = net.layer4(pooled)
        tag = bn_mode + "_"
        res[tag + "x"] = x.numpy()
        res[tag + "pooled"] = pooled.numpy()
        res[tag + "stem"] = stem.numpy()
        res[tag + "layer1"] = l1.numpy()
        res[tag + "layer2"] = l2.numpy()
        res[tag + "layer3"] = l3.numpy()
        res[tag + "layer4_probe"] = l4.numpy()[:, ::16]
        res[tag + "layer4_mean"] = l4.mean(3).mean(2).numpy()
    np.savez_compressed(os.path.join(HERE, "resnet101_stages.npz"), **res)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
