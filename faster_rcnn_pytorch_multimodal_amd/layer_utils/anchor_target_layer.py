"""RPN training targets on the device — counterpart of lib/layer_utils/anchor_target_layer.py:22-165
(``anchor_target_layer_torch``).

One call runs ``frcnn_anchor_target_layer``: inside-frame test, IoU (+1 convention) against every gt box, labels
(bg < 0.3, per-gt best anchors and >= 0.7 -> fg), random sub-sampling to RPN_FG_FRACTION*RPN_BATCHSIZE foreground /
RPN_BATCHSIZE examples, bbox_transform targets and the uniform 1/num_examples weights.  The random draws use a
hash keyed by a seed taken from torch's CPU generator (``torch.manual_seed`` keeps runs repeatable); they are not
the reference's torch.randperm stream, which depends on the device anyway.
"""
import torch

from .. import ops
from ..model.config import cfg


def _draw_seed():
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item())


def anchor_target_layer_device(gt_boxes, info, all_anchors, seed=None, seed_dev=None, gt_count=None):
    """Flat form used by the training forward: labels (N,), targets/inside/outside (N,4) in anchor order
    ((H,W,A), A fastest) and counts (2,) int32 = fg / bg candidates before sub-sampling.
    ``seed_dev``: one-element int32 device tensor added to ``seed`` on the device (model/train_graph.py: a replayed
    hipGraph keeps the launch argument, the per-step seed comes through that tensor); ``gt_count``: one-element int32 device
    tensor, the live rows of a ``gt_boxes`` buffer padded to a fixed capacity (same reason)."""
    # cfg.TRAIN.IGNORE_DC needs nothing here: the reference's branch (anchor_target_layer.py:58-64) writes -1 into labels
    # that are still all -1 and every later rule overwrites them, i.e. it has no effect on the output
    if cfg.TRAIN.RPN_CLOBBER_POSITIVES or cfg.TRAIN.RPN_POSITIVE_WEIGHT >= 0:
        raise NotImplementedError("RPN_CLOBBER_POSITIVES / RPN_POSITIVE_WEIGHT >= 0 are not on the HIP path")
    if tuple(cfg.TRAIN.RPN_BBOX_INSIDE_WEIGHTS) != (1.0, 1.0, 1.0, 1.0):
        raise NotImplementedError("RPN_BBOX_INSIDE_WEIGHTS other than (1,1,1,1)")
    return ops.anchor_target_layer(all_anchors.contiguous(), gt_boxes[:, :5].contiguous(), info, cfg.TRAIN.RPN_BATCHSIZE,
                                   cfg.TRAIN.RPN_FG_FRACTION, cfg.TRAIN.RPN_NEGATIVE_OVERLAP,
                                   cfg.TRAIN.RPN_POSITIVE_OVERLAP, _draw_seed() if seed is None else seed, seed_dev=seed_dev,
                                   gt_count=gt_count)


def anchor_target_layer_torch(gt_boxes, gt_boxes_dc, info, all_anchors, num_anchors, height, width, dev=None):
    """Reference signature and output shapes (anchor_target_layer.py:137-165): rpn_labels (1,A,H,W),
    rpn_bbox_targets / inside / outside weights (1,H,W,4A)."""
    labels, targets, inside, outside, _ = anchor_target_layer_device(gt_boxes, info, all_anchors)
    a = num_anchors
    return (labels.view(1, height, width, a).permute(0, 3, 1, 2), targets.view(1, height, width, a * 4),
            inside.view(1, height, width, a * 4), outside.view(1, height, width, a * 4))
