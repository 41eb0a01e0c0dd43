"""Second-stage RoI sampling and regression targets on the device — counterpart of
lib/layer_utils/proposal_target_layer.py:22-262 (image detector and the 7-DoF LiDAR form).

``frcnn_proposal_target_layer`` (one workgroup): IoU (+1 convention) of every proposal with every gt box,
foreground (>= FG_THRESH) / background ([BG_THRESH_LO, BG_THRESH_HI)) candidates, random sampling to
ROI_BATCH_SIZE rows with at most FG_FRACTION foreground (with replacement when a pool is too small, :207-231),
bbox_transform targets normalised by cfg.TRAIN.IMAGE.BBOX_NORMALIZE_{MEANS,STDS} and expanded to the 4-of-4K
class layout (:64-103).  Foreground rows come first, like the reference's ``torch.cat([fg_inds, bg_inds])``.
LiDAR (NET_TYPE 'lidar', :142-154,239-243): overlaps on the BEV rectangles, ``lidar_3d_bbox_transform`` targets against
the 8-column ``true_gt_boxes`` and each RoI's 3-D anchor, normalised by cfg.TRAIN.LIDAR.BBOX_NORMALIZE_*, 7-of-7K
layout; the sampled rows' 3-D anchors are returned as well.
"""
import torch

from .. import ops
from ..model.config import cfg
from .anchor_target_layer import _draw_seed


def proposal_target_layer_device(rpn_rois, rpn_scores, gt_boxes, num_classes, roi_count=None, seed=None,
                                 anchors_3d=None, true_gt_boxes=None, gt_boxes_dc=None, seed_dev=None, gt_count=None):
    """``gt_count``: one-element int32 device tensor = live rows of ``gt_boxes`` / ``true_gt_boxes`` buffers padded to a fixed
    capacity (captured training steps, model/train_graph.py)."""
    scores = None if rpn_scores is None else rpn_scores.contiguous().view(-1)
    skip = None
    if cfg.TRAIN.USE_GT:
        # proposal_target_layer.py:31-37: the gt boxes join the candidates (score 0; LiDAR: the 3-D gt box is its own
        # "anchor").  They go in FRONT of the proposals so that the device-side count of live rows still describes a
        # prefix; the sampling is random, so the position of a candidate carries no meaning.
        g = gt_boxes.shape[0]
        gt_rois = torch.cat((gt_boxes.new_zeros(g, 1), gt_boxes[:, :4]), 1)
        rpn_rois = torch.cat((gt_rois, rpn_rois), 0)
        if scores is not None:
            scores = torch.cat((scores.new_zeros(g), scores), 0)
        if roi_count is not None:
            roi_count = (roi_count.to(torch.int32) + g).to(torch.int32)
        if anchors_3d is not None:
            anchors_3d = torch.cat((true_gt_boxes[:, :7], anchors_3d), 0)
        if gt_count is not None:
            # padded gt buffer: its rows past the count are no candidates (and the live prefix of the proposals starts
            # behind ALL g rows, which the count above already assumes)
            row = torch.arange(rpn_rois.shape[0], device=rpn_rois.device, dtype=torch.int32)
            skip = ((row >= gt_count) & (row < g)).to(torch.uint8)
    if cfg.TRAIN.IGNORE_DC and gt_boxes_dc is not None and len(gt_boxes_dc) > 0:
        # proposal_target_layer.py:180-187: proposals whose best overlap with a don't-care box reaches DC_THRESH are
        # dropped before sampling (the mask is applied inside the sampling kernel: no compaction, no host sync)
        dc = gt_boxes_dc[:, :4].contiguous()
        dc_skip = (ops.bbox_overlaps(rpn_rois[:, 1:5].contiguous(), dc).max(1)[0] >= cfg.TRAIN.DC_THRESH).to(torch.uint8)
        skip = dc_skip if skip is None else (skip | dc_skip)
    if anchors_3d is not None:
        return ops.proposal_target_layer(rpn_rois.contiguous(), scores, gt_boxes[:, :5].contiguous(), num_classes,
                                         cfg.TRAIN.ROI_BATCH_SIZE, cfg.TRAIN.FG_FRACTION, cfg.TRAIN.FG_THRESH,
                                         cfg.TRAIN.BG_THRESH_HI, cfg.TRAIN.BG_THRESH_LO,
                                         cfg.TRAIN.LIDAR.BBOX_NORMALIZE_MEANS, cfg.TRAIN.LIDAR.BBOX_NORMALIZE_STDS,
                                         _draw_seed() if seed is None else seed, roi_count=roi_count,
                                         anchors_3d=anchors_3d.contiguous(), true_gt_boxes=true_gt_boxes.contiguous(),
                                         skip_mask=skip, seed_dev=seed_dev, gt_count=gt_count)
    return ops.proposal_target_layer(rpn_rois.contiguous(), scores, gt_boxes[:, :5].contiguous(), num_classes,
                                     cfg.TRAIN.ROI_BATCH_SIZE, cfg.TRAIN.FG_FRACTION, cfg.TRAIN.FG_THRESH,
                                     cfg.TRAIN.BG_THRESH_HI, cfg.TRAIN.BG_THRESH_LO,
                                     cfg.TRAIN.IMAGE.BBOX_NORMALIZE_MEANS, cfg.TRAIN.IMAGE.BBOX_NORMALIZE_STDS,
                                     _draw_seed() if seed is None else seed, roi_count=roi_count, skip_mask=skip,
                                     seed_dev=seed_dev, gt_count=gt_count)


def proposal_target_layer(rpn_rois, rpn_scores, anchors_3d, gt_boxes, true_gt_boxes, gt_boxes_dc, _num_classes,
                          num_bbox_elem):
    """Reference signature (:22) and return tuple: labels (R,1), rois (R,5), anchors_3d, roi_scores (R,),
    bbox_targets, bbox_inside_weights, bbox_outside_weights."""
    if num_bbox_elem == 7:
        out = proposal_target_layer_device(rpn_rois, rpn_scores, gt_boxes, _num_classes, anchors_3d=anchors_3d,
                                           true_gt_boxes=true_gt_boxes)
        return (out["labels"].view(-1, 1), out["rois"], out["anchors_3d"], out["scores"], out["targets"], out["inside"],
                out["outside"])
    out = proposal_target_layer_device(rpn_rois, rpn_scores, gt_boxes, _num_classes)
    return (out["labels"].view(-1, 1), out["rois"], anchors_3d, out["scores"], out["targets"], out["inside"],
            out["outside"])
