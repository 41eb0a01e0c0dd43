"""``TEST.MODE == 'top'`` proposals — counterpart of lib/layer_utils/proposal_top_layer.py:18-59: the
cfg.TEST.RPN_TOP_N best-scoring anchors, decoded and clipped, WITHOUT NMS.

Decode + clip is elementwise, so it runs over all anchors first (``frcnn_rpn_decode_clip``) and the top-N rows are
gathered afterwards (``frcnn_sort_topk_desc`` in the canonical order score desc / index asc, ``frcnn_gather_rows``):
the selected values are identical to selecting first and decoding after.  With fewer anchors than RPN_TOP_N the
reference samples indices WITH replacement from numpy's global RNG (:32-37); that draw happens on the host here too.
"""
import numpy.random as npr
import torch

from .. import ops
from ..model.config import cfg


def proposal_top_layer(rpn_cls_prob, rpn_bbox_pred, info, anchors, num_anchors):
    """rpn_cls_prob (1,H,W,2A) with the fg half last, rpn_bbox_pred (1,H,W,4A), anchors (H*W*A,4) device tensors.
    Returns (blob (N,5), scores (N,1), anchors (N,4)) with N = cfg.TEST.RPN_TOP_N."""
    rpn_top_n = int(cfg.TEST.RPN_TOP_N)
    fg = rpn_cls_prob[:, :, :, num_anchors:].contiguous().view(-1)
    deltas = rpn_bbox_pred.contiguous().view(-1, 4)
    scores, proposals = ops.rpn_decode_clip(anchors, info, num_anchors, probs=fg, deltas=deltas)
    length = scores.numel()
    if length < rpn_top_n:
        top = torch.from_numpy(npr.choice(length, size=rpn_top_n, replace=True)).long().to(scores.device)
        count = None
    else:
        top, _, count = ops.sort_topk_desc(scores, rpn_top_n)
    boxes = ops.gather_rows(proposals, top, count)
    sel_scores = ops.gather_rows(scores.view(-1, 1), top, count)
    sel_anchors = ops.gather_rows(anchors.contiguous(), top, count)
    blob = torch.cat((boxes.new_zeros(boxes.shape[0], 1), boxes), 1)
    return blob, sel_scores, sel_anchors
