"""RPN proposals on the device — counterpart of lib/layer_utils/proposal_layer.py:18-57.

decode + clip (one kernel) -> top-k sort with the canonical order (score desc, index asc; the reference's
sort is unstable, :39) -> NMS bit-matrix + single-wave greedy scan -> first post_nms_topN survivors.
No min-size filter (the reference has none).  Sizes that depend on the data stay on the device
(``ProposalResult.count``); only the reference-shaped wrapper ``proposal_layer`` synchronises to slice.
"""
import collections


from .. import ops
from ..model.config import cfg

ProposalResult = collections.namedtuple(
    'ProposalResult', 'rois roi_scores count order keep_idx sorted_boxes sorted_scores scores proposals sorted_count')


def proposal_layer_device(anchors, info, num_anchors, pre_nms_top_n, post_nms_top_n, nms_thresh, rpn=None,
                          rpn_cls_prob_fg=None, rpn_bbox_pred=None):
    """Asynchronous, fixed-shape form.  Either ``rpn`` (H*W, >=6A: [bg logits | fg logits | deltas]) or
    (``rpn_cls_prob_fg`` (H*W*A,), ``rpn_bbox_pred`` (H*W*A,4)).  rois is (post_nms_top_n, 5), rows past
    ``count`` are zero."""
    scores, proposals = ops.rpn_decode_clip(anchors, info, num_anchors, rpn=rpn, probs=rpn_cls_prob_fg,
                                            deltas=rpn_bbox_pred)
    top_n = pre_nms_top_n if pre_nms_top_n > 0 else scores.numel()
    order, sorted_scores, n_sorted = ops.sort_topk_desc(scores, top_n)
    sorted_boxes = ops.gather_rows(proposals, order, n_sorted)
    max_keep = post_nms_top_n if post_nms_top_n > 0 else order.numel()
    keep_idx, keep_count, _ = ops.nms_sorted(sorted_boxes, nms_thresh, max_keep=max_keep, n_dev=n_sorted)
    rois, roi_scores = ops.make_rois(sorted_boxes, sorted_scores, keep_idx, keep_count)
    return ProposalResult(rois, roi_scores, keep_count, order, keep_idx, sorted_boxes, sorted_scores, scores, proposals,
                          n_sorted)


def proposal_layer(rpn_cls_prob, rpn_bbox_pred, info, cfg_key, anchors, anchors_3d, num_anchors):
    """Reference signature.  rpn_cls_prob (1,H,W,2A) with the fg half last (:32), rpn_bbox_pred (1,H,W,4A).
    Returns (blob (n,5), scores (n,1), anchors_3d[n]) with n = number of survivors."""
    if isinstance(cfg_key, bytes):
        cfg_key = cfg_key.decode('utf-8')
    fg = rpn_cls_prob[:, :, :, num_anchors:].contiguous().view(-1)
    deltas = rpn_bbox_pred.contiguous().view(-1, 4)
    res = proposal_layer_device(anchors, info, num_anchors, cfg[cfg_key].RPN_PRE_NMS_TOP_N,
                                cfg[cfg_key].RPN_POST_NMS_TOP_N, cfg[cfg_key].RPN_NMS_THRESH,
                                rpn_cls_prob_fg=fg, rpn_bbox_pred=deltas)
    n = int(res.count.item())  # the one host sync of the reference-shaped call
    blob = res.rois[:n]
    scores = res.roi_scores[:n]
    if anchors_3d is not None:
        a3_sorted = ops.gather_rows(anchors_3d.contiguous(), res.order, res.sorted_count)   # :44
        anchors_3d = ops.gather_rows(a3_sorted, res.keep_idx, res.count)[:n]                # :52
    return blob, scores, anchors_3d
