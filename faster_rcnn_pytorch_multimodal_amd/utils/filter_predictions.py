"""Test-time post-processing on the device — counterpart of lib/utils/filter_predictions.py:45-130.

The reference loops over classes in Python and copies to the host once per class (``.cpu().numpy()`` at
:64,:67).  Here one launch (``frcnn_filter_per_class``: one workgroup per foreground class) clamps the
boxes to the frame, thresholds, sorts by (score desc, roi index asc), runs NMS at cfg.TEST.NMS_THRESH and
optionally applies the ``max_dets`` cut of lib/model/test.py:213-221; a single device->host copy follows.
"""
import numpy as np
import torch

from .. import ops
from ..model.config import cfg


def filter_device(rois_count, cls_prob, pred_boxes, info, thresh=0.1, max_dets=0, max_out=None, db_type='image',
                  uncertainties=None):
    """Asynchronous form: returns (dets (K, max_out, E+1+U), det_count (K,)) device tensors, E = 4 image /
    7 LiDAR.  Image ``pred_boxes`` are clamped IN PLACE like the reference (:85-91); LiDAR boxes are not clamped
    and are suppressed on their yaw-less BEV rectangle (:55-62,67).  With a non-empty ``uncertainties`` dict
    (cfg.UC.*) every detection row is followed by its U uncertainty columns in the order of
    nets/uncertainty.UNCERTAINTY_ORDER (nms_hstack_var_torch :23-43 + stack_uncertainties, lib/model/test.py:260-270)."""
    want = bool(uncertainties)
    if db_type == 'lidar':
        out = ops.filter_per_class_lidar(pred_boxes, cls_prob, thresh, cfg.TEST.NMS_THRESH, max_dets, max_out,
                                         roi_count=rois_count, want_rois=want)
    else:
        info = np.asarray(info, dtype=np.float32)
        frame_w, frame_h, scale = info[1] - info[0], info[3] - info[2], info[6]
        out = ops.filter_per_class(pred_boxes, cls_prob, float(frame_w), float(frame_h), float(scale), thresh,
                                   cfg.TEST.NMS_THRESH, max_dets, max_out, roi_count=rois_count, want_rois=want)
    if not want:
        return out
    from ..nets.uncertainty import stack_uncertainty_columns
    dets, det_count, det_roi = out
    cols = stack_uncertainty_columns(uncertainties, det_roi, cls_prob.shape[1])
    return torch.cat((dets, cols), 2).contiguous(), det_count


def filter_and_draw_prep(rois, cls_score, pred_boxes, uncertainties, info, num_classes, thresh=0.1, db_type='none'):
    """Reference signature and return value: (rois_np (R,4), all_boxes[K] of (n_j,5) (LiDAR (n_j,8)) float32 arrays,
    all_uncertainty[K] dicts).  ``cls_score`` is the class-probability tensor (lib/model/test.py:75-93)."""
    if db_type not in ('image', 'lidar'):
        return None
    dets, det_count = filter_device(None, cls_score.contiguous(), pred_boxes, info, thresh, db_type=db_type,
                                    uncertainties=uncertainties)
    dets_np = dets.cpu().numpy()          # the one device->host copy (+ implicit sync)
    counts = det_count.cpu().numpy()
    nbox = (7 if db_type == 'lidar' else 4) + 1
    all_boxes = [[] for _ in range(num_classes)]
    all_uncertainty = [{} for _ in range(num_classes)]
    from ..nets.uncertainty import UNCERTAINTY_ORDER
    for j in range(1, num_classes):
        rows = dets_np[j, :counts[j]]
        all_boxes[j] = rows[:, :nbox].copy() if counts[j] > 0 else np.empty(0)
        # per-class dict of (n_j, width) arrays, keys and widths as filter_predictions.py:113-124 hands them on
        col = nbox
        for key in UNCERTAINTY_ORDER:
            if not uncertainties or key not in uncertainties:
                continue
            v = uncertainties[key]
            width = 1 if v.dim() == 1 else (v.shape[1] // num_classes if key.endswith('bbox_var') else v.shape[1])
            all_uncertainty[j][key] = rows[:, col:col + width].copy()
            col += width
    return rois[:, 1:5].detach().cpu().numpy(), all_boxes, all_uncertainty
