"""Box codec on device tensors — counterpart of lib/model/bbox_transform.py (image detector part).

Quirks kept from the reference: centre deltas are scaled by the box DIAGONAL sqrt(w^2+h^2) (:55,:84),
widths use the +1 convention, decoded x2 = cx + 0.5*w' (no -1) (:99-103); clip uses ``info`` positions
[x_min, x_max, y_min, y_max] (:252-255).  The arithmetic runs in libfrcnn_hip.so.
"""
from .. import ops


def bbox_transform_inv(boxes, deltas, scales=None):
    """boxes (N,4+) , deltas (N,4K) -> (N,4K).  lib/model/bbox_transform.py:75-105 (no boxes: ``deltas * 0``, :79-80)."""
    if len(boxes) == 0:
        return deltas.detach() * 0
    return ops.bbox_transform_inv(boxes.contiguous(), deltas.contiguous(), scales)


def clip_boxes(boxes, shape):
    """Clamp x to [shape[0], shape[1]-1] and y to [shape[2], shape[3]-1].  bbox_transform.py:235-257."""
    return ops.clip_boxes(boxes.contiguous(), [float(v) for v in list(shape)[:4]])


def lidar_3d_bbox_transform_inv(rois, boxes, deltas, scales=None):
    """rois (N,4) axis-aligned BEV RoIs, boxes (N,7) their 3-D anchors, deltas (N,7K) -> (N,7K)
    [xc,yc,zc,l,w,h,ry].  lib/model/bbox_transform.py:174-233 (the reference also divides the anchors' x,y,l,w
    by ``scales`` in place, :177-178, but never reads them afterwards; that side effect is not reproduced)."""
    if len(boxes) == 0:                      # :181-182
        return deltas.detach() * 0
    return ops.lidar_bbox_transform_inv(rois.contiguous(), boxes.contiguous(), deltas.contiguous(), scales)


def bbox_transform(ex_rois, gt_rois):
    """Targets (N,4) [dx,dy,dw,dh] of gt_rois against ex_rois, row by row.  lib/model/bbox_transform.py:52-70."""
    return ops.bbox_transform(ex_rois.contiguous(), gt_rois.contiguous())


def lidar_3d_bbox_transform(ex_rois, ex_anchors, gt_rois):
    """Targets (N,7) of the 3-D gt rows against the BEV RoIs and their 3-D anchors.  lib/model/bbox_transform.py:16-49."""
    return ops.lidar_bbox_transform(ex_rois.contiguous(), ex_anchors.contiguous(), gt_rois.contiguous())


def uncertainty_transform_inv(boxes, deltas, uncertainty, scales=None):
    """lib/model/bbox_transform.py:107-130: uncertainty (N, 7K) of the deltas -> (N, 4K) squared BEV terms [x,y,l,w]; ``boxes``
    are the (N,4) RoIs the deltas were regressed against (``deltas`` is unused upstream too).  Per-box scaling: upstream's
    missing ``unsqueeze`` only gives that for one box per call (tests/golden/make_golden_uc_inv.py)."""
    return ops.uncertainty_transform_inv(boxes.contiguous(), uncertainty.contiguous(), scale=scales, lidar=False)


def lidar_3d_uncertainty_transform_inv(rois, boxes, deltas, uncertainty, scales=None):
    """lib/model/bbox_transform.py:132-169: rois (N,4) BEV RoIs, boxes (N,7) their 3-D anchors, uncertainty (N,7K) ->
    (N,7K) squared terms (the in-place division of the anchors by ``scales`` upstream does not change the result: only
    their height is read)."""
    return ops.uncertainty_transform_inv(rois.contiguous(), uncertainty.contiguous(), boxes.contiguous(), scales, lidar=True)
