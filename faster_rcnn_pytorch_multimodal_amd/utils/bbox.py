"""Host-side box helpers of the reference's lib/utils/bbox.py that the LiDAR training path needs.

``bbaa_graphics_gems``: axis-aligned BEV rectangle enclosing a yawed 3-D box (Arvo's "Transforming Axis-Aligned
Bounding Boxes", lib/utils/bbox.py:256-293).  It runs on the HOST on the handful of ground-truth boxes of a frame,
exactly where the reference runs it (numpy, inside the data path); the dense 3-D anchor grid gets the same rectangles
from the device (``frcnn_generate_anchors_3d``).
"""
import numpy as np


def bbaa_graphics_gems(bboxes, width=0, height=0, clip=False):
    """(N,>=7) [xc,yc,zc,l,w,h,ry] -> (N,4) [x1,y1,x2,y2].  The rotation matrix keeps the dtype of the boxes, the half
    extents are float64, the per-axis min/max sums are cast to float32 before the centre is added (bbox.py:259-283)."""
    b = np.asarray(bboxes)
    if b.shape[0] == 0:
        return np.zeros((0, 4), dtype=np.float32)
    cos, sin = np.cos(b[:, 6]), np.sin(b[:, 6])
    rot = np.stack((np.stack((cos, sin), 1), np.stack((-sin, cos), 1)), 1)           # (N,2,2) [[c,s],[-s,c]]
    half = np.stack((b[:, 3] / 2.0, b[:, 4] / 2.0), 1).astype(np.float64)
    lo_terms, hi_terms = rot * (-half)[:, None, :], rot * half[:, None, :]
    bmin = np.minimum(lo_terms, hi_terms).sum(2).astype(np.float32) + b[:, 0:2]
    bmax = np.maximum(lo_terms, hi_terms).sum(2).astype(np.float32) + b[:, 0:2]
    out = np.concatenate((bmin[:, 0:1], bmin[:, 1:2], bmax[:, 0:1], bmax[:, 1:2]), 1)
    if clip:                                                                         # _bbox_clip(width-1, height-1)
        out[:, 0::2] = np.clip(out[:, 0::2], 0, width - 1)
        out[:, 1::2] = np.clip(out[:, 1::2], 0, height - 1)
    return out


def bbox_overlaps(boxes, query_boxes):
    """(N,4) x (K,4) device tensors -> (N,K) IoU with the +1 area convention.  lib/utils/bbox.py:5-33."""
    from .. import ops
    return ops.bbox_overlaps(boxes.contiguous(), query_boxes.contiguous())


def bbox_pc_to_voxel_grid(bboxes, bev_extents, info):
    """lib/utils/bbox.py:113-125: metric [xc,yc,zc,l,w,h,ry] rows -> voxel-grid units of the UNSCALED frame, in place."""
    scale = info[6]
    s_info = np.asarray(info[0:6]) * 1 / scale
    kx = (s_info[1] - s_info[0]) / (bev_extents[3] - bev_extents[0])
    ky = (s_info[3] - s_info[2]) / (bev_extents[4] - bev_extents[1])
    bboxes[:, 0] = (bboxes[:, 0] - bev_extents[0]) * kx
    bboxes[:, 1] = (bboxes[:, 1] - bev_extents[1]) * ky
    bboxes[:, 3] = bboxes[:, 3] * kx
    bboxes[:, 4] = bboxes[:, 4] * ky
    return bboxes


def bbox_voxel_grid_to_pc(bboxes, bev_extents, info):
    """lib/utils/bbox.py:140-162, the inverse (host copy of the detections, lib/model/test.py:223-224)."""
    from ..model.test import bbox_voxel_grid_to_pc as impl
    return impl(bboxes, bev_extents, info)
