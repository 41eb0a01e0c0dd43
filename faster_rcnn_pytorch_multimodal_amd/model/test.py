"""Per-frame detection loop body — counterpart of lib/model/test.py:68-116 (frame_detect) and :210-228
(the ``max_dets`` cut that test_net applies per class), for the image detector.

``frame_detect`` keeps the reference's call sequence (test_frame -> filter_and_draw_prep).
``detect_frames`` is the throughput form used by bench.py and the 8-GPU eval collate: everything up to
the per-class, max_dets-limited detections stays on the device and a frame ends with ONE device->host
copy of a fixed-size record.
"""
import numpy as np
import torch

from ..model.config import cfg
from ..utils.filter_predictions import filter_and_draw_prep, filter_device


def frame_detect(net, blobs, num_classes, thresh):
    _, probs, bbox_pred, rois, uncertainties = net.test_frame(blobs['data'], blobs['info'])
    return filter_and_draw_prep(rois, probs, bbox_pred, uncertainties, blobs['info'], num_classes, thresh,
                                cfg.NET_TYPE)


def apply_max_dets(cls_boxes, max_dets):
    """test.py:213-221: keep every detection scoring >= the max_dets-th best (ties stay)."""
    if max_dets > 0 and len(cls_boxes) > max_dets:
        cut = np.sort(cls_boxes[:, -1])[-max_dets]
        cls_boxes = cls_boxes[np.where(cls_boxes[:, -1] >= cut)[0], :]
    return cls_boxes


def detect_frame_device(net, data, info, thresh=0.5, max_dets=100, max_out=None):
    """One frame, asynchronous: returns (dets (K, max_out, 5), det_count (K,)) device tensors holding,
    per class, the detections test_net would store in all_boxes[cls][frame]."""
    with torch.no_grad():
        net.forward(data, info, None, None, mode='TEST')
    p = net._predictions
    max_out = max_out if max_out is not None else p['cls_prob'].shape[0]
    return filter_device(p['rois_count'], p['cls_prob'], p['pred_boxes'], info, thresh, max_dets, max_out,
                         db_type=cfg.NET_TYPE)


def lidar_extents():
    """[x1,y1,z1,x2,y2,z2] of the LiDAR scan in metres (lib/datasets/db.py passes cfg.LIDAR.*_RANGE this way)."""
    return [cfg.LIDAR.X_RANGE[0], cfg.LIDAR.Y_RANGE[0], cfg.LIDAR.Z_RANGE[0],
            cfg.LIDAR.X_RANGE[1], cfg.LIDAR.Y_RANGE[1], cfg.LIDAR.Z_RANGE[1]]


def bbox_voxel_grid_to_pc(bboxes, bev_extents, info):
    """lib/utils/bbox.py:140-162 (called at lib/model/test.py:223-224 on the host copy of the detections):
    voxel-grid [xc,yc,zc,l,w,h,ry,...] rows -> metres, in place on the numpy array."""
    scale = info[6]
    s_info = np.asarray(info[0:6]) * 1 / scale
    kx = (bev_extents[3] - bev_extents[0]) / (s_info[1] - s_info[0])
    ky = (bev_extents[4] - bev_extents[1]) / (s_info[3] - s_info[2])
    bboxes[:, 0] = bboxes[:, 0] * kx + bev_extents[0]
    bboxes[:, 1] = bboxes[:, 1] * ky + bev_extents[1]
    bboxes[:, 3] = bboxes[:, 3] * kx
    bboxes[:, 4] = bboxes[:, 4] * ky
    return bboxes
