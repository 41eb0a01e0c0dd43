"""3-D anchor grid for the LiDAR-BEV detector — counterpart of lib/layer_utils/generate_3d_anchors.py:15-118,
together with the axis-aligned BEV rectangle of every anchor (lib/utils/bbox.py:256-293 with clip=False),
which is what the RPN regresses against.

Only the per-type table (one row per (size, rotation) pair: BEV half extents, z, l, w, h, ry) is computed on
the host; ``frcnn_generate_anchors_3d`` tiles it over the (H, W) grid on the device.  Results are cached
per (H, W, stride, scale, rotations, frame_scale, device).
"""
import numpy as np
import torch

from .. import ops
from ..model.config import cfg

_CACHE = {}


def anchor_type_table(anchor_scales, anchor_rotations, frame_scale):
    """(T, 9) float32 rows [lo_x, lo_y, hi_x, hi_y, z, l, w, h, ry], T = sizes x rotations, rotation fastest."""
    assert len(anchor_scales) == 1                                         # generate_3d_anchors.py:31
    voxel_len = cfg.LIDAR.VOXEL_LEN / frame_scale                          # :37 (inversely prop. to frame scale)
    sizes = (np.asarray(cfg.LIDAR.ANCHORS) / np.array([voxel_len, voxel_len, 1.0]) * anchor_scales[0]).astype(np.float32)
    rots = np.asarray(anchor_rotations, dtype=np.float64).astype(np.float32)
    z = np.float32(np.asarray(cfg.LIDAR.ANCHORS)[0][2] / np.array([voxel_len, voxel_len, 1.0])[2]
                   * anchor_scales[0] / 2.0)                               # :100 (first size only)
    rows = []
    for l, w, h in sizes:
        for ry in rots:
            # bbox.py:258-279: M = [[cos, sin], [-sin, cos]] in the boxes' dtype (float32), half extents float64,
            # per-axis min/max of the two products summed in float64, cast to float32
            c, s = np.cos(ry), np.sin(ry)                                  # float32
            m = np.array([[c, s], [-s, c]], dtype=np.float32).astype(np.float64)
            amin = np.array([-(l / np.float32(2.0)), -(w / np.float32(2.0))], dtype=np.float64)
            lo_p, hi_p = m * amin[None, :], m * (-amin)[None, :]
            lo = np.minimum(lo_p, hi_p).sum(1).astype(np.float32)
            hi = np.maximum(lo_p, hi_p).sum(1).astype(np.float32)
            rows.append([lo[0], lo[1], hi[0], hi[1], z, l, w, h, ry])
    return np.asarray(rows, dtype=np.float32)


def generate_anchors_3d(height, width, feature_stride, anchor_scales, anchor_rotations, frame_scale, device='cuda'):
    """Returns (num_anchors, anchors_3d (N,7), anchors_2d (N,4)) as DEVICE tensors, order (H, W, size, rot)."""
    scales = tuple(float(s) for s in np.asarray(anchor_scales, dtype=np.float64).ravel())
    rots = tuple(float(r) for r in np.asarray(anchor_rotations, dtype=np.float64).ravel())
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    key = (int(height), int(width), int(feature_stride), scales, rots, float(frame_scale), str(device),
           float(cfg.LIDAR.VOXEL_LEN), tuple(np.asarray(cfg.LIDAR.ANCHORS).ravel().tolist()))
    hit = _CACHE.get(key)
    if hit is None:
        table = torch.from_numpy(anchor_type_table(scales, np.asarray(rots), frame_scale)).to(device)
        hit = ops.generate_anchors_3d(table, int(height), int(width), int(feature_stride))
        _CACHE[key] = hit
    return hit[0].shape[0], hit[0], hit[1]


class GridAnchor3dGenerator(object):
    """Name and call shape of the reference class (generate_3d_anchors.py:10-44)."""

    def name_scope(self):
        return 'GridAnchor3dGenerator'

    def _generate(self, height, width, feature_stride, anchor_scales, anchor_rotations, frame_scale, device='cuda'):
        n, a3, _ = generate_anchors_3d(height, width, feature_stride, anchor_scales, anchor_rotations, frame_scale, device)
        return n, a3
