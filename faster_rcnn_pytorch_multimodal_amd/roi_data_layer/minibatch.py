"""LiDAR input producer - counterpart of ``_get_lidar_blob`` (lib/roi_data_layer/minibatch.py:237-516) for ONE frame
whose points are already in memory: range filter (:232-235), voxelisation and the BEV scatter (:434-512) run as
``frcnn_bev_voxelize`` on the device; file parsing, FOV calibration and the augmentations (:250-431) stay with the
caller (dataset plumbing, out of scope).  Returns the same ``(infos, blob)`` the reference's data layer hands to
``Network.forward``: blob (1, num_y_voxel, num_x_voxel, cfg.LIDAR.NUM_CHANNEL) NHWC on the device,
info = [0, num_x_voxel, 0, num_y_voxel, 0, NUM_SLICES, scale].
"""
import numpy as np
import torch

from .. import ops
from ..model.config import cfg


def lidar_frame_geometry(scale):
    """Voxel size, shifted point-cloud range and info vector of a frame (minibatch.py:434-451)."""
    voxel_len = cfg.LIDAR.VOXEL_LEN / scale
    num_x_voxel = int((cfg.LIDAR.X_RANGE[1] - cfg.LIDAR.X_RANGE[0]) * (1 / voxel_len))
    num_y_voxel = int((cfg.LIDAR.Y_RANGE[1] - cfg.LIDAR.Y_RANGE[0]) * (1 / voxel_len))
    vertical = (cfg.LIDAR.Z_RANGE[1] - cfg.LIDAR.Z_RANGE[0]) / (cfg.LIDAR.NUM_SLICES + 0.0)
    assert vertical == cfg.LIDAR.VOXEL_HEIGHT
    # pc_extents shifted so that the grid starts at z = 0 (:441-443)
    pc_range = [cfg.LIDAR.X_RANGE[0], cfg.LIDAR.Y_RANGE[0], 0.0, cfg.LIDAR.X_RANGE[1], cfg.LIDAR.Y_RANGE[1],
                cfg.LIDAR.Z_RANGE[1] - cfg.LIDAR.Z_RANGE[0]]
    info = np.array([0, num_x_voxel, 0, num_y_voxel, 0, int(cfg.LIDAR.NUM_SLICES), scale], dtype=np.float32)
    return [voxel_len, voxel_len, cfg.LIDAR.VOXEL_HEIGHT], pc_range, info


def get_lidar_blob(points, scale, device='cuda', elongation=None):
    """points: (N, >=4) float32 rows [x, y, z, intensity, (elongation)] in file order (numpy or device tensor).
    ``elongation``: column index of the elongation value (Waymo, minibatch.py:496-499) or None (channel = tanh(0))."""
    if isinstance(points, np.ndarray):
        points = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32)).to(device, non_blocking=True)
    voxel_size, pc_range, info = lidar_frame_geometry(scale)
    num_meta = int(cfg.LIDAR.NUM_META_CHANNEL)
    if int(cfg.LIDAR.NUM_SLICES) + num_meta != int(cfg.LIDAR.NUM_CHANNEL):
        raise ValueError("cfg.LIDAR.NUM_CHANNEL must equal NUM_SLICES + NUM_META_CHANNEL")
    bev, _ = ops.bev_voxelize(points.contiguous(), pc_range, voxel_size, cfg.LIDAR.Z_RANGE[0],
                              cfg.LIDAR.MAX_PTS_PER_VOXEL, cfg.LIDAR.MAX_NUM_VOXEL, cfg.LIDAR.NUM_SLICES, num_meta,
                              -1 if elongation is None else int(elongation))
    if (bev.shape[0], bev.shape[1]) != (int(info[3]), int(info[1])):
        raise RuntimeError("voxel grid %s does not match the info vector %s" % (tuple(bev.shape), info.tolist()))
    return [info.tolist()], bev.unsqueeze(0)
