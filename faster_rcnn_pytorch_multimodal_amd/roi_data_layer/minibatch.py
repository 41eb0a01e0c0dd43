"""LiDAR input producer - counterpart of ``_get_lidar_blob`` (lib/roi_data_layer/minibatch.py:237-516) for ONE frame
whose points are already in memory: range filter (:232-235), voxelisation and the BEV scatter (:434-512) run as
``frcnn_bev_voxelize`` on the device; file parsing, FOV calibration and the augmentations (:250-431) stay with the
caller (dataset plumbing, out of scope).  Returns the same ``(infos, blob)`` the reference's data layer hands to
``Network.forward``: blob (1, num_y_voxel, num_x_voxel, cfg.LIDAR.NUM_CHANNEL) NHWC on the device,
info = [0, num_x_voxel, 0, num_y_voxel, 0, NUM_SLICES, scale].

The reference-named entry points (``_get_image_blob``, ``_get_lidar_blob``, ``get_minibatch``: what
``lib/model/test.py:32-44`` and ``lib/roi_data_layer/layer.py:66-82`` call) load ONE frame from a file and hand it to the
device producers (``frcnn_prep_image``, ``frcnn_bev_voxelize``).  Augmentation (imgaug, flips, rain simulation:
minibatch.py:250-431,542-664) is dataset tooling outside the accelerated path: ``augment_en=True`` raises.
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


# ---------------------------------------------------------------------------------------------------------------
# reference-named loaders
# ---------------------------------------------------------------------------------------------------------------
def read_image_file(filename):
    """uint8 (H, W, 3) array in BGR channel order - what ``cv2.imread`` returns (minibatch.py:529,532).  ``.npy`` files
    hold that array directly; anything else is decoded with PIL (cv2 is not a dependency of this package)."""
    if str(filename).endswith('.npy'):
        im = np.load(filename)
    else:
        from PIL import Image
        with Image.open(filename) as f:
            im = np.asarray(f.convert('RGB'))[:, :, ::-1]
    if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
        raise ValueError("%s: expected a uint8 (H, W, 3) image, got %s %s" % (filename, im.dtype, im.shape))
    return np.ascontiguousarray(im)


def read_point_cloud_file(filename):
    """(N, >=4) float32 rows [x, y, z, intensity, ...]: ``.bin`` = flat float32 quadruples, ``.npy`` as stored
    (minibatch.py:251-270).  The KITTI / CADC field-of-view filters need the datasets' calibration files (out of scope)."""
    if '.bin' in str(filename):
        if cfg.DB_NAME in ('kitti', 'cadc'):
            raise NotImplementedError("camera field-of-view filtering of %s scans needs the dataset's calibration tooling"
                                      % cfg.DB_NAME)
        return np.fromfile(filename, dtype=np.float32).reshape(-1, 4)
    if '.npy' in str(filename):
        return np.load(filename)
    raise ValueError('Cannot handle this type of binary file: %s' % filename)


def _no_augmentation(augment_en):
    if augment_en:
        raise NotImplementedError("augment_en=True: the imgaug / flip / rain augmentations of lib/roi_data_layer/minibatch.py "
                                  "are dataset tooling outside this package; pass augment_en=False")


def _get_image_blob(roidb, im_scale, augment_en=False, mode='train', device='cuda'):
    """minibatch.py:518-676.  ``roidb``: list with ONE filename (mode 'test') or ONE roidb entry (dict with 'filename').
    Returns (im_infos, blob (1, H', W', 3) float32 device tensor, local_roidb)."""
    from copy import deepcopy
    from ..utils.blob import im_list_to_blob, prep_im_for_blob
    _no_augmentation(augment_en)
    if len(roidb) != 1:
        raise NotImplementedError("single-frame batches only (minibatch.py:111)")
    if mode == 'test':
        im, local_roidb = read_image_file(roidb[0]), None
    else:
        im, local_roidb = read_image_file(roidb[0]['filename']), deepcopy(roidb)
        local_roidb[0]['flipped'] = False
    im = prep_im_for_blob(im, cfg.PIXEL_MEANS, cfg.PIXEL_STDDEVS, cfg.PIXEL_ARRANGE, im_scale, device=device)
    info = np.array([0, im.shape[1], 0, im.shape[0], 0, 0, im_scale], dtype=np.float32)          # :670
    return [info], im_list_to_blob([im]), local_roidb


def _get_lidar_blob(roidb, pc_extents, scale, augment_en=False, mode='train', device='cuda'):
    """minibatch.py:237-516 without the augmentations: file -> points -> ``get_lidar_blob`` (range filter, voxel
    generator and BEV scatter on the device).  ``pc_extents`` is what the reference passes (cfg.LIDAR.*_RANGE); the
    voxeliser reads the same ranges from cfg.  Waymo scans carry the elongation in column 4 (:496-499)."""
    from copy import deepcopy
    _no_augmentation(augment_en)
    if len(roidb) != 1:
        raise NotImplementedError("single-frame batches only (minibatch.py:111)")
    if mode == 'test':
        filen, local_roidb = roidb[0], None
    else:
        filen, local_roidb = roidb[0]['filename'], deepcopy(roidb)
        local_roidb[0]['flipped'] = False
    expected = [cfg.LIDAR.X_RANGE[0], cfg.LIDAR.Y_RANGE[0], cfg.LIDAR.Z_RANGE[0],
                cfg.LIDAR.X_RANGE[1], cfg.LIDAR.Y_RANGE[1], cfg.LIDAR.Z_RANGE[1]]
    if [float(v) for v in pc_extents] != [float(v) for v in expected]:
        raise ValueError("pc_extents %s differ from cfg.LIDAR.*_RANGE %s" % (list(pc_extents), expected))
    points = read_point_cloud_file(filen)
    elongation = 4 if (cfg.DB_NAME == 'waymo' and points.shape[1] > 4) else None
    infos, blob = get_lidar_blob(points, scale, device=device, elongation=elongation)
    return [np.asarray(infos[0], dtype=np.float32)], blob, local_roidb


def get_image_minibatch(roidb, num_classes, augment_en, scale, cnt):
    """minibatch.py:180-227: blobs {data, info, gt_boxes (G,5) [x1,y1,x2,y2,cls] scaled, gt_boxes_dc, filename};
    None when the frame has no ground truth left."""
    infos, im_blob, local_roidb = _get_image_blob(roidb, scale, augment_en)
    info, entry = infos[0], local_roidb[0]
    im_scale = info[6]
    gt_inds = np.where(np.asarray(entry['ignore']) == 0)[0]
    gt_boxes = np.empty((len(gt_inds), 5), dtype=np.float32)
    gt_boxes[:, 0:4] = np.asarray(entry['boxes'])[gt_inds, :] * im_scale
    gt_boxes[:, 4] = np.asarray(entry['gt_classes'])[gt_inds]
    dc = np.asarray(entry.get('boxes_dc', np.zeros((0, 4))), dtype=np.float32).reshape(-1, 4)
    gt_boxes_dc = np.empty((dc.shape[0], 5), dtype=np.float32)
    if cfg.TRAIN.IGNORE_DC:
        gt_boxes_dc[:, 0:4] = dc * im_scale
        gt_boxes_dc[:, 4] = 0
    blobs = {'data': im_blob, 'info': info, 'filename': entry['filename'], 'gt_boxes': gt_boxes,
             'gt_boxes_dc': gt_boxes_dc, 'flipped': False}
    return blobs if len(gt_boxes) else None


def get_lidar_minibatch(roidb, num_classes, augment_en, scale, cnt):
    """minibatch.py:124-178: gt rows (G,8) [xc,yc,zc,l,w,h,ry,cls] moved onto the voxel grid and scaled."""
    from ..utils.bbox import bbox_pc_to_voxel_grid
    extents = [cfg.LIDAR.X_RANGE[0], cfg.LIDAR.Y_RANGE[0], cfg.LIDAR.Z_RANGE[0],
               cfg.LIDAR.X_RANGE[1], cfg.LIDAR.Y_RANGE[1], cfg.LIDAR.Z_RANGE[1]]
    infos, pc_blob, local_roidb = _get_lidar_blob(roidb, extents, scale, augment_en)
    info, entry = infos[0], local_roidb[0]
    gt_inds = np.where(np.asarray(entry['ignore']) == 0)[0]
    width = cfg.LIDAR.NUM_BBOX_ELEM + 1
    gt_boxes = np.empty((len(gt_inds), width), dtype=np.float32)
    gt_boxes[:, 0:-1] = bbox_pc_to_voxel_grid(np.array(entry['boxes'], dtype=np.float64)[gt_inds, :], extents, info)
    gt_boxes[:, 0:2] *= scale
    gt_boxes[:, 3:5] *= scale
    gt_boxes[:, -1] = np.asarray(entry['gt_classes'])[gt_inds]
    blobs = {'data': pc_blob, 'flipped': False, 'filename': entry['filename'], 'gt_boxes': gt_boxes,
             'gt_boxes_dc': np.empty(0) * scale, 'info': np.array(info, dtype=np.float32)}
    return blobs if len(gt_boxes) else None


def get_minibatch(roidb, num_classes, augment_en, cnt):
    """minibatch.py:108-122."""
    assert len(roidb) == 1, "Single batch only"
    scale = cfg.TRAIN.SCALES[np.random.randint(0, high=len(cfg.TRAIN.SCALES), size=1)[0]]
    if cfg.NET_TYPE == 'image':
        return get_image_minibatch(roidb, num_classes, augment_en, scale, cnt)
    if cfg.NET_TYPE == 'lidar':
        return get_lidar_minibatch(roidb, num_classes, augment_en, scale, cnt)
    print('getting minibatch failed. Invalid NET TYPE in cfg')
    return None
