"""Dense anchor grid on the device — counterpart of the reference's lib/layer_utils/snippets.py:13-40.

The reference rebuilds the (H*W*A, 4) array with numpy on every frame and copies it to the GPU; here the
A x 4 float64 base table is computed once on the host and the shift grid is applied by
``frcnn_generate_anchors`` on the device, cached per (H, W, stride, scales, ratios, frame_scale, device).
"""
import numpy as np
import torch

from .. import ops
from .generate_anchors import generate_anchors

_CACHE = {}


def generate_anchors_pre(height, width, feat_stride, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2),
                         frame_scale=1.0, device='cuda'):
    """Returns (anchors, length): anchors is a (H*W*A, 4) float32 DEVICE tensor in (H, W, A) order with A
    fastest (snippets.py:35-37), length = H*W*A (np.int32 like the reference)."""
    scales = tuple(float(s) for s in np.asarray(anchor_scales, dtype=np.float64).ravel())
    ratios = tuple(float(r) for r in np.asarray(anchor_ratios, dtype=np.float64).ravel())
    device = torch.device(device)
    if device.type == 'cuda' and device.index is None:
        device = torch.device('cuda', torch.cuda.current_device())
    key = (int(height), int(width), int(feat_stride), scales, ratios, float(frame_scale), str(device))
    hit = _CACHE.get(key)
    if hit is None:
        base = generate_anchors(ratios=np.array(ratios), scales=np.array(scales) * frame_scale)  # snippets.py:22
        base_dev = torch.from_numpy(np.ascontiguousarray(base)).to(device)
        hit = ops.generate_anchors(base_dev, int(height), int(width), int(feat_stride))
        _CACHE[key] = hit
    return hit, np.int32(hit.shape[0])
