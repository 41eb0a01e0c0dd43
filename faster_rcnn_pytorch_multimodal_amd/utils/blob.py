"""Blob helpers on the device — counterpart of lib/utils/blob.py:16-54.

``prep_im_for_blob`` takes the uint8 HxWx3 image (as ``cv2.imread`` returns it, numpy or device tensor) and produces
the resized, channel-arranged, mean-subtracted float32 image on the MI355X in one ``frcnn_prep_image`` launch; with
``pad_to=4`` the blob is written with a zero fourth channel so ``Network.forward`` needs no padding pass.
"""
import ctypes

import numpy as np
import torch

from .. import _hip
from ..model.config import cfg


def prep_im_for_blob(im, pixel_means=None, pixel_stddev=None, pixel_arrange=None, im_scale=1.0, pad_to=3, device='cuda'):
    """Returns a (H', W', pad_to) float32 DEVICE tensor (reference: numpy (H', W', 3))."""
    lib = _hip.load()
    means = np.asarray(cfg.PIXEL_MEANS if pixel_means is None else pixel_means, dtype=np.float64).reshape(-1)
    stds = np.asarray(cfg.PIXEL_STDDEVS if pixel_stddev is None else pixel_stddev, dtype=np.float64).reshape(-1)
    arrange = [int(v) for v in (cfg.PIXEL_ARRANGE if pixel_arrange is None else pixel_arrange)]
    if isinstance(im, np.ndarray):
        im = torch.from_numpy(np.ascontiguousarray(im))
    if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3:
        raise _hip.HipError("prep_im_for_blob: expected a uint8 (H, W, 3) image, got %s %s" % (im.dtype, tuple(im.shape)))
    im = im.to(device).contiguous()
    h, w = int(im.shape[0]), int(im.shape[1])
    oh, ow = ctypes.c_int(), ctypes.c_int()
    _hip.check(lib.frcnn_prep_image_out_size(h, w, float(im_scale), ctypes.byref(oh), ctypes.byref(ow)),
               "frcnn_prep_image_out_size")
    blob = torch.empty((oh.value, ow.value, pad_to), dtype=torch.float32, device=im.device)
    _hip.check(lib.frcnn_prep_image(im.data_ptr(), h, w, float(im_scale), (ctypes.c_double * 3)(*means[:3]),
                                    (ctypes.c_double * 3)(*stds[:3]), (ctypes.c_int * 3)(*arrange), pad_to,
                                    blob.data_ptr(), torch.cuda.current_stream().cuda_stream), "frcnn_prep_image")
    return blob


def im_list_to_blob(ims):
    """One frame per batch on this path (lib/roi_data_layer/minibatch.py:111): (1, H, W, C) view of the prepared image."""
    if len(ims) != 1:
        raise NotImplementedError("single-frame batches only (README.md:29 of the reference: no real batching)")
    return ims[0].unsqueeze(0)


def image_info(blob, im_scale):
    """The 7-vector the network expects (lib/roi_data_layer/minibatch.py:670)."""
    return np.array([0, blob.shape[-2], 0, blob.shape[-3], 0, 0, im_scale], dtype=np.float32)
