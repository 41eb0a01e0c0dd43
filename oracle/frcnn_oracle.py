"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

A plain numpy / PyTorch-CPU restatement of the reference algorithm for the Faster R-CNN hot path of
mathild7/faster_rcnn_pytorch_multimodal.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
(``faster_rcnn_pytorch_multimodal_amd``) never does and has no CPU execution path.

Pinning status:
  * pinned by golden vectors generated from the importable reference modules
    (tests/golden/make_golden.py, make_golden_lidar_train.py, make_golden_eval.py): generate_anchors /
    generate_anchors_pre, bbox_transform(_inv), clip_boxes, ResNet-101 stage outputs, bbox_overlaps, the 3-D anchor
    grid and bbaa_graphics_gems, the LiDAR codec, proposal_top_layer, anchor / proposal target layers (image and
    LiDAR), huber / smooth-L1 (RPN, DET, LiDAR-DET, aleatoric), the MC statistics, voc_eval;
  * PARITY UNPINNED: nms and roi_align restate the documented semantics of torchvision==0.4.0
    (req.txt:283), prep_im_for_blob restates cv2.resize, points_to_voxel restates spconv's voxel generator - none
    of them is vendored in the reference or installed here, and the reference has no tests for them; the Network
    pipeline restates the RECONSTRUCTED contract of the missing lib/nets/network.py.

Every function cites the reference file:line it follows (paths relative to the reference root).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------------
# constants of lib/model/config.py used on the path
# ----------------------------------------------------------------------------------------------
ANCHOR_SCALES = (2, 4, 8, 16, 32)             # config.py:373
ANCHOR_RATIOS = (0.5, 0.75, 1, 1.25, 2)       # config.py:378
RPN_CHANNELS = 512                            # config.py:381
POOLING_SIZE = 7                              # config.py:366
TEST_RPN_PRE_NMS_TOP_N = 6000                 # config.py:253
TEST_RPN_POST_NMS_TOP_N = 300                 # config.py:256
TEST_RPN_NMS_THRESH = 0.7                     # config.py:250
TEST_NMS_THRESH = 0.6                         # config.py:234
BBOX_NORMALIZE_MEANS = (0.0, 0.0, 0.0, 0.0)   # config.py:222
BBOX_NORMALIZE_STDS = (0.1, 0.1, 0.2, 0.2)    # config.py:223
ROI_ALIGN_SAMPLING_RATIO = 0                  # unpinned (network.py missing); ancestor convention: adaptive


# ----------------------------------------------------------------------------------------------
# anchors — lib/layer_utils/generate_anchors.py:41-105, lib/layer_utils/snippets.py:13-40
# ----------------------------------------------------------------------------------------------
def _whctrs(a):
    w = a[2] - a[0] + 1
    h = a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _mkanchors(ws, hs, xc, yc):
    ws = np.asarray(ws, dtype=np.float64)[:, None]
    hs = np.asarray(hs, dtype=np.float64)[:, None]
    return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1), xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)):
    """generate_anchors.py:41-54: ratio enumeration (np.round, :82-93) then scale enumeration (:96-105)."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    base = np.array([1, 1, base_size, base_size], dtype=np.float64) - 1
    w, h, xc, yc = _whctrs(base)
    size_ratios = (w * h) / ratios
    ws = np.round(np.sqrt(size_ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = _mkanchors(ws, hs, xc, yc)
    out = []
    for i in range(ratio_anchors.shape[0]):
        w, h, xc, yc = _whctrs(ratio_anchors[i])
        out.append(_mkanchors(w * scales, h * scales, xc, yc))
    return np.vstack(out)


def generate_anchors_pre(height, width, feat_stride, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2),
                         frame_scale=1.0):
    """snippets.py:13-40: (H, W, A) layout, A fastest; float64 sum cast once to float32."""
    base = generate_anchors(ratios=np.array(anchor_ratios), scales=np.array(anchor_scales) * frame_scale)
    a = base.shape[0]
    sx = np.arange(0, width) * feat_stride
    sy = np.arange(0, height) * feat_stride
    sx, sy = np.meshgrid(sx, sy)
    shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
    k = shifts.shape[0]
    anchors = base.reshape((1, a, 4)) + shifts.reshape((1, k, 4)).transpose((1, 0, 2))
    return anchors.reshape((k * a, 4)).astype(np.float32, copy=False), np.int32(k * a)


# ----------------------------------------------------------------------------------------------
# box codec — lib/model/bbox_transform.py:52-70, 75-105, 235-257
# ----------------------------------------------------------------------------------------------
def bbox_transform(ex_rois, gt_rois):
    ew = ex_rois[:, 2] - ex_rois[:, 0] + 1.0
    eh = ex_rois[:, 3] - ex_rois[:, 1] + 1.0
    diag = torch.sqrt(torch.pow(ew, 2) + torch.pow(eh, 2))
    ecx = ex_rois[:, 0] + 0.5 * ew
    ecy = ex_rois[:, 1] + 0.5 * eh
    gw = gt_rois[:, 2] - gt_rois[:, 0] + 1.0
    gh = gt_rois[:, 3] - gt_rois[:, 1] + 1.0
    gcx = gt_rois[:, 0] + 0.5 * gw
    gcy = gt_rois[:, 1] + 0.5 * gh
    return torch.stack(((gcx - ecx) / diag, (gcy - ecy) / diag, torch.log(gw / ew), torch.log(gh / eh)), 1)


def bbox_transform_inv(boxes, deltas, scales=None):
    if scales is not None:
        boxes = boxes / scales
    if len(boxes) == 0:
        return deltas.detach() * 0
    w = boxes[:, 2] - boxes[:, 0] + 1.0
    h = boxes[:, 3] - boxes[:, 1] + 1.0
    diag = torch.sqrt(torch.pow(w, 2) + torch.pow(h, 2))
    cx = boxes[:, 0] + 0.5 * w
    cy = boxes[:, 1] + 0.5 * h
    dx, dy, dw, dh = deltas[:, 0::4], deltas[:, 1::4], deltas[:, 2::4], deltas[:, 3::4]
    pcx = dx * diag.unsqueeze(1) + cx.unsqueeze(1)
    pcy = dy * diag.unsqueeze(1) + cy.unsqueeze(1)
    pw = torch.exp(dw) * w.unsqueeze(1)
    ph = torch.exp(dh) * h.unsqueeze(1)
    parts = [pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph]
    return torch.cat([p.unsqueeze(2) for p in parts], 2).view(len(boxes), -1)


def clip_boxes(boxes, info):
    """bbox_transform.py:252-255: x in [info[0], info[1]-1], y in [info[2], info[3]-1]."""
    b = boxes.view(boxes.size(0), -1, 4)
    info = np.asarray(info, dtype=np.float32)
    xl, xh, yl, yh = float(info[0]), float(info[1] - np.float32(1)), float(info[2]), float(info[3] - np.float32(1))
    return torch.stack([b[:, :, 0].clamp(xl, xh), b[:, :, 1].clamp(yl, yh), b[:, :, 2].clamp(xl, xh),
                        b[:, :, 3].clamp(yl, yh)], 2).view(boxes.size(0), -1)


def bbox_overlaps(boxes, query_boxes):
    """lib/utils/bbox.py:5-33: pairwise IoU with the +1 area convention."""
    boxes = torch.as_tensor(boxes, dtype=torch.float32)
    q = torch.as_tensor(query_boxes, dtype=torch.float32)
    ba = (boxes[:, 2] - boxes[:, 0] + 1) * (boxes[:, 3] - boxes[:, 1] + 1)
    qa = (q[:, 2] - q[:, 0] + 1) * (q[:, 3] - q[:, 1] + 1)
    iw = (torch.min(boxes[:, 2:3], q[:, 2:3].t()) - torch.max(boxes[:, 0:1], q[:, 0:1].t()) + 1).clamp(min=0)
    ih = (torch.min(boxes[:, 3:4], q[:, 3:4].t()) - torch.max(boxes[:, 1:2], q[:, 1:2].t()) + 1).clamp(min=0)
    ua = ba.view(-1, 1) + qa.view(1, -1) - iw * ih
    return iw * ih / ua


# ----------------------------------------------------------------------------------------------
# torchvision.ops.nms (0.4.0, un-vendored; PARITY UNPINNED) — call sites proposal_layer.py:46,
# filter_predictions.py:67-69.  Greedy, areas without +1, drop when IoU > thresh, survivors returned in
# descending-score order; ties in score broken by ascending index (canonical order of this build).
# ----------------------------------------------------------------------------------------------
def stable_desc_order(scores):
    s = torch.as_tensor(scores, dtype=torch.float32).reshape(-1)
    return torch.sort(s, descending=True, stable=True)[1]


# IoU == threshold exactly: torchvision 0.4.0's CPU kernel suppresses (`>=`), its CUDA kernel (and every later release) keeps
# the box (`>`).  The module-level switch mirrors the device library's run-time setting (frcnn_nms_set_suppress_at_equal);
# the default is the CPU kernel's, the path the reference is compared against.  torchvision is not available here, so the
# choice is recorded, not verified (PARITY UNPINNED, see the header); on real-valued inputs the two forms differ on a set of
# measure zero.
NMS_SUPPRESS_AT_EQUAL = True


def nms(boxes, scores, thresh):
    boxes = torch.as_tensor(boxes, dtype=torch.float32)
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64)
    order = stable_desc_order(scores).numpy()
    b = boxes.numpy()[order]
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    thr = np.float32(thresh)
    dead = np.zeros(n, dtype=bool)
    keep = []
    with np.errstate(invalid="ignore", divide="ignore"):
        for i in range(n):
            if dead[i]:
                continue
            keep.append(order[i])
            if i + 1 == n:
                break
            xx1 = np.maximum(x1[i], x1[i + 1:])
            yy1 = np.maximum(y1[i], y1[i + 1:])
            xx2 = np.minimum(x2[i], x2[i + 1:])
            yy2 = np.minimum(y2[i], y2[i + 1:])
            w = np.maximum(np.float32(0), xx2 - xx1)
            h = np.maximum(np.float32(0), yy2 - yy1)
            inter = w * h
            ovr = inter / (areas[i] + areas[i + 1:] - inter)
            dead[i + 1:] |= (ovr >= thr) if NMS_SUPPRESS_AT_EQUAL else (ovr > thr)
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


# ----------------------------------------------------------------------------------------------
# proposal_layer — lib/layer_utils/proposal_layer.py:18-57
# ----------------------------------------------------------------------------------------------
def proposal_layer(rpn_cls_prob, rpn_bbox_pred, info, anchors, num_anchors, pre_nms_top_n=TEST_RPN_PRE_NMS_TOP_N,
                   post_nms_top_n=TEST_RPN_POST_NMS_TOP_N, nms_thresh=TEST_RPN_NMS_THRESH, return_debug=False):
    """rpn_cls_prob (1,H,W,2A) with fg = [..., A:] (:32); rpn_bbox_pred (1,H,W,4A) viewed (-1,4) (:33)."""
    scores = rpn_cls_prob[:, :, :, num_anchors:].contiguous().view(-1)
    deltas = rpn_bbox_pred.reshape(-1, 4)
    proposals = clip_boxes(bbox_transform_inv(anchors, deltas), info)
    order = stable_desc_order(scores)       # :39 (sort is unstable in torch 1.2; canonical order here)
    if pre_nms_top_n > 0:
        order = order[:pre_nms_top_n]
    sorted_scores = scores[order]
    sorted_props = proposals[order]
    keep = nms(sorted_props, sorted_scores, nms_thresh)
    if post_nms_top_n > 0:
        keep = keep[:post_nms_top_n]
    rois = torch.cat((sorted_props.new_zeros(len(keep), 1), sorted_props[keep]), 1)
    out_scores = sorted_scores[keep].view(-1, 1)
    if return_debug:
        return rois, out_scores, {"scores": scores, "proposals": proposals, "order": order, "keep": keep}
    return rois, out_scores


# ----------------------------------------------------------------------------------------------
# torchvision.ops.roi_align 0.4.0 (aligned=False; PARITY UNPINNED) — call sites
# lib/utils/torchpoolers.py:165-170,194-197.  feat (1,C,H,W); rois (R,5); out (R,C,P,P).
# The (iy, ix) accumulation order and the per-sample expression order follow the library's CPU kernel.
# ----------------------------------------------------------------------------------------------
def roi_align(feat, rois, pooled, spatial_scale, sampling_ratio=ROI_ALIGN_SAMPLING_RATIO):
    f = feat.detach().numpy().astype(np.float32, copy=False)
    r_np = rois.detach().numpy().astype(np.float32, copy=False)
    _, c, hgt, wid = f.shape
    out = np.zeros((r_np.shape[0], c, pooled, pooled), dtype=np.float32)
    scale = np.float32(spatial_scale)
    pf = np.float32(pooled)
    ph_idx = np.arange(pooled, dtype=np.float32)
    for r in range(r_np.shape[0]):
        b = int(r_np[r, 0])
        sw, sh = r_np[r, 1] * scale, r_np[r, 2] * scale
        ew, eh = r_np[r, 3] * scale, r_np[r, 4] * scale
        rw = max(ew - sw, np.float32(1.0))
        rh = max(eh - sh, np.float32(1.0))
        bh, bw = np.float32(rh / pf), np.float32(rw / pf)
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(np.float32(rh / pf)))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(np.float32(rw / pf)))
        count = np.float32(gh * gw)
        acc = np.zeros((c, pooled, pooled), dtype=np.float32)
        fm = f[b]
        for iy in range(gh):
            y = (sh + ph_idx * bh) + (np.float32(iy) + np.float32(0.5)) * bh / np.float32(gh)  # (P,)
            for ix in range(gw):
                x = (sw + ph_idx * bw) + (np.float32(ix) + np.float32(0.5)) * bw / np.float32(gw)  # (P,)
                yy, xx = np.meshgrid(y, x, indexing="ij")
                empty = (yy < -1.0) | (yy > hgt) | (xx < -1.0) | (xx > wid)
                yy = np.where(yy <= 0, np.float32(0), yy).astype(np.float32)
                xx = np.where(xx <= 0, np.float32(0), xx).astype(np.float32)
                yl, xl = yy.astype(np.int32), xx.astype(np.int32)
                ytop, xtop = yl >= hgt - 1, xl >= wid - 1
                yl = np.where(ytop, hgt - 1, yl)
                xl = np.where(xtop, wid - 1, xl)
                yh_ = np.where(ytop, hgt - 1, yl + 1)
                xh_ = np.where(xtop, wid - 1, xl + 1)
                yy = np.where(ytop, yl.astype(np.float32), yy)
                xx = np.where(xtop, xl.astype(np.float32), xx)
                ly, lx = yy - yl.astype(np.float32), xx - xl.astype(np.float32)
                hy, hx = np.float32(1.0) - ly, np.float32(1.0) - lx
                w1, w2, w3, w4 = hy * hx, hy * lx, ly * hx, ly * lx
                val = w1 * fm[:, yl, xl] + w2 * fm[:, yl, xh_] + w3 * fm[:, yh_, xl] + w4 * fm[:, yh_, xh_]
                acc += np.where(empty[None], np.float32(0), val)
        out[r] = acc / count
    return torch.from_numpy(out)


# ----------------------------------------------------------------------------------------------
# ResNet-101, caffe stride placement — lib/nets/resnet.py:74-128 (Bottleneck), 131-224 (ResNet),
# 227-240 (ResNetWrapper stride edits), 275-284 (resnet101).  State-dict keys equal the reference's.
# ----------------------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, batchnorm_en=True):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.batchnorm_en = batchnorm_en

    def forward(self, x):
        out = self.conv1(x)
        if self.batchnorm_en:
            out = self.bn1(out)
        out = F.relu(out)
        out = self.conv2(out)
        if self.batchnorm_en:
            out = self.bn2(out)
        out = F.relu(out)
        out = self.conv3(out)
        if self.batchnorm_en:
            out = self.bn3(out)
        identity = x if self.downsample is None else self.downsample(x)
        return F.relu(out + identity)


class ResNet101(nn.Module):
    """Module tree named like the reference's ``resnet`` attribute (conv1, bn1, layer1..layer4)."""

    def __init__(self, in_channels=3, use_fpn=False, batchnorm_en=True, blocks=(3, 4, 23, 3)):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1 = self._make_layer(64, blocks[0], 1, True)
        self.layer2 = self._make_layer(128, blocks[1], 2, True)
        self.layer3 = self._make_layer(256, blocks[2], 2, True)
        self.layer4 = self._make_layer(512, blocks[3], 2, batchnorm_en)
        for i in (2, 3):  # resnet.py:232-234: stride on the first 1x1
            blk = getattr(self, "layer%d" % i)[0]
            blk.conv1.stride = (2, 2)
            blk.conv2.stride = (1, 1)
        if not use_fpn:   # resnet.py:236-238
            self.layer4[0].conv2.stride = (1, 1)
            self.layer4[0].downsample[0].stride = (1, 1)

    def _make_layer(self, planes, blocks, stride, bn):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down, bn)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes, batchnorm_en=bn) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def stem(self, x):
        return F.max_pool2d(F.relu(self.bn1(self.conv1(x))), kernel_size=3, stride=2, padding=1)


# Name-keyed seeded weight initialiser shared with bench.py (lives in the product package because the
# benchmark's GPU leg must not import the oracle; it is an initialiser, not part of the algorithm).
from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict  # noqa: E402,F401


# ----------------------------------------------------------------------------------------------
# Network (image detector, non-FPN) — RECONSTRUCTED contract of the missing lib/nets/network.py
#  + lib/nets/imagenet.py:29-134.
# ----------------------------------------------------------------------------------------------
class ImageNetOracle(nn.Module):
    def __init__(self, num_classes=2, anchor_scales=ANCHOR_SCALES, anchor_ratios=ANCHOR_RATIOS, in_channels=3,
                 num_layers=101):
        super().__init__()
        self._num_classes = num_classes
        self._anchor_scales = tuple(anchor_scales)
        self._anchor_ratios = tuple(anchor_ratios)
        self._num_anchors = len(anchor_scales) * len(anchor_ratios)
        self._feat_stride = 16                       # imagenet.py:44
        # resnet.py:275-292: the bottleneck depths the detectors can be built on
        self.resnet = ResNet101(in_channels=in_channels,
                                blocks={50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}[num_layers])
        a = self._num_anchors
        self.rpn_net = nn.Conv2d(1024, RPN_CHANNELS, 3, padding=1)
        self.rpn_cls_score_net = nn.Conv2d(RPN_CHANNELS, 2 * a, 1)
        self.rpn_bbox_pred_net = nn.Conv2d(RPN_CHANNELS, 4 * a, 1)
        self.cls_score_net = nn.Linear(2048, num_classes)
        self.bbox_pred_net = nn.Linear(2048, num_classes * 4)
        self.eval()

    # imagenet.py:131-134: head = conv1, bn1, relu, maxpool, layer1..3
    def _image_to_head(self, image):
        r = self.resnet
        return r.layer3(r.layer2(r.layer1(r.stem(image))))

    def _region_proposal(self, net_conv, info, structured=None):
        a = self._num_anchors
        h, w = net_conv.shape[2], net_conv.shape[3]
        anchors, _ = generate_anchors_pre(h, w, self._feat_stride, self._anchor_scales, self._anchor_ratios,
                                          float(info[6]))
        anchors = torch.from_numpy(anchors)
        rpn = F.relu(self.rpn_net(net_conv))
        cls_score = self.rpn_cls_score_net(rpn)                      # (1, 2A, H, W)
        bbox_pred = self.rpn_bbox_pred_net(rpn).permute(0, 2, 3, 1).contiguous()  # (1, H, W, 4A)
        if structured is not None:                                   # injected logits / deltas
            cls_score, bbox_pred = structured
        # (1,2A,H,W) -> (1,2,A*H,W) softmax over dim 1 -> back -> (1,H,W,2A): channel a pairs with a+A
        prob = F.softmax(cls_score.view(1, 2, a * h, w), dim=1).view(1, 2 * a, h, w).permute(0, 2, 3, 1).contiguous()
        rois, scores, dbg = proposal_layer(prob, bbox_pred, info, anchors, a, return_debug=True)
        self._dbg = {"anchors": anchors, "rpn_cls_prob": prob, "rpn_bbox_pred": bbox_pred, "rpn_cls_score": cls_score,
                     **dbg}
        return rois, scores

    def _crop_pool_layer(self, net_conv, rois):
        return roi_align(net_conv, rois, POOLING_SIZE, 1.0 / self._feat_stride, ROI_ALIGN_SAMPLING_RATIO)

    def _head_to_tail(self, pool5):
        return self.resnet.layer4(pool5).mean(3).mean(2)

    def _region_classification(self, fc7):
        cls_score = self.cls_score_net(fc7)
        return cls_score, F.softmax(cls_score, dim=1), self.bbox_pred_net(fc7)

    @torch.no_grad()
    def test_frame(self, data, info, structured=None):
        """data (1,H,W,C) NHWC float32 numpy blob, info 7-vector (minibatch.py:670).
        Returns (cls_score, cls_prob, pred_boxes, rois, uncertainties={})."""
        image = torch.from_numpy(np.ascontiguousarray(data)).permute(0, 3, 1, 2).contiguous()
        net_conv = self._image_to_head(image)
        rois, _ = self._region_proposal(net_conv, info, structured)
        pool5 = self._crop_pool_layer(net_conv, rois)
        fc7 = self._head_to_tail(pool5)
        cls_score, cls_prob, bbox_pred = self._region_classification(fc7)
        stds = torch.tensor(BBOX_NORMALIZE_STDS).repeat(self._num_classes).unsqueeze(0)
        means = torch.tensor(BBOX_NORMALIZE_MEANS).repeat(self._num_classes).unsqueeze(0)
        deltas = bbox_pred.mul(stds).add(means)
        pred_boxes = bbox_transform_inv(rois[:, 1:5], deltas, float(info[6]))
        self._dbg.update({"net_conv": net_conv, "pool5": pool5, "fc7": fc7, "bbox_pred": bbox_pred})
        return cls_score, cls_prob, pred_boxes, rois, {}


def _image_train_forward(self, data, info, gt_boxes, generator=None, pre_nms=12000, post_nms=2000, proposals=None):
    """One TRAIN forward of the non-FPN image detector (layer4 tail): losses with a graph + the sampled targets.
    Uses the differentiable roi_align_torch; RECONSTRUCTED like FpnNetOracle.train_forward below."""
    image = torch.from_numpy(np.ascontiguousarray(data)).permute(0, 3, 1, 2).contiguous()
    gt = torch.as_tensor(gt_boxes, dtype=torch.float32)
    net_conv = self._image_to_head(image)
    a, (h, w) = self._num_anchors, net_conv.shape[2:]
    anchors = torch.from_numpy(generate_anchors_pre(h, w, self._feat_stride, self._anchor_scales, self._anchor_ratios,
                                                    float(info[6]))[0])
    rpn = F.relu(self.rpn_net(net_conv))
    cls_score = self.rpn_cls_score_net(rpn)
    bbox_pred = self.rpn_bbox_pred_net(rpn).permute(0, 2, 3, 1).contiguous()
    with torch.no_grad():
        prob = F.softmax(cls_score.view(1, 2, a * h, w), dim=1).view(1, 2 * a, h, w).permute(0, 2, 3, 1).contiguous()
        rois, scores = proposal_layer(prob, bbox_pred, info, anchors, a, pre_nms, post_nms, TEST_RPN_NMS_THRESH)
        if proposals is not None:
            rois, scores = proposals
        lab, tgt, inw, outw = anchor_target_layer(gt, info, anchors, a, h, w, generator=generator)
        pl, prois, _, _, ptgt, pin, pout = proposal_target_layer(rois, scores, torch.zeros(rois.shape[0], 7), gt, None,
                                                                 self._num_classes, 4, generator=generator)
    logits = torch.stack((cls_score[0, :a].permute(1, 2, 0).reshape(-1), cls_score[0, a:].permute(1, 2, 0).reshape(-1)), 1)
    labels_hwa = lab[0].permute(1, 2, 0).reshape(-1)
    sel = labels_hwa >= 0
    rpn_ce = F.cross_entropy(logits[sel], labels_hwa[sel].long())
    rpn_box = smooth_l1_loss("RPN", bbox_pred, tgt, inw, outw, dim=(1, 2, 3))
    pool5 = roi_align_torch(net_conv, prois, POOLING_SIZE, 1.0 / self._feat_stride)
    fc7 = self.resnet.layer4(pool5).mean(3).mean(2)
    det_cls, det_box = self.cls_score_net(fc7), self.bbox_pred_net(fc7)
    ce = F.cross_entropy(det_cls, pl.view(-1).long())
    box = smooth_l1_loss("DET", det_box, ptgt, pin, pout)
    losses = {"rpn_cross_entropy": rpn_ce, "rpn_loss_box": rpn_box, "cross_entropy": ce, "loss_box": box,
              "total_loss": rpn_ce + rpn_box + ce + box}
    dbg = {"anchor_labels": labels_hwa, "anchor_targets": tgt.reshape(-1, 4), "anchor_inside": inw.reshape(-1, 4),
           "anchor_outside": outw.reshape(-1, 4), "rois": prois, "labels": pl.view(-1), "targets": ptgt, "inside": pin,
           "outside": pout, "net_conv": net_conv, "fc7": fc7, "cls_score": det_cls}
    return losses, dbg


def _image_set_trainable(self, fixed_blocks=1):
    """imagenet.py:96-116: stem and layerN (N <= FIXED_BLOCKS) frozen, every BatchNorm frozen."""
    frozen = [self.resnet.conv1, self.resnet.bn1] + [getattr(self.resnet, "layer%d" % n) for n in (1, 2, 3)
                                                     if fixed_blocks >= n]
    for m in frozen:
        for p in m.parameters():
            p.requires_grad = False
    for m in self.resnet.modules():
        if isinstance(m, nn.BatchNorm2d):
            for p in m.parameters():
                p.requires_grad = False


ImageNetOracle.train_forward = _image_train_forward
ImageNetOracle.set_trainable = _image_set_trainable


# ----------------------------------------------------------------------------------------------
# filter_and_draw_prep / nms_hstack_torch — lib/utils/filter_predictions.py:45-130, and the max_dets
# cut of lib/model/test.py:210-221.  Image detector only here.
# ----------------------------------------------------------------------------------------------
def filter_and_draw_prep(rois, cls_prob, pred_boxes, info, num_classes, thresh=0.1, nms_thresh=TEST_NMS_THRESH):
    info = np.asarray(info, dtype=np.float32)
    fw, fh, scale = info[1] - info[0], info[3] - info[2], info[6]
    pred_boxes = pred_boxes.clone()
    pred_boxes[:, 0::4] = torch.clamp_min(pred_boxes[:, 0::4], 0)
    pred_boxes[:, 1::4] = torch.clamp_min(pred_boxes[:, 1::4], 0)
    pred_boxes[:, 2::4] = torch.clamp_max(pred_boxes[:, 2::4], float(fw / scale - np.float32(1)))
    pred_boxes[:, 3::4] = torch.clamp_max(pred_boxes[:, 3::4], float(fh / scale - np.float32(1)))
    all_boxes = [np.empty((0, 5), dtype=np.float32) for _ in range(num_classes)]
    for j in range(1, num_classes):
        inds = torch.where(cls_prob[:, j] > thresh)[0]
        if inds.numel() == 0:
            continue
        cs = cls_prob[inds, j]
        cb = pred_boxes[inds, j * 4:(j + 1) * 4]
        dets = np.hstack((cb.numpy(), cs.unsqueeze(1).numpy())).astype(np.float32, copy=False)
        keep = nms(cb, cs, nms_thresh).numpy()
        all_boxes[j] = dets[keep, :]
    return rois[:, 1:5].numpy(), all_boxes, pred_boxes


def max_dets_cut(cls_boxes, max_dets):
    """test.py:213-221: keep score >= the max_dets-th best (ties stay)."""
    if max_dets > 0 and len(cls_boxes) > max_dets:
        cut = np.sort(cls_boxes[:, -1])[-max_dets]
        cls_boxes = cls_boxes[np.where(cls_boxes[:, -1] >= cut)[0], :]
    return cls_boxes


def frame_detect(net, data, info, num_classes, thresh=0.5, max_dets=100, structured=None):
    """lib/model/test.py:68-93 + :210-221 for one frame -> list over classes of (n,5) arrays."""
    _, probs, boxes, rois, _ = net.test_frame(data, info, structured)
    _, all_boxes, _ = filter_and_draw_prep(rois, probs, boxes, info, num_classes, thresh)
    return [max_dets_cut(b, max_dets) for b in all_boxes]


# ----------------------------------------------------------------------------------------------
# VOC-style AP (lib/datasets/voc_eval.py:53-69, continuous-area rule) for the "mAP delta" report.
# ----------------------------------------------------------------------------------------------
def voc_ap(rec, prec):
    mrec = np.concatenate(([0.0], rec, [1.0]))
    mpre = np.concatenate(([0.0], prec, [0.0]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return float(np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1]))


def average_precision(dets, gt_boxes, iou_thresh=0.7):
    """dets (n,5) [x1,y1,x2,y2,score]; gt (g,4).  Greedy matching by descending score, +1 IoU convention."""
    if len(gt_boxes) == 0:
        return 0.0
    if len(dets) == 0:
        return 0.0
    order = np.argsort(-dets[:, 4], kind="stable")
    dets = dets[order]
    ious = bbox_overlaps(dets[:, :4], gt_boxes).numpy()
    used = np.zeros(len(gt_boxes), dtype=bool)
    tp = np.zeros(len(dets))
    fp = np.zeros(len(dets))
    for d in range(len(dets)):
        j = int(np.argmax(ious[d]))
        if ious[d, j] >= iou_thresh and not used[j]:
            tp[d] = 1
            used[j] = True
        else:
            fp[d] = 1
    tp, fp = np.cumsum(tp), np.cumsum(fp)
    rec = tp / float(len(gt_boxes))
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return voc_ap(rec, prec)


# ==============================================================================================
# LiDAR-BEV variant
# ==============================================================================================
LIDAR_X_RANGE, LIDAR_Y_RANGE, LIDAR_Z_RANGE = (0, 70), (-40, 40), (-3, 3)   # config.py:397-399
LIDAR_VOXEL_LEN, LIDAR_VOXEL_HEIGHT = 0.1, 0.5                              # config.py:400-401
LIDAR_NUM_CHANNEL = 15                                                      # config.py:402-404
LIDAR_ANCHORS = np.array([[4.73, 2.08, 1.77]])                              # config.py:421
LIDAR_ANCHOR_SCALES = (1,)                                                  # config.py:422
LIDAR_ANCHOR_ANGLES = np.array([0, np.pi / 2])                              # config.py:423
LIDAR_NUM_BBOX_ELEM = 7                                                     # config.py:425
LIDAR_REG_LOSS_WEIGHT = (1.0,) * 7                                          # config.py:426
LIDAR_BBOX_NORMALIZE_MEANS = (0.0,) * 7                                     # config.py:219
LIDAR_BBOX_NORMALIZE_STDS = (0.1, 0.1, 0.1, 0.2, 0.2, 0.2, 1.0)             # config.py:220


def generate_anchors_3d(height, width, feature_stride, anchor_scales=LIDAR_ANCHOR_SCALES,
                        anchor_rotations=LIDAR_ANCHOR_ANGLES, frame_scale=1.0):
    """generate_3d_anchors.py:15-118: grid of [x, y, z=h/2, l, w, h, ry] in voxel units, order (H, W, size, rot)
    with rot fastest; centres arange(0, W*stride-1, stride) (float32)."""
    assert len(anchor_scales) == 1
    voxel_len = LIDAR_VOXEL_LEN / frame_scale                                    # :37
    sizes = LIDAR_ANCHORS / np.array([voxel_len, voxel_len, 1]) * anchor_scales[0]  # :38
    rots = np.asarray(anchor_rotations)
    xs = np.arange(0, width * feature_stride - 1, feature_stride).astype(np.float32)
    ys = np.arange(0, height * feature_stride - 1, feature_stride).astype(np.float32)
    n = len(ys) * len(xs) * len(sizes) * len(rots)
    out = np.zeros((len(ys), len(xs), len(sizes), len(rots), 7), dtype=np.float32)
    out[..., 0] = xs[None, :, None, None]
    out[..., 1] = ys[:, None, None, None]
    out[..., 2] = sizes[0][2] / 2.0                                              # :100 (first size only)
    out[..., 3:6] = sizes[None, None, :, None, :]
    out[..., 6] = rots[None, None, None, :]
    return n, out.reshape(n, 7)


def bbaa_graphics_gems(bboxes):
    """utils/bbox.py:256-293 with clip=False: axis-aligned BEV box of [xc,yc,zc,l,w,h,ry].  The rotation
    matrix keeps the dtype of the boxes (float32), the half extents are float64, the sum of per-axis
    min/max products is cast to float32 and the float32 centre is added last."""
    b = np.asarray(bboxes)
    rot = b[:, 6]
    c, s = np.cos(rot), np.sin(rot)                          # dtype of b
    m = np.stack((np.stack((c, s), 1), np.stack((-s, c), 1)), 1)      # (N, 2, 2) = [[cos, sin], [-sin, cos]]
    amin = np.stack((-(b[:, 3] / 2.0), -(b[:, 4] / 2.0)), 1).astype(np.float64)
    amax = -amin
    lo = m * amin[:, None, :]
    hi = m * amax[:, None, :]
    bmin = np.minimum(lo, hi).sum(2).astype(np.float32) + b[:, 0:2]
    bmax = np.maximum(lo, hi).sum(2).astype(np.float32) + b[:, 0:2]
    return np.concatenate((bmin[:, 0:1], bmin[:, 1:2], bmax[:, 0:1], bmax[:, 1:2]), 1)


def lidar_3d_bbox_transform(ex_rois, ex_anchors, gt_rois):
    """bbox_transform.py:16-49: centre deltas over the RoI diagonal, z/h from the 3-D anchor, ry = raw gt yaw."""
    ln = ex_rois[:, 2] - ex_rois[:, 0] + 1
    wd = ex_rois[:, 3] - ex_rois[:, 1] + 1
    ht = ex_anchors[:, 5]
    cx, cy, cz = ex_rois[:, 0] + ln / 2.0, ex_rois[:, 1] + wd / 2.0, ex_anchors[:, 2]
    diag = torch.sqrt(torch.pow(ln, 2) + torch.pow(wd, 2))
    return torch.stack(((gt_rois[:, 0] - cx) / diag, (gt_rois[:, 1] - cy) / diag, (gt_rois[:, 2] - cz) / ht,
                        torch.log(gt_rois[:, 3] / ln), torch.log(gt_rois[:, 4] / wd), torch.log(gt_rois[:, 5] / ht),
                        gt_rois[:, 6]), 1)


def lidar_3d_bbox_transform_inv(rois, boxes, deltas, scales=None):
    """bbox_transform.py:174-233.  rois (N,4) axis-aligned, boxes (N,7) 3-D anchors, deltas (N,7K)."""
    if scales is not None:
        rois = rois / scales          # the anchors' x,y,l,w are scaled too (:177-178) but never read afterwards
    if len(boxes) == 0:
        return deltas.detach() * 0
    ln = rois[:, 2] - rois[:, 0] + 1
    wd = rois[:, 3] - rois[:, 1] + 1
    ht = boxes[:, 5]
    cx, cy, cz = rois[:, 0] + ln / 2.0, rois[:, 1] + wd / 2.0, boxes[:, 2]
    diag = torch.sqrt(torch.pow(ln, 2) + torch.pow(wd, 2))
    d = [deltas[:, i::7] for i in range(7)]
    parts = [d[0] * diag.unsqueeze(1) + cx.unsqueeze(1), d[1] * diag.unsqueeze(1) + cy.unsqueeze(1),
             d[2] * ht.unsqueeze(1) + cz.unsqueeze(1), torch.exp(d[3]) * ln.unsqueeze(1),
             torch.exp(d[4]) * wd.unsqueeze(1), torch.exp(d[5]) * ht.unsqueeze(1), d[6]]
    return torch.cat([p.unsqueeze(2) for p in parts], 2).view(len(boxes), -1)


def uncertainty_transform_inv(boxes, deltas, uncertainty, scales=None):
    """bbox_transform.py:107-130 with the per-box scaling the function means (upstream multiplies (N,K) by (N,) without
    unsqueeze(1), which is per-box only for N == 1; pinned one box per call by tests/golden/uc_inv.npz)."""
    if scales is not None:
        boxes = boxes / scales
    ln = (boxes[:, 2] - boxes[:, 0] + 1).unsqueeze(1)
    wd = (boxes[:, 3] - boxes[:, 1] + 1).unsqueeze(1)
    parts = [uncertainty[:, 0::7] * ln, uncertainty[:, 1::7] * wd, torch.exp(uncertainty[:, 3::7]) - 1,
             torch.exp(uncertainty[:, 4::7]) - 1]
    return torch.pow(torch.cat([p.unsqueeze(2) for p in parts], 2).view(len(boxes), -1), 2)


def lidar_3d_uncertainty_transform_inv(rois, boxes, deltas, uncertainty, scales=None):
    """bbox_transform.py:132-169."""
    if scales is not None:
        rois = rois / scales
    ln = (rois[:, 2] - rois[:, 0] + 1).unsqueeze(1)
    wd = (rois[:, 3] - rois[:, 1] + 1).unsqueeze(1)
    ht = boxes[:, 5].unsqueeze(1)
    u = [uncertainty[:, i::7] for i in range(7)]
    parts = [u[0] * ln, u[1] * wd, u[2] * ht, torch.exp(u[3]) - 1, torch.exp(u[4]) - 1, torch.exp(u[5]) - 1, u[6]]
    return torch.pow(torch.cat([p.unsqueeze(2) for p in parts], 2).view(len(boxes), -1), 2)


def lidar_extents():
    return [LIDAR_X_RANGE[0], LIDAR_Y_RANGE[0], LIDAR_Z_RANGE[0], LIDAR_X_RANGE[1], LIDAR_Y_RANGE[1], LIDAR_Z_RANGE[1]]


def bbox_voxel_grid_to_pc(bboxes, bev_extents, info):
    """utils/bbox.py:140-162 (aabb=False): voxel-grid [xc,yc,zc,l,w,h,ry] -> metres; numpy in place."""
    scale = info[6]
    s_info = np.asarray(info[0:6]) * 1 / scale
    kx = (bev_extents[3] - bev_extents[0]) / (s_info[1] - s_info[0])
    ky = (bev_extents[4] - bev_extents[1]) / (s_info[3] - s_info[2])
    bboxes[:, 0] = bboxes[:, 0] * kx + bev_extents[0]
    bboxes[:, 1] = bboxes[:, 1] * ky + bev_extents[1]
    bboxes[:, 3] = bboxes[:, 3] * kx
    bboxes[:, 4] = bboxes[:, 4] * ky
    return bboxes


def proposal_top_layer(rpn_cls_prob, rpn_bbox_pred, info, anchors, num_anchors, rpn_top_n=5000, rng=None):
    """proposal_top_layer.py:18-59: top-N by fg score, no NMS; random choice WITH replacement when fewer
    scores than N (:32-37, numpy RNG)."""
    scores = rpn_cls_prob[:, :, :, num_anchors:].contiguous().view(-1, 1)
    deltas = rpn_bbox_pred.reshape(-1, 4)
    length = scores.size(0)
    if length < rpn_top_n:
        rng = rng or np.random
        top = torch.from_numpy(rng.choice(length, size=rpn_top_n, replace=True)).long()
    else:
        top = stable_desc_order(scores)[:rpn_top_n]
    anchors, deltas, scores = anchors[top, :], deltas[top, :], scores[top]
    proposals = clip_boxes(bbox_transform_inv(anchors, deltas), info)
    return torch.cat((proposals.new_zeros(proposals.size(0), 1), proposals), 1), scores, anchors


# ==============================================================================================
# Training targets and losses
# ==============================================================================================
TRAIN_RPN_POSITIVE_OVERLAP, TRAIN_RPN_NEGATIVE_OVERLAP = 0.7, 0.3      # config.py:174-177
TRAIN_RPN_FG_FRACTION, TRAIN_RPN_BATCHSIZE = 0.5, 256                  # config.py:181-184
TRAIN_ROI_BATCH_SIZE, TRAIN_FG_FRACTION = 256, 0.25                    # config.py:123-126
TRAIN_FG_THRESH, TRAIN_BG_THRESH_HI, TRAIN_BG_THRESH_LO = 0.6, 0.5, 0.0  # config.py:129-134


def anchor_target_layer(gt_boxes, info, all_anchors, num_anchors, height, width, rpn_batchsize=TRAIN_RPN_BATCHSIZE,
                        fg_fraction=TRAIN_RPN_FG_FRACTION, generator=None):
    """anchor_target_layer.py:22-165 (torch variant, IGNORE_DC off, CLOBBER off, uniform weights)."""
    total = all_anchors.shape[0]
    inside = torch.where((all_anchors[:, 0] >= info[0]) & (all_anchors[:, 1] >= info[2]) &
                         (all_anchors[:, 2] < info[1]) & (all_anchors[:, 3] < info[3]))[0]
    anchors = all_anchors[inside, :]
    labels = torch.full((len(inside),), -1, dtype=torch.int64)
    ov = bbox_overlaps(anchors.contiguous(), gt_boxes[:, :4].contiguous())
    argmax = ov.argmax(dim=1)
    max_ov = ov[torch.arange(len(inside)), argmax]
    gt_max = ov[ov.argmax(dim=0), torch.arange(ov.shape[1])]
    gt_max = torch.clamp(gt_max, torch.finfo(torch.float32).eps, float("inf"))
    gt_argmax = torch.where(ov == gt_max)[0]                     # every anchor tying a gt's best overlap
    labels[max_ov < TRAIN_RPN_NEGATIVE_OVERLAP] = 0
    labels[gt_argmax] = 1
    labels[max_ov >= TRAIN_RPN_POSITIVE_OVERLAP] = 1
    num_fg = int(fg_fraction * rpn_batchsize)
    fg = torch.where(labels == 1)[0]
    if len(fg) > num_fg:
        labels[fg[torch.randperm(fg.numel(), generator=generator)[num_fg:]]] = -1
    num_bg = rpn_batchsize - int(torch.sum(labels == 1))
    bg = torch.where(labels == 0)[0]
    if len(bg) > num_bg:
        labels[bg[torch.randperm(bg.numel(), generator=generator)[num_bg:]]] = -1
    targets = bbox_transform(anchors, gt_boxes[argmax, :4])
    inside_w = torch.zeros((len(inside), 4))
    inside_w[labels == 1, :] = 1.0                                # RPN_BBOX_INSIDE_WEIGHTS (1,1,1,1)
    outside_w = torch.zeros((len(inside), 4))
    w = 1.0 / float(torch.sum(labels >= 0))
    outside_w[labels == 1, :] = w
    outside_w[labels == 0, :] = w

    def unmap(data, fill):
        shape = (total,) + tuple(data.shape[1:])
        ret = torch.full(shape, float(fill), dtype=torch.float32)
        ret[inside] = data.float()
        return ret

    a = num_anchors
    return (unmap(labels, -1).reshape(1, height, width, a).permute(0, 3, 1, 2),
            unmap(targets, 0).reshape(1, height, width, a * 4), unmap(inside_w, 0).reshape(1, height, width, a * 4),
            unmap(outside_w, 0).reshape(1, height, width, a * 4))


def _choice(n, k, replace, generator):
    """proposal_target_layer.py:265-284."""
    if replace:
        return torch.randint(n, (k,), generator=generator)
    if k > n:
        idx = torch.arange(n).repeat(math.ceil(k / n))
        return idx[torch.randperm(idx.shape[0], generator=generator)][:k]
    return torch.randperm(n, generator=generator)[:k]


def proposal_target_layer(rpn_rois, rpn_scores, anchors_3d, gt_boxes, true_gt_boxes, num_classes, num_bbox_elem,
                          net_type="image", generator=None):
    """proposal_target_layer.py:22-262 (USE_GT off, IGNORE_DC off, targets normalised by the configured
    means/stds :144-147,160-163).  Returns labels (R,1), rois (R,5), anchors_3d, roi_scores, targets, inside, outside."""
    rois_per_frame = TRAIN_ROI_BATCH_SIZE
    fg_quota = int(round(TRAIN_FG_FRACTION * rois_per_frame))
    ov = bbox_overlaps(rpn_rois[:, 1:5], gt_boxes[:, :4])
    max_ov, assign = ov.max(1)
    labels = gt_boxes[assign, 4]
    fg = (max_ov >= TRAIN_FG_THRESH).nonzero().view(-1)
    bg = ((max_ov < TRAIN_BG_THRESH_HI) & (max_ov >= TRAIN_BG_THRESH_LO)).nonzero().view(-1)
    if fg.numel() > 0 and bg.numel() > 0:
        n_fg = min(fg_quota, fg.numel())
        fg = fg[_choice(fg.numel(), int(n_fg), False, generator)]
        n_bg = rois_per_frame - n_fg
        bg = bg[_choice(bg.numel(), int(n_bg), bg.numel() < n_bg, generator)]
    elif fg.numel() > 0:
        fg = fg[_choice(fg.numel(), int(rois_per_frame), fg.numel() < rois_per_frame, generator)]
        n_fg = rois_per_frame
    elif bg.numel() > 0:
        bg = bg[_choice(bg.numel(), int(rois_per_frame), bg.numel() < rois_per_frame, generator)]
        n_fg = 0
    else:
        raise RuntimeError("no foreground and no background RoI (the reference drops into pdb here)")
    keep = torch.cat([fg, bg], 0)
    labels = labels[keep].contiguous()
    labels[int(n_fg):] = 0
    rois, scores, a3 = rpn_rois[keep].contiguous(), rpn_scores[keep].contiguous(), anchors_3d[keep].contiguous()
    if net_type == "lidar":
        t = lidar_3d_bbox_transform(rois[:, 1:5], a3, true_gt_boxes[assign[keep]][:, :-1])
        t = (t - t.new_tensor(LIDAR_BBOX_NORMALIZE_MEANS)) / t.new_tensor(LIDAR_BBOX_NORMALIZE_STDS)
    else:
        t = bbox_transform(rois[:, 1:5], gt_boxes[assign[keep]][:, :4])
        t = (t - t.new_tensor(BBOX_NORMALIZE_MEANS)) / t.new_tensor(BBOX_NORMALIZE_STDS)
    e = num_bbox_elem
    targets = labels.new_zeros(labels.numel(), e * num_classes)
    inside = labels.new_zeros(targets.shape)
    for i in (labels > 0).nonzero().view(-1).tolist():
        c = int(labels[i])
        targets[i, e * c:e * c + e] = t[i]
        inside[i, e * c:e * c + e] = 1.0
    return labels.view(-1, 1), rois.view(-1, 5), a3, scores.view(-1), targets, inside, (inside > 0).float()


def huber_loss(pred, targets, delta=1.0, sin_en=False):
    """loss_utils.py:28-37."""
    diff = pred - targets
    if sin_en:
        diff = torch.sin(diff)
    a = torch.abs(diff)
    small = (a < delta).detach().float()
    return 0.5 * torch.pow(diff, 2) * small + delta * (a - 0.5 * delta) * (1.0 - small)


def smooth_l1_loss(stage, bbox_pred, bbox_targets, inside_w, outside_w, dim=(1,), net_type="image", bbox_var=None):
    """loss_utils.py:39-101; ``bbox_var`` (predicted log-variance) switches the aleatoric branch on (:82-85, the
    cfg.UC.EN_BBOX_ALEATORIC / EN_RPN_BBOX_ALEATORIC case), default off like the reference's config."""
    pred = bbox_pred * inside_w
    tgt = bbox_targets * inside_w
    if net_type == "lidar" and stage == "DET":
        p7, t7 = pred.reshape(-1, 7), tgt.reshape(-1, 7)
        loss = torch.cat((huber_loss(p7[:, 0:6], t7[:, 0:6]), huber_loss(p7[:, 6:7], t7[:, 6:7], sin_en=True)), 1)
        loss = (loss * loss.new_tensor(LIDAR_REG_LOSS_WEIGHT)).reshape(pred.shape)
    else:
        loss = huber_loss(pred, tgt)
    if bbox_var is not None:
        loss = (0.5 * loss * torch.exp(-bbox_var) + 0.5 * bbox_var) * inside_w
    loss = outside_w * loss
    for i in sorted(dim, reverse=True):
        loss = loss.sum(i)
    return loss.mean()


def compute_bbox_var(bbox_samples):
    """loss_utils.py:114-120: unbiased variance over the T stochastic passes, clamped at 0."""
    n = bbox_samples.shape[0]
    mean_sq = torch.pow(torch.sum(bbox_samples, dim=0), 2)
    var = torch.sum(torch.pow(bbox_samples, 2), dim=0)
    var = var + (-mean_sq / n)
    return (var / (n - 1)).clamp_min(0.0)


def categorical_entropy(cls_prob):
    """loss_utils.py:122-130 (bits)."""
    return -torch.sum(cls_prob * torch.log2(cls_prob), dim=1)


def categorical_mutual_information(cls_score):
    """loss_utils.py:133-141: cls_score (T,N,C) logits -> H(mean softmax) - mean H(softmax)."""
    prob = F.softmax(cls_score, dim=2)
    total = categorical_entropy(torch.mean(prob, dim=0))
    return torch.mean(torch.sum(prob * torch.log2(prob), dim=2), dim=0) + total


# ----------------------------------------------------------------------------------------------
# Uncertainty heads (cfg.UC.*) — module names / sizes / rates lib/nets/imagenet.py:52-91, lidarnet.py:56-102; Monte-Carlo
# protocol lib/model/test.py:74-77; arithmetic lib/utils/loss_utils.py:114-169.  The wiring restates the RECONSTRUCTED
# contract of the missing network.py (constants mirrored from the product's nets/uncertainty.py).  The counter-based
# random draws of the device (csrc/rng.h) are replayed here so that stochastic outputs can be compared value by value.
# ----------------------------------------------------------------------------------------------
UC_STREAM = {'bbox_drop1': 11, 'bbox_drop2': 12, 'cls_drop1': 13, 'cls_drop2': 14, 'logit_distort': 15, 'bayes_ce': 16}
UNCERTAINTY_ORDER = ('a_entropy', 'a_mutual_info', 'a_cls_var', 'e_entropy', 'e_mutual_info', 'e_cls_var',
                     'a_bbox_var', 'e_bbox_var')


def _hash32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d); x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b); x ^= x >> np.uint32(16)
    return x


def rand_key(seed, stream, idx):
    """csrc/rng.h rand_key: hash32(hash32(seed ^ stream * 0x9e3779b9) + i * 0x85ebca6b), all mod 2^32."""
    with np.errstate(over="ignore"):
        base = _hash32(np.array([(int(seed) ^ ((int(stream) * 0x9e3779b9) & 0xFFFFFFFF)) & 0xFFFFFFFF], dtype=np.uint32))[0]
        i = np.asarray(idx, dtype=np.uint64)
        return _hash32(((np.uint64(base) + i * np.uint64(0x85ebca6b)) & np.uint64(0xFFFFFFFF)).astype(np.uint32))


def uniform01(seed, stream, idx):
    return ((rand_key(seed, stream, idx) >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def normal01(seed, stream, idx):
    u1, u2 = uniform01(seed, 2 * stream, idx), uniform01(seed, 2 * stream + 1, idx)
    return (np.sqrt(np.float32(-2.0) * np.log(u1)) * np.cos(np.float32(6.2831853071795864) * u2)).astype(np.float32)


def dropout_replay(x, p, seed, stream, repeat=1):
    """nn.Dropout(p) in train() mode with the device's masks: x (n...) -> (repeat, n...) (squeezed for repeat == 1)."""
    n = x.numel()
    keep = torch.from_numpy(uniform01(seed, stream, np.arange(n * repeat)) >= np.float32(p)).view((repeat,) + tuple(x.shape))
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(p))
    y = torch.where(keep, x.unsqueeze(0) * float(scale), torch.zeros((), dtype=x.dtype))
    return y[0] if repeat == 1 else y


def logit_distort_replay(score, var, num_sample, seed, stream=UC_STREAM['logit_distort']):
    """logit_distort (loss_utils.py:143-147) with the device's normal draws: (N,K) -> (S,N,K)."""
    n = score.numel()
    eps = torch.from_numpy(normal01(seed, stream, np.arange(n * num_sample))).view((num_sample,) + tuple(score.shape))
    return score.unsqueeze(0) + torch.sqrt(var).unsqueeze(0) * eps.to(score.dtype), eps


def bayesian_cross_entropy(cls_score, cls_var, targets, num_sample, seed, stream=UC_STREAM['bayes_ce']):
    """loss_utils.py:149-169 (differentiable; the draws replay the device's)."""
    samples, _ = logit_distort_replay(cls_score, cls_var, num_sample, seed, stream)
    avg = torch.mean(F.softmax(samples, dim=2), dim=0)
    sel = -torch.log(avg).gather(1, targets.long().unsqueeze(1))
    return torch.mean(sel), categorical_mutual_information(samples)


class UcHeadsOracle(nn.Module):
    """The detection heads with cfg.UC.* flags on (R, fc7) features.  flags: dict of the four EN_* booleans."""

    def __init__(self, flags, num_classes=2, bbox_elem=4, fc7=2048, lidar=False):
        super().__init__()
        self.flags, self.k, self.e, self.lidar = dict(flags), num_classes, bbox_elem, lidar
        epi = flags.get('EN_BBOX_EPISTEMIC') or flags.get('EN_CLS_EPISTEMIC')
        d = fc7 // 4 if epi else fc7                                   # imagenet.py:52-53
        h = fc7 // 2                                                    # reconstruction constant UC_FC1_DIVISOR
        self.rates = {'cls': 0.2 if lidar else 0.3, 'bbox': 0.5 if lidar else 0.1}   # imagenet.py:55-56, lidarnet.py:60-61
        for prefix, key in (('bbox', 'EN_BBOX_EPISTEMIC'), ('cls', 'EN_CLS_EPISTEMIC')):
            if flags.get(key):
                setattr(self, prefix + '_fc1', nn.Linear(fc7, h))
                setattr(self, prefix + '_fc2', nn.Linear(h, d))
                if lidar:
                    setattr(self, prefix + '_bn1', nn.BatchNorm1d(h))
                    setattr(self, prefix + '_bn2', nn.BatchNorm1d(d))
        self.cls_score_net = nn.Linear(d, num_classes)
        self.bbox_pred_net = nn.Linear(d, num_classes * bbox_elem)
        if flags.get('EN_BBOX_ALEATORIC'):
            self.bbox_al_var_net = nn.Linear(d, num_classes * bbox_elem)
        if flags.get('EN_CLS_ALEATORIC'):
            self.cls_al_var_net = nn.Linear(d, num_classes)
        self.eval()

    def _branch(self, prefix, fc7, on, t, seed):
        if not on:
            return fc7, 1
        fc1, fc2 = getattr(self, prefix + '_fc1'), getattr(self, prefix + '_fc2')
        bn1, bn2 = getattr(self, prefix + '_bn1', None), getattr(self, prefix + '_bn2', None)
        p = self.rates[prefix]
        h = fc1(fc7)
        h = F.relu(bn1(h) if bn1 is not None else h)
        h = dropout_replay(h, p, seed, UC_STREAM[prefix + '_drop1'], repeat=t).reshape(t * fc7.shape[0], -1)
        h = fc2(h)
        h = F.relu(bn2(h) if bn2 is not None else h)
        return dropout_replay(h, p, seed, UC_STREAM[prefix + '_drop2']), t

    @torch.no_grad()
    def test(self, fc7, e_num_sample, a_num_ce_sample, seed, stds, means):
        f, r, k, e = self.flags, fc7.shape[0], self.k, self.e
        unc = {}
        feat_c, tc = self._branch('cls', fc7, f.get('EN_CLS_EPISTEMIC'), e_num_sample, seed)
        score_s = self.cls_score_net(feat_c).view(tc, r, k)
        prob_s = F.softmax(score_s, dim=2)
        cls_prob, cls_score = prob_s.mean(0), score_s.mean(0)
        if f.get('EN_CLS_ALEATORIC'):
            logvar = self.cls_al_var_net(feat_c).view(tc, r, k).mean(0)
            var = torch.exp(logvar)
            dist, _ = logit_distort_replay(cls_score, var, a_num_ce_sample, seed)
            unc['a_entropy'] = categorical_entropy(F.softmax(dist, dim=2).mean(0))
            unc['a_mutual_info'] = categorical_mutual_information(dist)
            unc['a_cls_var'] = var
        if f.get('EN_CLS_EPISTEMIC'):
            unc['e_entropy'] = categorical_entropy(cls_prob)
            unc['e_mutual_info'] = categorical_mutual_information(score_s)
            unc['e_cls_var'] = compute_bbox_var(prob_s) if tc > 1 else torch.zeros_like(cls_prob)
        feat_b, tb = self._branch('bbox', fc7, f.get('EN_BBOX_EPISTEMIC'), e_num_sample, seed)
        box_s = self.bbox_pred_net(feat_b).view(tb, r, k * e)
        bbox_pred = box_s.mean(0)
        stds_t, means_t = torch.tensor(stds).repeat(k), torch.tensor(means).repeat(k)
        if f.get('EN_BBOX_ALEATORIC'):
            # both box variances in de-normalised delta units (BBOX_VAR_ON_DENORMALISED of the build, a named choice)
            unc['a_bbox_var'] = torch.exp(self.bbox_al_var_net(feat_b).view(tb, r, k * e).mean(0)) * stds_t * stds_t
        if f.get('EN_BBOX_EPISTEMIC'):
            unc['e_bbox_var'] = (compute_bbox_var(box_s) if tb > 1 else torch.zeros_like(bbox_pred)) * stds_t * stds_t
        return cls_score, cls_prob, bbox_pred, bbox_pred * stds_t + means_t, {n: unc[n] for n in UNCERTAINTY_ORDER if n in unc}


# ----------------------------------------------------------------------------------------------
# LiDAR-BEV detector — lib/nets/lidarnet.py:28-151 on the RECONSTRUCTED Network contract: 15-plane stem,
# layer4 without BatchNorm (non-FPN), 3-D anchors + their BEV rectangles for the RPN, 7-DoF decode.
# ----------------------------------------------------------------------------------------------
class LidarNetOracle(ImageNetOracle):
    def __init__(self, num_classes=2):
        nn.Module.__init__(self)
        self._num_classes = num_classes
        self._num_anchors = len(LIDAR_ANCHOR_SCALES) * len(LIDAR_ANCHOR_ANGLES)
        self._feat_stride = 16                                     # lidarnet.py:48
        self.resnet = ResNet101(in_channels=LIDAR_NUM_CHANNEL, batchnorm_en=False)   # lidarnet.py:52,107
        a = self._num_anchors
        self.rpn_net = nn.Conv2d(1024, RPN_CHANNELS, 3, padding=1)
        self.rpn_cls_score_net = nn.Conv2d(RPN_CHANNELS, 2 * a, 1)
        self.rpn_bbox_pred_net = nn.Conv2d(RPN_CHANNELS, 4 * a, 1)
        self.cls_score_net = nn.Linear(2048, num_classes)
        self.bbox_pred_net = nn.Linear(2048, num_classes * LIDAR_NUM_BBOX_ELEM)
        self.eval()

    def _region_proposal(self, net_conv, info, structured=None):
        a = self._num_anchors
        h, w = net_conv.shape[2], net_conv.shape[3]
        _, a3 = generate_anchors_3d(h, w, self._feat_stride, frame_scale=float(info[6]))
        anchors = torch.from_numpy(bbaa_graphics_gems(a3))
        rpn = F.relu(self.rpn_net(net_conv))
        cls_score = self.rpn_cls_score_net(rpn)
        bbox_pred = self.rpn_bbox_pred_net(rpn).permute(0, 2, 3, 1).contiguous()
        if structured is not None:
            cls_score, bbox_pred = structured
        prob = F.softmax(cls_score.view(1, 2, a * h, w), dim=1).view(1, 2 * a, h, w).permute(0, 2, 3, 1).contiguous()
        rois, scores, dbg = proposal_layer(prob, bbox_pred, info, anchors, a, return_debug=True)
        roi_a3 = torch.from_numpy(a3)[dbg["order"]][dbg["keep"]]   # proposal_layer.py:44,52
        self._dbg = {"anchors": anchors, "anchors_3d": torch.from_numpy(a3), "roi_anchors_3d": roi_a3,
                     "rpn_cls_prob": prob, "rpn_bbox_pred": bbox_pred, "rpn_cls_score": cls_score, **dbg}
        return rois, scores

    @torch.no_grad()
    def test_frame(self, data, info, structured=None):
        image = torch.from_numpy(np.ascontiguousarray(data)).permute(0, 3, 1, 2).contiguous()
        net_conv = self._image_to_head(image)
        rois, _ = self._region_proposal(net_conv, info, structured)
        pool5 = self._crop_pool_layer(net_conv, rois)
        fc7 = self._head_to_tail(pool5)
        cls_score, cls_prob, bbox_pred = self._region_classification(fc7)
        stds = torch.tensor(LIDAR_BBOX_NORMALIZE_STDS).repeat(self._num_classes).unsqueeze(0)
        means = torch.tensor(LIDAR_BBOX_NORMALIZE_MEANS).repeat(self._num_classes).unsqueeze(0)
        deltas = bbox_pred.mul(stds).add(means)
        pred_boxes = lidar_3d_bbox_transform_inv(rois[:, 1:5], self._dbg["roi_anchors_3d"], deltas, float(info[6]))
        self._dbg.update({"net_conv": net_conv, "pool5": pool5, "fc7": fc7, "bbox_pred": bbox_pred})
        return cls_score, cls_prob, pred_boxes, rois, {}


# ----------------------------------------------------------------------------------------------
# BEV input producer — lib/roi_data_layer/minibatch.py:232-235,434-512.  spconv.utils.VoxelGeneratorV2 is a
# third-party dependency (spconv 1.x, req.txt) that is not vendored: PARITY UNPINNED; its points_to_voxel loop is
# restated from the published algorithm (voxels numbered by first appearance, `continue` once max_voxels exist,
# first max_points points per voxel, fp32 arithmetic).  The numpy scatter afterwards follows the reference literally,
# including "the voxel created last in an (x, y) column wins" for the meta channels.
# ----------------------------------------------------------------------------------------------
LIDAR_NUM_SLICES, LIDAR_NUM_META_CHANNEL = 12, 3                           # config.py:402-403
LIDAR_MAX_PTS_PER_VOXEL, LIDAR_MAX_NUM_VOXEL = 32, 25000                    # config.py:405-406


def points_to_voxel(points, voxel_size, coors_range, max_points, max_voxels):
    """spconv.utils.points_to_voxel (reverse_index=True): returns voxels (V, max_points, F), coordinates (V, 3) in
    zyx order, num_points_per_voxel (V,)."""
    points = np.asarray(points, dtype=np.float32)
    voxel_size = np.asarray(voxel_size, dtype=np.float32)
    coors_range = np.asarray(coors_range, dtype=np.float32)
    grid = np.round((coors_range[3:] - coors_range[:3]) / voxel_size).astype(np.int64)
    lookup = -np.ones(tuple(grid[::-1]), dtype=np.int32)
    voxels = np.zeros((max_voxels, max_points, points.shape[1]), dtype=np.float32)
    coors = np.zeros((max_voxels, 3), dtype=np.int32)
    num = np.zeros((max_voxels,), dtype=np.int32)
    cells = np.floor((points[:, :3] - coors_range[:3]) / voxel_size)      # float32, like the numba loop
    ok = ((cells >= 0) & (cells < grid.astype(np.float32))).all(1)
    cells = cells.astype(np.int64)
    voxel_num = 0
    for i in np.nonzero(ok)[0]:
        cz, cy, cx = cells[i, 2], cells[i, 1], cells[i, 0]
        v = lookup[cz, cy, cx]
        if v == -1:
            if voxel_num >= max_voxels:
                continue
            v = voxel_num
            voxel_num += 1
            lookup[cz, cy, cx] = v
            coors[v] = (cz, cy, cx)
        if num[v] < max_points:
            voxels[v, num[v]] = points[i]
            num[v] += 1
    return voxels[:voxel_num], coors[:voxel_num], num[:voxel_num]


def get_lidar_blob(points, scale, elongation=False, max_points=LIDAR_MAX_PTS_PER_VOXEL, max_voxels=LIDAR_MAX_NUM_VOXEL):
    """minibatch.py:232-235 (filter_points) + :434-512 for one frame; returns (info (7,), blob (1, Y, X, 15))."""
    pc = np.asarray(points, dtype=np.float32).copy()
    pc = pc[(pc[:, 0] >= LIDAR_X_RANGE[0]) & (pc[:, 1] >= LIDAR_Y_RANGE[0]) & (pc[:, 2] >= LIDAR_Z_RANGE[0])]
    pc = pc[(pc[:, 0] < LIDAR_X_RANGE[1]) & (pc[:, 1] < LIDAR_Y_RANGE[1]) & (pc[:, 2] < LIDAR_Z_RANGE[1])]
    voxel_len = LIDAR_VOXEL_LEN / scale
    nx = int((LIDAR_X_RANGE[1] - LIDAR_X_RANGE[0]) * (1 / voxel_len))
    ny = int((LIDAR_Y_RANGE[1] - LIDAR_Y_RANGE[0]) * (1 / voxel_len))
    info = np.array([0, nx, 0, ny, 0, LIDAR_NUM_SLICES, scale], dtype=np.float32)
    extents = [LIDAR_X_RANGE[0], LIDAR_Y_RANGE[0], 0, LIDAR_X_RANGE[1], LIDAR_Y_RANGE[1], LIDAR_Z_RANGE[1] - LIDAR_Z_RANGE[0]]
    pc[:, 2] -= LIDAR_Z_RANGE[0]
    voxels, coords, npts = points_to_voxel(pc, [voxel_len, voxel_len, LIDAR_VOXEL_HEIGHT], extents, max_points, max_voxels)
    bev = np.zeros((nx, ny, LIDAR_NUM_CHANNEL), dtype=np.float32)
    coords = coords.copy()
    coords[:, [2, 1, 0]] = coords[:, [0, 1, 2]]                               # zyx -> xyz (:463)
    max_height = np.amax(voxels[:, :, 2], axis=1) - coords[:, 2] * LIDAR_VOXEL_HEIGHT
    bev[tuple(zip(*coords))] = max_height
    xy = coords[:, 0:2]
    density = npts / max_points
    bev[tuple(zip(*np.hstack((xy, np.full((xy.shape[0], 1), LIDAR_NUM_SLICES)))))] = density
    intensity = np.sum(voxels[:, :, 3], axis=1) / npts
    bev[tuple(zip(*np.hstack((xy, np.full((xy.shape[0], 1), LIDAR_NUM_SLICES + 1)))))] = np.tanh(intensity)
    elong = np.sum(voxels[:, :, 4], axis=1) / npts if elongation else np.zeros((voxels.shape[0],))
    bev[tuple(zip(*np.hstack((xy, np.full((xy.shape[0], 1), LIDAR_NUM_SLICES + 2)))))] = np.tanh(elong)
    return info, np.transpose(bev, axes=[1, 0, 2])[None]


def _lidar_set_trainable(self, fixed_blocks=1):
    """lidarnet.py:104-134: every BatchNorm trainable (set_bn_var) except in the frozen blocks; the stem is frozen
    for FIXED_BLOCKS >= 0, layerN for N <= FIXED_BLOCKS."""
    for m in self.resnet.modules():
        if isinstance(m, nn.BatchNorm2d):
            for p in m.parameters():
                p.requires_grad = True
    frozen = [self.resnet.conv1, self.resnet.bn1] + [getattr(self.resnet, "layer%d" % n) for n in (1, 2, 3, 4)
                                                     if fixed_blocks >= n]
    for m in frozen:
        for p in m.parameters():
            p.requires_grad = False


def _lidar_train_mode(self, fixed_blocks=1):
    """lidarnet.py:152-175: the network in train() mode, every backbone BatchNorm in eval() mode except those of
    layerN, N > FIXED_BLOCKS, which use batch statistics."""
    self.train()
    for m in self.resnet.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eval()
    for n in (1, 2, 3):
        if fixed_blocks <= n:
            for m in getattr(self.resnet, "layer%d" % (n + 1)).modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.train()


def _lidar_train_forward(self, data, info, true_gt_boxes, generator=None, pre_nms=12000, post_nms=2000, proposals=None):
    """One TRAIN forward of the LiDAR detector (RECONSTRUCTED contract, like the image forms above): the RPN and
    anchor targets work on the BEV rectangles of the 3-D anchors / gt boxes, the second stage on the 7-of-7K targets
    (proposal_target_layer.py:142-154) with the sin(ry) Huber term (loss_utils.py:61-77).
    ``proposals=(rois (N,5), scores (N,1), anchors_3d (N,7))`` replaces the proposal_layer output."""
    image = torch.from_numpy(np.ascontiguousarray(data)).permute(0, 3, 1, 2).contiguous()
    tgt3 = torch.as_tensor(true_gt_boxes, dtype=torch.float32)
    gt = torch.cat((torch.from_numpy(bbaa_graphics_gems(tgt3[:, :7].numpy())), tgt3[:, 7:8]), 1)
    net_conv = self._image_to_head(image)
    a, (h, w) = self._num_anchors, net_conv.shape[2:]
    _, a3 = generate_anchors_3d(h, w, self._feat_stride, frame_scale=float(info[6]))
    anchors = torch.from_numpy(bbaa_graphics_gems(a3))
    rpn = F.relu(self.rpn_net(net_conv))
    cls_score = self.rpn_cls_score_net(rpn)
    bbox_pred = self.rpn_bbox_pred_net(rpn).permute(0, 2, 3, 1).contiguous()
    with torch.no_grad():
        prob = F.softmax(cls_score.view(1, 2, a * h, w), dim=1).view(1, 2 * a, h, w).permute(0, 2, 3, 1).contiguous()
        rois, scores, dbg = proposal_layer(prob, bbox_pred, info, anchors, a, pre_nms, post_nms, TEST_RPN_NMS_THRESH,
                                           return_debug=True)
        roi_a3 = torch.from_numpy(a3)[dbg["order"]][dbg["keep"]]
        if proposals is not None:
            rois, scores, roi_a3 = proposals
        lab, tgt, inw, outw = anchor_target_layer(gt, info, anchors, a, h, w, generator=generator)
        pl, prois, pa3, _, ptgt, pin, pout = proposal_target_layer(rois, scores, roi_a3, gt, tgt3, self._num_classes, 7,
                                                                   net_type="lidar", generator=generator)
    logits = torch.stack((cls_score[0, :a].permute(1, 2, 0).reshape(-1), cls_score[0, a:].permute(1, 2, 0).reshape(-1)), 1)
    labels_hwa = lab[0].permute(1, 2, 0).reshape(-1)
    sel = labels_hwa >= 0
    rpn_ce = F.cross_entropy(logits[sel], labels_hwa[sel].long())
    rpn_box = smooth_l1_loss("RPN", bbox_pred, tgt, inw, outw, dim=(1, 2, 3))
    pool5 = roi_align_torch(net_conv, prois, POOLING_SIZE, 1.0 / self._feat_stride)
    fc7 = self.resnet.layer4(pool5).mean(3).mean(2)
    det_cls, det_box = self.cls_score_net(fc7), self.bbox_pred_net(fc7)
    ce = F.cross_entropy(det_cls, pl.view(-1).long())
    box = smooth_l1_loss("DET", det_box, ptgt, pin, pout, net_type="lidar")
    losses = {"rpn_cross_entropy": rpn_ce, "rpn_loss_box": rpn_box, "cross_entropy": ce, "loss_box": box,
              "total_loss": rpn_ce + rpn_box + ce + box}
    dbg = {"anchor_labels": labels_hwa, "anchor_targets": tgt.reshape(-1, 4), "anchor_inside": inw.reshape(-1, 4),
           "anchor_outside": outw.reshape(-1, 4), "rois": prois, "anchors_3d": pa3, "labels": pl.view(-1), "targets": ptgt,
           "inside": pin, "outside": pout, "net_conv": net_conv, "fc7": fc7, "cls_score": det_cls, "gt_aabb": gt}
    return losses, dbg


LidarNetOracle.train_forward = _lidar_train_forward
LidarNetOracle.set_trainable = _lidar_set_trainable
LidarNetOracle.train_mode = _lidar_train_mode


def filter_and_draw_prep_lidar(rois, cls_prob, pred_boxes, num_classes, thresh=0.1, nms_thresh=TEST_NMS_THRESH):
    """filter_predictions.py:45-72,92-93 with db_type 'lidar': no clamp, NMS on xc -+ l/2, yc -+ w/2, dets rows
    [xc,yc,zc,l,w,h,ry,score]."""
    all_boxes = [np.empty((0, 8), dtype=np.float32) for _ in range(num_classes)]
    for j in range(1, num_classes):
        inds = torch.where(cls_prob[:, j] > thresh)[0]
        if inds.numel() == 0:
            continue
        cs = cls_prob[inds, j]
        cb = pred_boxes[inds, j * 7:(j + 1) * 7]
        aabb = torch.cat((cb[:, 0:1] - cb[:, 3:4] / 2.0, cb[:, 1:2] - cb[:, 4:5] / 2.0,
                          cb[:, 0:1] + cb[:, 3:4] / 2.0, cb[:, 1:2] + cb[:, 4:5] / 2.0), dim=1)
        dets = np.hstack((cb.numpy(), cs.unsqueeze(1).numpy())).astype(np.float32, copy=False)
        all_boxes[j] = dets[nms(aabb, cs, nms_thresh).numpy(), :]
    return rois[:, 1:5].numpy(), all_boxes


def frame_detect_lidar(net, data, info, num_classes, thresh=0.5, max_dets=100, structured=None):
    """lib/model/test.py:68-93,210-228 for one LiDAR frame: detections in METRES (bbox_voxel_grid_to_pc)."""
    _, probs, boxes, rois, _ = net.test_frame(data, info, structured)
    _, all_boxes = filter_and_draw_prep_lidar(rois, probs, boxes, num_classes, thresh)
    out = []
    for b in all_boxes:
        b = max_dets_cut(b, max_dets)
        out.append(bbox_voxel_grid_to_pc(b.copy(), lidar_extents(), info) if len(b) else b)
    return out


# ==============================================================================================
# FPN image detector + one training step (BASELINE config 4).  RECONSTRUCTED where the missing network.py
# is silent: RPN on p2 only (_feat_stride = 4, imagenet.py:34), MultiScaleRoIAlign over p2..p5, ReLU MLP tail
# t_fc1..3 (names imagenet.py:70-73), total loss = sum of the four terms.
# ==============================================================================================
class FPN(nn.Module):
    """lib/nets/fpn.py:23-68."""

    def __init__(self, planes=256):
        super().__init__()
        self.latlayer2 = nn.Conv2d(256, planes, 1)
        self.latlayer3 = nn.Conv2d(512, planes, 1)
        self.latlayer4 = nn.Conv2d(1024, planes, 1)
        self.latlayer5 = nn.Conv2d(2048, planes, 1)
        self.aalayer2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.aalayer3 = nn.Conv2d(planes, planes, 3, padding=1)
        self.aalayer4 = nn.Conv2d(planes, planes, 3, padding=1)     # defined, never used (fpn.py:39)

    @staticmethod
    def _upsample_add(x, y):
        return F.interpolate(x, size=y.shape[-2:], mode="bilinear", align_corners=False) + y

    def forward(self, c2, c3, c4, c5):
        p5 = self.latlayer5(c5)
        p4 = self._upsample_add(p5, self.latlayer4(c4))
        p3 = self.aalayer3(self._upsample_add(p4, self.latlayer3(c3)))
        p2 = self.aalayer2(self._upsample_add(p3, self.latlayer2(c2)))
        return p2, p3, p4, p5


def fpn_level_map(rois, k_min=2, k_max=5, s0=224.0, lvl0=4.0, eps=1e-6):
    """LevelMapper.__call__ (lib/utils/torchpoolers.py:39-51); area without +1."""
    area = (rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])
    lvl = torch.floor(lvl0 + torch.log2(torch.sqrt(area) / s0) + torch.tensor(eps, dtype=area.dtype))
    return (torch.clamp(lvl, min=k_min, max=k_max).to(torch.int64) - k_min).to(torch.int64)


def roi_align_torch(feat, rois, pooled, spatial_scale, sampling_ratio=ROI_ALIGN_SAMPLING_RATIO):
    """Differentiable restatement of roi_align() above built from torch gathers (used for gradient parity).
    feat (1,C,H,W), rois (R,5) -> (R,C,P,P)."""
    _, c, hgt, wid = feat.shape
    fm = feat[0].reshape(c, -1)
    outs = []
    pidx = torch.arange(pooled, dtype=torch.float32)
    for r in range(rois.shape[0]):
        sw, sh, ew, eh = [float(v) for v in (rois[r, 1:5].detach() * spatial_scale)]
        sw, sh, ew, eh = np.float32(sw), np.float32(sh), np.float32(ew), np.float32(eh)
        rw, rh = max(ew - sw, np.float32(1.0)), max(eh - sh, np.float32(1.0))
        bw, bh = np.float32(rw / np.float32(pooled)), np.float32(rh / np.float32(pooled))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(bh))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(bw))
        iy = torch.arange(gh, dtype=torch.float32)
        ix = torch.arange(gw, dtype=torch.float32)
        y = (float(sh) + pidx[:, None] * float(bh)) + (iy[None, :] + 0.5) * float(bh) / float(gh)     # (P, gh)
        x = (float(sw) + pidx[:, None] * float(bw)) + (ix[None, :] + 0.5) * float(bw) / float(gw)     # (P, gw)
        yy = y.reshape(-1)[:, None].expand(-1, pooled * gw)                                            # (P*gh, P*gw)
        xx = x.reshape(-1)[None, :].expand(pooled * gh, -1)
        empty = (yy < -1.0) | (yy > hgt) | (xx < -1.0) | (xx > wid)
        yy, xx = yy.clamp(min=0), xx.clamp(min=0)
        yl, xl = yy.floor().long(), xx.floor().long()
        ytop, xtop = yl >= hgt - 1, xl >= wid - 1
        yl = torch.where(ytop, torch.full_like(yl, hgt - 1), yl)
        xl = torch.where(xtop, torch.full_like(xl, wid - 1), xl)
        yh_ = torch.where(ytop, yl, yl + 1)
        xh_ = torch.where(xtop, xl, xl + 1)
        yy = torch.where(ytop, yl.float(), yy)
        xx = torch.where(xtop, xl.float(), xx)
        ly, lx = yy - yl.float(), xx - xl.float()
        hy, hx = 1.0 - ly, 1.0 - lx
        val = ((hy * hx) * fm[:, (yl * wid + xl)] + (hy * lx) * fm[:, (yl * wid + xh_)] +
               (ly * hx) * fm[:, (yh_ * wid + xl)] + (ly * lx) * fm[:, (yh_ * wid + xh_)])
        val = torch.where(empty[None], torch.zeros_like(val), val)
        outs.append(val.view(c, pooled, gh, pooled, gw).sum(dim=(2, 4)) / float(gh * gw))
    return torch.stack(outs, 0)


class FpnNetOracle(nn.Module):
    def __init__(self, num_classes=2, anchor_scales=ANCHOR_SCALES, anchor_ratios=ANCHOR_RATIOS):
        super().__init__()
        self._num_classes = num_classes
        self._anchor_scales, self._anchor_ratios = tuple(anchor_scales), tuple(anchor_ratios)
        self._num_anchors = len(anchor_scales) * len(anchor_ratios)
        self._feat_stride = 4                                      # imagenet.py:34 (multiscale pooling)
        self.resnet = ResNet101(use_fpn=True)
        self._fpn = FPN(256)
        a = self._num_anchors
        self.rpn_net = nn.Conv2d(256, RPN_CHANNELS, 3, padding=1)
        self.rpn_cls_score_net = nn.Conv2d(RPN_CHANNELS, 2 * a, 1)
        self.rpn_bbox_pred_net = nn.Conv2d(RPN_CHANNELS, 4 * a, 1)
        self.cls_score_net = nn.Linear(2048, num_classes)
        self.bbox_pred_net = nn.Linear(2048, num_classes * 4)
        self.t_fc1 = nn.Linear(POOLING_SIZE * POOLING_SIZE * 256, 2048)
        self.t_fc2 = nn.Linear(2048, 2048)
        self.t_fc3 = nn.Linear(2048, 2048)
        self.eval()                                                # frozen BN (imagenet.py:110-116,156-163)

    def set_trainable(self, fixed_blocks=1):
        """imagenet.py:96-116: stem and layerN (N <= FIXED_BLOCKS) frozen, every BatchNorm frozen."""
        frozen = [self.resnet.conv1, self.resnet.bn1] + [getattr(self.resnet, "layer%d" % n) for n in (1, 2, 3)
                                                         if fixed_blocks >= n]
        for m in frozen:
            for p in m.parameters():
                p.requires_grad = False
        for m in self.resnet.modules():
            if isinstance(m, nn.BatchNorm2d):
                for p in m.parameters():
                    p.requires_grad = False

    def pyramid(self, image):
        r = self.resnet
        c2 = r.layer1(r.stem(image))
        c3 = r.layer2(c2)
        c4 = r.layer3(c3)
        c5 = r.layer4(c4)
        return self._fpn(c2, c3, c4, c5)

    def pool(self, pyr, rois, image_hw):
        scales = [2.0 ** float(np.round(np.log2(float(f.shape[2]) / float(image_hw[0])))) for f in pyr]
        levels = fpn_level_map(rois)
        out = torch.zeros((rois.shape[0], pyr[0].shape[1], POOLING_SIZE, POOLING_SIZE))
        for lvl, (f, sc) in enumerate(zip(pyr, scales)):
            idx = (levels == lvl).nonzero().view(-1)
            if idx.numel():
                out = out.index_put((idx,), roi_align_torch(f, rois[idx], POOLING_SIZE, sc))
        return out, levels

    def tail(self, pool5):
        h = F.relu(self.t_fc1(pool5.reshape(pool5.shape[0], -1)))          # NCHW flattening (C,7,7)
        return F.relu(self.t_fc3(F.relu(self.t_fc2(h))))

    @torch.no_grad()
    def test_frame(self, data, info, structured=None):
        """TEST-mode frame of the FPN detector, the variant tools/test_net.py:196-197 switches on (cfg.USE_FPN for
        evaluation): pyramid (lib/nets/fpn.py:56-68) -> RPN on p2 -> proposal_layer with the TEST settings 6000 / 300 / 0.7 on
        all H/4 x W/4 x A anchors (lib/layer_utils/proposal_layer.py:18-57) -> LevelMapper + per-level RoIAlign
        (lib/utils/torchpoolers.py:137-200) -> t_fc1..3 -> heads -> de-normalised decode.  Same return as
        ImageNetOracle.test_frame; ``structured=(cls_score (1,2A,H,W), bbox_pred (1,H,W,4A))`` replaces the RPN head's output."""
        image = torch.from_numpy(np.ascontiguousarray(data)).permute(0, 3, 1, 2).contiguous()
        pyr = self.pyramid(image)
        p2 = pyr[0]
        a, (h, w) = self._num_anchors, p2.shape[2:]
        anchors = torch.from_numpy(generate_anchors_pre(h, w, self._feat_stride, self._anchor_scales, self._anchor_ratios,
                                                        float(info[6]))[0])
        rpn = F.relu(self.rpn_net(p2))
        cls_score = self.rpn_cls_score_net(rpn)
        bbox_pred = self.rpn_bbox_pred_net(rpn).permute(0, 2, 3, 1).contiguous()
        if structured is not None:
            cls_score, bbox_pred = structured
        prob = F.softmax(cls_score.view(1, 2, a * h, w), dim=1).view(1, 2 * a, h, w).permute(0, 2, 3, 1).contiguous()
        rois, scores, dbg = proposal_layer(prob, bbox_pred, info, anchors, a, return_debug=True)
        pool5, levels = self.pool(pyr, rois, image.shape[2:])
        fc7 = self.tail(pool5)
        cls_score_d, det_box = self.cls_score_net(fc7), self.bbox_pred_net(fc7)
        cls_prob = F.softmax(cls_score_d, dim=1)
        stds = torch.tensor(BBOX_NORMALIZE_STDS).repeat(self._num_classes).unsqueeze(0)
        means = torch.tensor(BBOX_NORMALIZE_MEANS).repeat(self._num_classes).unsqueeze(0)
        pred_boxes = bbox_transform_inv(rois[:, 1:5], det_box.mul(stds).add(means), float(info[6]))
        self._dbg = {"anchors": anchors, "rpn_cls_prob": prob, "rpn_bbox_pred": bbox_pred, "rpn_cls_score": cls_score,
                     "pyramid": pyr, "pool5": pool5, "fc7": fc7, "levels": levels, "bbox_pred": det_box, **dbg}
        return cls_score_d, cls_prob, pred_boxes, rois, {}

    def train_forward(self, data, info, gt_boxes, generator=None, pre_nms=12000, post_nms=2000, proposals=None):
        """One TRAIN forward: returns the dict of losses (torch scalars with a graph) and the sampled targets.
        ``proposals=(rois (N,5), scores (N,1))`` replaces the proposal_layer output (tests: a random-init RPN never
        proposes a foreground box)."""
        image = torch.from_numpy(np.ascontiguousarray(data)).permute(0, 3, 1, 2).contiguous()
        gt = torch.as_tensor(gt_boxes, dtype=torch.float32)
        pyr = self.pyramid(image)
        p2 = pyr[0]
        a, (h, w) = self._num_anchors, p2.shape[2:]
        anchors = torch.from_numpy(generate_anchors_pre(h, w, self._feat_stride, self._anchor_scales, self._anchor_ratios,
                                                        float(info[6]))[0])
        rpn = F.relu(self.rpn_net(p2))
        cls_score = self.rpn_cls_score_net(rpn)                                # (1,2A,H,W)
        bbox_pred = self.rpn_bbox_pred_net(rpn).permute(0, 2, 3, 1).contiguous()   # (1,H,W,4A)
        with torch.no_grad():
            prob = F.softmax(cls_score.view(1, 2, a * h, w), dim=1).view(1, 2 * a, h, w).permute(0, 2, 3, 1).contiguous()
            rois, scores = proposal_layer(prob, bbox_pred, info, anchors, a, pre_nms, post_nms, TEST_RPN_NMS_THRESH)
            if proposals is not None:
                rois, scores = proposals
            lab, tgt, inw, outw = anchor_target_layer(gt, info, anchors, a, h, w, generator=generator)
            pt = proposal_target_layer(rois, scores, torch.zeros(rois.shape[0], 7), gt, None, self._num_classes, 4,
                                       generator=generator)
        pl, prois, _, _, ptgt, pin, pout = pt
        # RPN losses: logits paired (a, a+A); labels (1,A,H,W) -> (H,W,A) order
        logits = torch.stack((cls_score[0, :a].permute(1, 2, 0).reshape(-1), cls_score[0, a:].permute(1, 2, 0).reshape(-1)), 1)
        labels_hwa = lab[0].permute(1, 2, 0).reshape(-1)
        sel = labels_hwa >= 0
        rpn_ce = F.cross_entropy(logits[sel], labels_hwa[sel].long())
        rpn_box = smooth_l1_loss("RPN", bbox_pred, tgt, inw, outw, dim=(1, 2, 3))
        pool5, levels = self.pool(pyr, prois, image.shape[2:])
        fc7 = self.tail(pool5)
        det_cls, det_box = self.cls_score_net(fc7), self.bbox_pred_net(fc7)
        ce = F.cross_entropy(det_cls, pl.view(-1).long())
        box = smooth_l1_loss("DET", det_box, ptgt, pin, pout)
        losses = {"rpn_cross_entropy": rpn_ce, "rpn_loss_box": rpn_box, "cross_entropy": ce, "loss_box": box,
                  "total_loss": rpn_ce + rpn_box + ce + box}
        dbg = {"anchors": anchors, "anchor_labels": labels_hwa, "anchor_targets": tgt.reshape(-1, 4),
               "anchor_inside": inw.reshape(-1, 4), "anchor_outside": outw.reshape(-1, 4), "rois_all": rois,
               "rois": prois, "labels": pl.view(-1), "targets": ptgt, "inside": pin, "outside": pout, "levels": levels,
               "pyramid": pyr, "pool5": pool5, "fc7": fc7, "cls_score": det_cls, "bbox_pred": det_box,
               "rpn_cls_score": cls_score, "rpn_bbox_pred": bbox_pred}
        return losses, dbg


# ----------------------------------------------------------------------------------------------
# LiDAR detector on the FPN backbone — lib/nets/lidarnet.py:31-40,136-146 (cfg.USE_FPN with NET_TYPE 'lidar'):
# the image pyramid of FpnNetOracle with the 15-plane stem, the 3-D anchor grid on p2 (stride 4) and the 7-DoF heads.
# RECONSTRUCTED like the other Network-level restatements.
# ----------------------------------------------------------------------------------------------
class LidarFpnNetOracle(FpnNetOracle):
    def __init__(self, num_classes=2):
        nn.Module.__init__(self)
        self._num_classes = num_classes
        self._num_anchors = len(LIDAR_ANCHOR_SCALES) * len(LIDAR_ANCHOR_ANGLES)
        self._feat_stride = 4
        self.resnet = ResNet101(in_channels=LIDAR_NUM_CHANNEL, use_fpn=True)      # layer4 keeps its BatchNorm (:38)
        self._fpn = FPN(256)
        a = self._num_anchors
        self.rpn_net = nn.Conv2d(256, RPN_CHANNELS, 3, padding=1)
        self.rpn_cls_score_net = nn.Conv2d(RPN_CHANNELS, 2 * a, 1)
        self.rpn_bbox_pred_net = nn.Conv2d(RPN_CHANNELS, 4 * a, 1)
        self.cls_score_net = nn.Linear(2048, num_classes)
        self.bbox_pred_net = nn.Linear(2048, num_classes * LIDAR_NUM_BBOX_ELEM)
        self.t_fc1 = nn.Linear(POOLING_SIZE * POOLING_SIZE * 256, 2048)
        self.t_fc2 = nn.Linear(2048, 2048)
        self.t_fc3 = nn.Linear(2048, 2048)
        self.eval()

    set_trainable = _lidar_set_trainable
    train_mode = _lidar_train_mode

    def _rpn(self, p2, info):
        a = self._num_anchors
        h, w = p2.shape[2], p2.shape[3]
        _, a3 = generate_anchors_3d(h, w, self._feat_stride, frame_scale=float(info[6]))
        anchors = torch.from_numpy(bbaa_graphics_gems(a3))
        rpn = F.relu(self.rpn_net(p2))
        cls_score = self.rpn_cls_score_net(rpn)
        bbox_pred = self.rpn_bbox_pred_net(rpn).permute(0, 2, 3, 1).contiguous()
        return a3, anchors, cls_score, bbox_pred

    @torch.no_grad()
    def test_frame(self, data, info):
        image = torch.from_numpy(np.ascontiguousarray(data)).permute(0, 3, 1, 2).contiguous()
        pyr = self.pyramid(image)
        a = self._num_anchors
        a3, anchors, cls_score, bbox_pred = self._rpn(pyr[0], info)
        h, w = pyr[0].shape[2:]
        prob = F.softmax(cls_score.view(1, 2, a * h, w), dim=1).view(1, 2 * a, h, w).permute(0, 2, 3, 1).contiguous()
        rois, scores, dbg = proposal_layer(prob, bbox_pred, info, anchors, a, return_debug=True)
        roi_a3 = torch.from_numpy(a3)[dbg["order"]][dbg["keep"]]
        pool5, levels = self.pool(pyr, rois, image.shape[2:])
        fc7 = self.tail(pool5)
        cls_score_d, det_box = self.cls_score_net(fc7), self.bbox_pred_net(fc7)
        cls_prob = F.softmax(cls_score_d, dim=1)
        stds = torch.tensor(LIDAR_BBOX_NORMALIZE_STDS).repeat(self._num_classes).unsqueeze(0)
        means = torch.tensor(LIDAR_BBOX_NORMALIZE_MEANS).repeat(self._num_classes).unsqueeze(0)
        pred_boxes = lidar_3d_bbox_transform_inv(rois[:, 1:5], roi_a3, det_box.mul(stds).add(means), float(info[6]))
        self._dbg = {"anchors": anchors, "anchors_3d": torch.from_numpy(a3), "roi_anchors_3d": roi_a3,
                     "rpn_cls_prob": prob, "rpn_bbox_pred": bbox_pred, "pyramid": pyr, "pool5": pool5, "fc7": fc7,
                     "levels": levels, "bbox_pred": det_box, **dbg}
        return cls_score_d, cls_prob, pred_boxes, rois, {}

    def train_forward(self, data, info, true_gt_boxes, generator=None, pre_nms=12000, post_nms=2000, proposals=None):
        """TRAIN forward; ``proposals=(rois (N,5), scores (N,1), anchors_3d (N,7))`` replaces the proposal_layer output."""
        image = torch.from_numpy(np.ascontiguousarray(data)).permute(0, 3, 1, 2).contiguous()
        tgt3 = torch.as_tensor(true_gt_boxes, dtype=torch.float32)
        gt = torch.cat((torch.from_numpy(bbaa_graphics_gems(tgt3[:, :7].numpy())), tgt3[:, 7:8]), 1)
        pyr = self.pyramid(image)
        a = self._num_anchors
        a3, anchors, cls_score, bbox_pred = self._rpn(pyr[0], info)
        h, w = pyr[0].shape[2:]
        with torch.no_grad():
            prob = F.softmax(cls_score.view(1, 2, a * h, w), dim=1).view(1, 2 * a, h, w).permute(0, 2, 3, 1).contiguous()
            rois, scores, dbg = proposal_layer(prob, bbox_pred, info, anchors, a, pre_nms, post_nms, TEST_RPN_NMS_THRESH,
                                               return_debug=True)
            roi_a3 = torch.from_numpy(a3)[dbg["order"]][dbg["keep"]]
            if proposals is not None:
                rois, scores, roi_a3 = proposals
            lab, tgt, inw, outw = anchor_target_layer(gt, info, anchors, a, h, w, generator=generator)
            pl, prois, pa3, _, ptgt, pin, pout = proposal_target_layer(rois, scores, roi_a3, gt, tgt3, self._num_classes,
                                                                       7, net_type="lidar", generator=generator)
        logits = torch.stack((cls_score[0, :a].permute(1, 2, 0).reshape(-1), cls_score[0, a:].permute(1, 2, 0).reshape(-1)), 1)
        labels_hwa = lab[0].permute(1, 2, 0).reshape(-1)
        sel = labels_hwa >= 0
        rpn_ce = F.cross_entropy(logits[sel], labels_hwa[sel].long())
        rpn_box = smooth_l1_loss("RPN", bbox_pred, tgt, inw, outw, dim=(1, 2, 3))
        pool5, levels = self.pool(pyr, prois, image.shape[2:])
        fc7 = self.tail(pool5)
        det_cls, det_box = self.cls_score_net(fc7), self.bbox_pred_net(fc7)
        ce = F.cross_entropy(det_cls, pl.view(-1).long())
        box = smooth_l1_loss("DET", det_box, ptgt, pin, pout, net_type="lidar")
        losses = {"rpn_cross_entropy": rpn_ce, "rpn_loss_box": rpn_box, "cross_entropy": ce, "loss_box": box,
                  "total_loss": rpn_ce + rpn_box + ce + box}
        dbg = {"anchor_labels": labels_hwa, "anchor_targets": tgt.reshape(-1, 4), "anchor_inside": inw.reshape(-1, 4),
               "anchor_outside": outw.reshape(-1, 4), "rois": prois, "anchors_3d": pa3, "labels": pl.view(-1),
               "targets": ptgt, "inside": pin, "outside": pout, "levels": levels, "pyramid": pyr, "gt_aabb": gt}
        return losses, dbg


# ----------------------------------------------------------------------------------------------
# prep_im_for_blob — lib/utils/blob.py:32-54.  cv2.resize(INTER_LINEAR) is restated from its documented
# semantics (cv2 is neither vendored nor installed here: PARITY UNPINNED): dsize = round-half-even(size*scale),
# source coordinate (dst+0.5)/scale - 0.5 in fp32, clamped taps, horizontal blend then vertical.
# ----------------------------------------------------------------------------------------------
def _resize_taps(n_out, inv_scale, size):
    f = ((np.arange(n_out, dtype=np.float64) + 0.5) * np.float64(np.float32(inv_scale)) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    frac = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    frac[lo], s[lo] = 0.0, 0
    hi = s >= size - 1
    frac[hi], s[hi] = 0.0, size - 1
    return s, np.minimum(s + 1, size - 1), (np.float32(1.0) - frac).astype(np.float32), frac


def prep_im_for_blob(im, pixel_means, pixel_stddev, pixel_arrange, im_scale):
    im = im.astype(np.float32, copy=False)
    h, w = im.shape[:2]
    oh, ow = int(np.rint(h * float(im_scale))), int(np.rint(w * float(im_scale)))
    inv = np.float32(1.0 / float(im_scale))
    y0, y1, b0, b1 = _resize_taps(oh, inv, h)
    x0, x1, a0, a1 = _resize_taps(ow, inv, w)
    top = im[y0][:, x0] * a0[None, :, None] + im[y0][:, x1] * a1[None, :, None]
    bot = im[y1][:, x0] * a0[None, :, None] + im[y1][:, x1] * a1[None, :, None]
    out = (top * b0[:, None, None] + bot * b1[:, None, None]).astype(np.float32)
    out = out[:, :, list(pixel_arrange)]
    out -= np.asarray(pixel_means, dtype=np.float64)                    # in-place: float32(float64 - mean)
    return (out / np.asarray(pixel_stddev)).astype(np.float32)          # float64 quotient, stored as float32
