"""``Network`` — base class of the detectors, rebuilt from the contract the reference's callers rely on.

The reference's own ``lib/nets/network.py`` is absent from the snapshot; this class provides the surface
that ``tools/test_net.py:265-290``, ``lib/model/test.py:68-93`` and ``lib/model/train_val.py:167-213,
410-414,449,458`` call and that the subclasses ``lib/nets/imagenet.py:29-134`` / ``lib/nets/lidarnet.py``
fill in:

    create_architecture(num_classes, tag, anchor_scales, anchor_ratios)
    _image_to_head / _anchor_component / _region_proposal / _crop_pool_layer / _head_to_tail /
    _region_classification / _predict / forward / test_frame / set_e_num_sample
    hooks implemented by subclasses: _init_head_tail(), init_weights(); attributes _feat_stride, _fpn_en,
    _net_conv_channels, _roi_pooling_channels, _fc7_channels, _det_net_channels, ...

Every tensor operation of the forward pass is a libfrcnn_hip.so kernel (see ``ops.py``); torch provides
parameter storage, device memory and the stream only.  Tensors handed across the public methods are
NCHW-shaped like the reference's (``net_conv`` is (1, C, H, W)) but physically NHWC (channels_last), so
no layout transposition ever runs on the device.

Choices the missing file leaves open, fixed here as named constants (documented in DESIGN.md):
    ROI_ALIGN_SAMPLING_RATIO = 0  (adaptive ceil(roi/7); ancestor pytorch-faster-rcnn convention)
    spatial_scale = 1 / _feat_stride;  RPN softmax pairs channel a (bg) with channel a+A (fg).
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..layer_utils.anchor_target_layer import anchor_target_layer_device
from ..layer_utils.generate_3d_anchors import generate_anchors_3d
from ..layer_utils.proposal_layer import proposal_layer_device
from ..layer_utils.proposal_target_layer import proposal_target_layer_device
from ..layer_utils.snippets import generate_anchors_pre
from ..model.config import cfg
from ..utils.bbox import bbaa_graphics_gems
from . import resnet as custom_resnet
from . import autograd_ops
from . import uncertainty
from .autograd_ops import (_Holder, conv_bn_act_train, conv_on_patches_train, det_loss_train, fused_head_train, fused_head_weights, linear_train,
                           roi_align_train, rpn_loss_train, spatial_mean_train)
from .hip_modules import conv_bn_act, pad4, prepared_conv, prepared_conv_concat, to_nchw_view, to_nhwc

ROI_ALIGN_SAMPLING_RATIO = 0
# inference runs layer4[0]'s input-side 1x1 convolutions on the feature map, before the RoIAlign (Network._layer4_projected);
# False: the reference's order of operations (pool, then convolve 300 x 7 x 7 pixels)
PROJECT_BEFORE_POOLING = True
# ... as ONE 1x1 convolution with concatenated filters, pooled through ONE RoIAlign plan (frcnn_roi_align_fwd_split);
# False: two convolutions and two complete RoIAlign calls (round 3)
FUSE_PROJECTIONS = True
# training runs the RPN's differentiable pass on the pixels that carry a labelled anchor only (Network._rpn_losses_on_labelled_pixels);
# False: dense backward through the whole RPN head
RPN_BACKWARD_ON_LABELLED_PIXELS = True
UC_SEED_RANK_STRIDE = 7919        # decorrelates the uncertainty heads' random draws across data-parallel ranks
# FPN choices the missing network.py leaves open (DESIGN.md "reconstructed contract"):
FPN_RPN_LEVEL = 0                 # the RPN runs on p2 only (_feat_stride = 4, lib/nets/imagenet.py:34)
FPN_POOL_LEVELS = (2, 5)          # MultiScaleRoIAlign over p2..p5 (k_min, k_max of LevelMapper)
NO_CANDIDATES = "proposal_target_layer: neither foreground nor background candidate RoIs in this frame"
CUSTOM_TAIL_WIDTH = None          # t_fc1: P*P*C -> _fc7_channels, t_fc2 / t_fc3: _fc7_channels -> _fc7_channels


class Network(nn.Module):
    def __init__(self):
        nn.Module.__init__(self)
        self._predictions = {}
        self._losses = {}
        self._anchor_targets = {}
        self._proposal_targets = {}
        self._layers = {}
        self._act_summaries = {}
        self._score_summaries = {}
        self._event_summaries = {}
        self._gt_summaries = {}
        self._variables_to_fix = {}
        self._device = 'cuda'
        self._e_num_sample = 1
        self._frame_scale = 1.0
        self.timers = {}
        self._rpn_fused = None
        self._uc_seed = int(cfg.RNG_SEED)      # counter-based draws of the uncertainty heads: seed + forward count
        self._uc_calls = 0
        self._uc_seed_dev = None

    # ------------------------------------------------------------------------------------------
    # construction
    # ------------------------------------------------------------------------------------------
    def create_architecture(self, num_classes, tag=None, anchor_scales=(8, 16, 32), anchor_ratios=(0.5, 1, 2)):
        self._tag = tag
        self._num_classes = int(num_classes)
        self._anchor_scales = [float(s) for s in np.asarray(anchor_scales).ravel()]
        self._num_scales = len(self._anchor_scales)
        self._anchor_ratios = [float(r) for r in np.asarray(anchor_ratios).ravel()]
        self._num_ratios = len(self._anchor_ratios)
        self._num_anchors = self._num_scales * self._num_ratios
        assert tag is not None
        self._init_modules()

    def _bbox_elem(self):
        return cfg[cfg.NET_TYPE.upper()].NUM_BBOX_ELEM if cfg.NET_TYPE in ('image', 'lidar') else 4

    def _init_modules(self):
        self._init_head_tail()
        # RPN: names pinned by lib/nets/imagenet.py:66,83-84
        self.rpn_net = nn.Conv2d(self._net_conv_channels, cfg.RPN_CHANNELS, kernel_size=3, padding=1)
        self.rpn_cls_score_net = nn.Conv2d(cfg.RPN_CHANNELS, self._num_anchors * 2, kernel_size=1)
        self.rpn_bbox_pred_net = nn.Conv2d(cfg.RPN_CHANNELS, self._num_anchors * 4, kernel_size=1)
        # detection heads: lib/nets/imagenet.py:85-86
        self.cls_score_net = nn.Linear(self._det_net_channels, self._num_classes)
        self.bbox_pred_net = nn.Linear(self._det_net_channels, self._num_classes * self._bbox_elem())
        if cfg.ENABLE_CUSTOM_TAIL:
            # names from lib/nets/imagenet.py:70-73; ReLU MLP P*P*C -> fc7 -> fc7 -> fc7 (sizes: see module docstring)
            width = CUSTOM_TAIL_WIDTH or self._fc7_channels
            self.t_fc1 = nn.Linear(self._roi_pooling_channels, width)
            self.t_fc2 = nn.Linear(width, width)
            self.t_fc3 = nn.Linear(width, self._fc7_channels)
        elif getattr(self, '_fpn_en', False):
            raise NotImplementedError("the FPN detector needs cfg.ENABLE_CUSTOM_TAIL (layer4 is part of its backbone); "
                                      "tools/trainval_net.py:326-330 sets both")
        if uncertainty.enabled():
            uncertainty.check_flags()
            uncertainty.build_modules(self, lidar=cfg.NET_TYPE == 'lidar')
        self.init_weights()

    def _build_resnet(self):
        depth = self._num_resnet_layers
        if depth == 50:
            return custom_resnet.resnet50(dropout_en=self._dropout_en, drop_rate=self._resnet_drop_rate,
                                          batchnorm_en=self._batchnorm_en)
        if depth == 101:
            return custom_resnet.resnet101(dropout_en=self._dropout_en, drop_rate=self._resnet_drop_rate,
                                           batchnorm_en=self._batchnorm_en)
        if depth == 152:
            return custom_resnet.resnet152(dropout_en=self._dropout_en, drop_rate=self._resnet_drop_rate,
                                           batchnorm_en=self._batchnorm_en)
        raise NotImplementedError('resnet depth %s' % depth)

    def set_e_num_sample(self, n):
        """Monte-Carlo passes of the epistemic heads per frame (lib/model/test.py:74-77)."""
        self._e_num_sample = int(n)

    def set_uc_seed(self, seed):
        """Restart the counter-based random draws of the uncertainty heads (dropout masks, logit distortion)."""
        self._uc_seed, self._uc_calls = int(seed), 0

    def next_uc_seed(self):
        """Seed of this forward's counter-based draws: base + 7919 * rank + forward count, so data-parallel replicas draw
        different dropout masks / logit noise; ``_uc_calls`` travels in the solver's per-rank snapshot state."""
        import torch.distributed as dist
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        s = (self._uc_seed + UC_SEED_RANK_STRIDE * rank + self._uc_calls) & 0xFFFFFFFF
        self._uc_calls += 1
        return s

    def uc_seed_args(self):
        """(seed, seed_dev) for the counter-based draws of this forward: eagerly the host counter (``next_uc_seed``) and no
        device word; inside a captured frame / training step (model/frame_graph.py, model/train_graph.py set
        ``_uc_seed_dev``) the scalar 0 and the device word the runner rewrites with ``next_uc_seed()`` before each replay -
        a replayed launch keeps its scalar arguments.  Both forms draw from seed + *seed_dev, so a replayed frame gets the
        masks the eager call of the same forward count gets."""
        sd = getattr(self, '_uc_seed_dev', None)
        if sd is not None:
            return 0, sd
        return self.next_uc_seed(), None

    # ------------------------------------------------------------------------------------------
    # forward pieces (reference names).  Public tensors are NCHW-shaped views of NHWC storage.
    # ------------------------------------------------------------------------------------------
    def _image_to_head(self):
        x = to_nhwc(self._image)
        if getattr(self, '_fpn_en', False):
            c1 = self._layers['head'](x)
            c2 = self._layers['layer1'](c1)
            c3 = self._layers['layer2'](c2)
            c4 = self._layers['layer3'](c3)
            c5 = self._layers['layer4'](c4)
            self._pyramid = list(self._layers['fpn'](c2, c3, c4, c5))          # p2, p3, p4, p5 (NHWC)
            net_conv = self._pyramid[FPN_RPN_LEVEL]
        else:
            self._pyramid = None
            net_conv = self._layers['head'](x)
        self._act_summaries['conv'] = net_conv
        return to_nchw_view(net_conv)

    _input_to_head = _image_to_head  # name used by the other backbones of the reference (vgg16.py:49)

    def _anchor_component(self, height, width):
        """Image: dense 2-D anchors (snippets.py:13-40).  LiDAR: 3-D grid anchors (generate_3d_anchors.py) and
        their axis-aligned BEV rectangles, which the RPN regresses against; ``anchor_scales`` is the single size
        multiplier and ``anchor_ratios`` carries the yaw angles (tools/test_net.py:267-271)."""
        if cfg.NET_TYPE == 'lidar':
            length, a3, a2 = generate_anchors_3d(height, width, self._feat_stride, self._anchor_scales,
                                                 self._anchor_ratios, self._frame_scale, device=self._image.device)
            self._anchors_3d = a3
            self._anchors = a2
            self._anchor_length = np.int32(length)
            return a2
        anchors, length = generate_anchors_pre(height, width, self._feat_stride, self._anchor_scales,
                                               self._anchor_ratios, self._frame_scale, device=self._image.device)
        self._anchors = anchors
        self._anchors_3d = None
        self._anchor_length = length
        return anchors

    def _fused_rpn_head(self):
        """rpn_cls_score_net and rpn_bbox_pred_net share their input, so they run as ONE 1x1 conv with the filters
        concatenated and zero-padded to a multiple of 4 outputs: channels [0,2A) = class logits, [2A,6A) = box
        deltas, [6A, ld) = padding (never read)."""
        return fused_head_weights(self, self.rpn_cls_score_net, self.rpn_bbox_pred_net, '_rpn_fused_cache')

    def _rpn_head(self, net_conv_nhwc):
        """relu(rpn_net) then the fused cls+bbox 1x1.  Returns (1, H, W, ld >= 6A) NHWC logits|deltas."""
        self._rpn_grad_src = None
        hw = net_conv_nhwc.shape[1] * net_conv_nhwc.shape[2]
        if torch.is_grad_enabled() and not self._rpn_backward_on_labelled_pixels(hw):
            rpn = conv_bn_act_train(net_conv_nhwc, self.rpn_net, None, relu=True)
            self._act_summaries['rpn'] = rpn
            return fused_head_train(rpn, self, self.rpn_cls_score_net, self.rpn_bbox_pred_net, '_rpn_fused_cache')
        if torch.is_grad_enabled():
            # training: the dense head only feeds the proposal layer (which detaches it anyway, proposal_layer.py:18-57); the
            # differentiable pass runs later on the labelled pixels alone (_rpn_losses_on_labelled_pixels)
            with torch.no_grad():
                out = self._rpn_head(net_conv_nhwc.detach())
            self._rpn_grad_src = net_conv_nhwc
            return out
        rpn = conv_bn_act(net_conv_nhwc, self.rpn_net, None, relu=True)
        self._act_summaries['rpn'] = rpn
        w, b = self._fused_rpn_head()
        return ops.conv2d_nhwc(rpn, w, None, b, None, stride=1, pad=0, relu=False)

    def _rpn_backward_on_labelled_pixels(self, hw):
        """Training: is the RPN's differentiable pass restricted to the pixels that carry a labelled anchor?  Possible for the
        plain losses (cross-entropy over labels != -1, smooth-L1 with zero inside-weights elsewhere: no other anchor
        contributes to loss or gradient), not with the RPN uncertainty heads, whose terms read every anchor; worthwhile when
        the map has several times more pixels than the sampler labels anchors (the LiDAR detector's 25 x 22 map has not)."""
        if not RPN_BACKWARD_ON_LABELLED_PIXELS or self._mode != 'TRAIN' or hw < 4 * int(cfg.TRAIN.RPN_BATCHSIZE):
            return False
        if any(cfg.UC.get(k, False) for k in ('EN_RPN_BBOX_ALEATORIC', 'EN_RPN_CLS_ALEATORIC', 'EN_RPN_BBOX_EPISTEMIC',
                                               'EN_RPN_CLS_EPISTEMIC')):
            return False
        conv = self.rpn_net
        return tuple(conv.stride) == (1, 1) and conv.groups == 1 and tuple(conv.dilation) == (1, 1)

    def _rpn_losses_on_labelled_pixels(self, at):
        """(cross-entropy, box loss) of the RPN and their gradients, evaluated on the labelled pixels only.

        The anchor target layer labels at most cfg.TRAIN.RPN_BATCHSIZE anchors (anchor_target_layer.py:91-107); every other
        anchor has label -1 and zero inside-weights, i.e. contributes neither to the losses nor to any gradient.  So the
        head is re-evaluated - differentiably - at the <= 256 pixels that carry a labelled anchor: rpn_net as a VALID
        convolution of the 3x3 windows around them, the fused cls | bbox 1x1 on the 256 results, the same loss kernel on
        the gathered target rows.  Same loss, same gradients (zeros are simply not summed); the backward of the two RPN
        convolutions shrinks from H x W to 256 output pixels (on the FPN's p2: 37 500 -> 256, 177 GFLOP per step)."""
        x, self._rpn_grad_src = self._rpn_grad_src, None
        hw, a = x.shape[1] * x.shape[2], self._num_anchors
        labels = at['labels'].contiguous().view(-1)
        cap = int(cfg.TRAIN.RPN_BATCHSIZE)
        if getattr(self, '_target_override', None):
            # injected targets (tests) need not respect the sampler's cap: size the list for them (host sync, tests only)
            cap = max(cap, int((labels.view(hw, a) != -1).any(1).sum().item()))
        idx, count = ops.labelled_pixels(labels, hw, a, cap)
        hidden = conv_on_patches_train(x, idx, count, self.rpn_net, _Holder.of(self, '_rpn_patch'), relu=True)
        out = fused_head_train(hidden, self, self.rpn_cls_score_net, self.rpn_bbox_pred_net, '_rpn_fused_cache').view(cap, -1)
        live = count[0:1]
        # rows past the count: label -1 (ignored), zero targets and weights
        lab = ops.gather_rows((labels + 1.0).view(hw, a), idx, live) - 1.0
        tgt, inw, outw = (ops.gather_rows(at[k].contiguous().view(hw, 4 * a), idx, live).view(cap * a, 4)
                          for k in ('targets', 'inside', 'outside'))
        self._predictions['rpn_labelled_pixels'] = (idx, count)
        losses = rpn_loss_train(out, lab.view(-1), tgt, inw, outw, a)
        # more labelled pixels than the list holds cannot happen with the sampler's cap; if it ever did, the step must not
        # train on a truncated loss silently: the losses (and through them every gradient) turn NaN
        poison = torch.where(count[1:2] > cap, float('nan'), 0.0).to(losses.dtype)
        return losses + poison

    def _region_proposal(self, net_conv):
        """RPN head -> proposal_layer.  Returns rois (post_nms_topN, 5) [0,x1,y1,x2,y2]; rows past
        ``self._predictions['rois_count']`` (device int) are zero padding."""
        x = to_nhwc(net_conv)
        h, w = x.shape[1], x.shape[2]
        self._anchor_component(h, w)
        rpn_out = self._rpn_head(x)                                   # (1, H, W, ld >= 6A)
        override = getattr(self, '_rpn_override', None)               # evaluation hook: injected RPN logits | deltas
        if override is not None:
            rpn_out = override
        key = 'TRAIN' if self._mode == 'TRAIN' else 'TEST'
        if key == 'TEST' and cfg.TEST.get('MODE', 'nms') == 'top':
            return self._proposal_top(rpn_out, h, w)
        with torch.no_grad():   # proposals carry no gradient (proposal_layer.py works on detached scores/deltas)
            res = proposal_layer_device(self._anchors, self._info, self._num_anchors, cfg[key].RPN_PRE_NMS_TOP_N,
                                        cfg[key].RPN_POST_NMS_TOP_N, cfg[key].RPN_NMS_THRESH,
                                        rpn=rpn_out.detach().view(h * w, rpn_out.shape[-1]))
        self._predictions['rpn_out'] = rpn_out
        self._predictions['rpn_scores'] = res.scores
        self._predictions['rpn_proposals'] = res.proposals
        self._predictions['rpn_order'] = res.order
        self._predictions['rpn_keep'] = res.keep_idx
        self._predictions['rois'] = res.rois
        self._predictions['roi_scores'] = res.roi_scores
        self._predictions['rois_count'] = res.count
        if self._anchors_3d is not None:
            # proposal_layer.py:44,52: the 3-D anchors follow the same order -> keep selection as the boxes
            a3_sorted = ops.gather_rows(self._anchors_3d, res.order, res.sorted_count)
            self._predictions['roi_anchors_3d'] = ops.gather_rows(a3_sorted, res.keep_idx, res.count)
        return res.rois

    def _proposal_top(self, rpn_out, h, w):
        """cfg.TEST.MODE == 'top' (lib/model/config.py:263): RPN_TOP_N best anchors, no NMS (proposal_top_layer.py)."""
        from ..layer_utils.proposal_top_layer import proposal_top_layer
        a = self._num_anchors
        flat = rpn_out.detach().view(h * w, rpn_out.shape[-1])
        zero_anchor = torch.zeros_like(self._anchors)
        # 2-way softmax through the decode kernel (its box output is ignored here), then the reference-shaped call
        fg_prob, _ = ops.rpn_decode_clip(zero_anchor, self._info, a, rpn=flat)
        fgv = fg_prob.view(1, h, w, a)
        prob = torch.cat((torch.zeros_like(fgv), fgv), 3)      # proposal_top_layer reads the fg half only (:26)
        deltas = flat[:, 2 * a:6 * a].contiguous().view(1, h, w, 4 * a)
        rois, scores, _ = proposal_top_layer(prob, deltas, self._info, self._anchors, a)
        p = self._predictions
        p['rpn_out'], p['rois'], p['roi_scores'] = rpn_out, rois, scores
        p['rois_count'] = torch.full((1,), rois.shape[0], dtype=torch.int32, device=rois.device)
        return rois

    # ------------------------------------------------------------------------------------------
    # ancestor-named layer methods (the reconstructed method set of the missing network.py): thin
    # reference-shaped calls into the same device layers the fused pipeline above uses
    # ------------------------------------------------------------------------------------------
    def _proposal_layer(self, rpn_cls_prob, rpn_bbox_pred):
        """rpn_cls_prob (1,H,W,2A) fg half last, rpn_bbox_pred (1,H,W,4A) -> (rois (n,5), rpn_scores (n,1))."""
        from ..layer_utils.proposal_layer import proposal_layer
        rois, scores, a3 = proposal_layer(rpn_cls_prob, rpn_bbox_pred, self._info, self._mode, self._anchors,
                                          self._anchors_3d, self._num_anchors)
        if a3 is not None:
            self._predictions['roi_anchors_3d'] = a3
        return rois, scores

    def _proposal_top_layer(self, rpn_cls_prob, rpn_bbox_pred):
        from ..layer_utils.proposal_top_layer import proposal_top_layer
        rois, scores, _ = proposal_top_layer(rpn_cls_prob, rpn_bbox_pred, self._info, self._anchors, self._num_anchors)
        return rois, scores

    def _anchor_target_layer(self, rpn_cls_score):
        """rpn_cls_score (1,2A,H,W) only supplies the map size, as in the ancestor.  Fills self._anchor_targets."""
        from ..layer_utils.anchor_target_layer import anchor_target_layer_torch
        h, w = rpn_cls_score.shape[2], rpn_cls_score.shape[3]
        out = anchor_target_layer_torch(self._gt_boxes, None, self._info, self._anchors, self._num_anchors, h, w)
        self._anchor_targets = dict(zip(('rpn_labels', 'rpn_bbox_targets', 'rpn_bbox_inside_weights',
                                         'rpn_bbox_outside_weights'), out))
        return out[0]

    def _proposal_target_layer(self, rois, roi_scores):
        """Returns (rois, roi_scores) of the sampled rows; the targets go to self._proposal_targets."""
        from ..layer_utils.proposal_target_layer import proposal_target_layer
        lidar = cfg.NET_TYPE == 'lidar'
        labels, rois_s, a3, scores_s, tgt, inw, outw = proposal_target_layer(
            rois, roi_scores, self._predictions.get('roi_anchors_3d') if lidar else None, self._gt_boxes,
            getattr(self, '_true_gt_boxes', None) if lidar else None, None, self._num_classes, self._bbox_elem())
        self._proposal_targets = {'rois': rois_s, 'labels': labels.view(-1), 'targets': tgt, 'inside': inw, 'outside': outw}
        if lidar:
            self._proposal_targets['anchors_3d'] = a3
            self._predictions['roi_anchors_3d'] = a3
        return rois_s, scores_s

    def _pyramid_scales(self):
        """MultiScaleRoIAlign.infer_scale (lib/utils/torchpoolers.py:107-117): 2 ** round(log2(feat / image))."""
        img_h = float(self._image.shape[2])
        return [2.0 ** float(np.round(np.log2(float(f.shape[1]) / img_h))) for f in self._pyramid]

    def _crop_pool_layer(self, bottom, rois):
        """RoIAlign 7x7: single level for POOLING_MODE 'align' (lib/model/config.py:364); over p2..p5 with the FPN
        level heuristic for 'multiscale' (lib/utils/torchpoolers.py:137-200).  (R, C, 7, 7) NCHW-shaped."""
        rois = rois.contiguous()
        if self._pyramid is not None and cfg.POOLING_MODE == 'multiscale':
            levels = ops.fpn_level_map(rois, FPN_POOL_LEVELS[0], FPN_POOL_LEVELS[1])
            self._predictions['roi_levels'] = levels
            pooled = roi_align_train(self._pyramid, rois, levels, cfg.POOLING_SIZE, self._pyramid_scales(),
                                     ROI_ALIGN_SAMPLING_RATIO)
        elif torch.is_grad_enabled() and bottom.requires_grad:
            pooled = roi_align_train([to_nhwc(bottom)], rois, None, cfg.POOLING_SIZE, [1.0 / self._feat_stride],
                                     ROI_ALIGN_SAMPLING_RATIO)
        else:
            pooled = ops.roi_align_nhwc(to_nhwc(bottom), rois, cfg.POOLING_SIZE, 1.0 / self._feat_stride,
                                        ROI_ALIGN_SAMPLING_RATIO, roi_count=self._predictions.get('rois_count'))
        return to_nchw_view(pooled)

    _roi_align_layer = _crop_pool_layer   # ancestor name of the same step

    def _layer4(self, pool5_nhwc):
        return self.resnet.layer4(pool5_nhwc)

    def _tail_kernel(self, x_nhwc, rois):
        key = cfg.NET_TYPE.upper() if cfg.NET_TYPE in ('image', 'lidar') else 'IMAGE'
        return ops.head_fc_softmax_decode(
            x_nhwc, self.cls_score_net.weight.detach(), self.cls_score_net.bias.detach(),
            self.bbox_pred_net.weight.detach(), self.bbox_pred_net.bias.detach(), rois.contiguous(),
            cfg.TRAIN[key].BBOX_NORMALIZE_STDS, cfg.TRAIN[key].BBOX_NORMALIZE_MEANS, self._frame_scale,
            roi_anchors_3d=self._predictions.get('roi_anchors_3d') if key == 'LIDAR' else None)

    def _head_to_tail(self, pool5):
        """Without FPN: layer4 on the pooled RoIs then ``.mean(3).mean(2)`` -> fc7 (R, 2048).
        With the custom tail (FPN): relu(t_fc1) -> relu(t_fc2) -> relu(t_fc3) on the flattened (C,7,7) RoI features;
        the NHWC flattening is absorbed into a column permutation of t_fc1's weight."""
        if cfg.ENABLE_CUSTOM_TAIL:
            x = to_nhwc(pool5)
            r, p, _, c = x.shape
            h = linear_train(x.reshape(r, p * p * c), self.t_fc1, relu=True, weight_nhwc_from=(c, p))
            h = linear_train(h, self.t_fc2, relu=True)
            return linear_train(h, self.t_fc3, relu=True)
        if torch.is_grad_enabled() and self._mode == 'TRAIN':
            return spatial_mean_train(self._layer4(to_nhwc(pool5)))      # Bottleneck nodes + mean, all differentiable
        return self._fc7_from_layer4(self._layer4(to_nhwc(pool5)))

    def _fc7_from_layer4(self, y):
        # layer4 output (R,7,7,2048) -> fc7 in a chip-wide streaming pass (HBM-bound, 120 MB at 300 RoIs), then the
        # heads on fc7 alone: one workgroup per RoI doing both needed 72 us (profiles/r01h_kernel_stats.md)
        fc7 = ops.spatial_mean(y)
        if uncertainty.enabled():
            return fc7                                  # the heads run in _region_classification (nets/uncertainty.py)
        r, c = fc7.shape
        out = self._tail_kernel(fc7.view(r, 1, 1, c), self._predictions['rois'])
        out['fc7'] = fc7
        self._predictions['_tail'] = out
        return fc7

    # ---- inference: layer4[0]'s two 1x1 convolutions BEFORE the pooling --------------------------------------------------
    def _projected_head_ok(self):
        """True when ``_predict`` may run layer4[0].conv1 and layer4[0].downsample[0] on the feature map instead of on
        pool5 (see ``_layer4_projected``): inference without autograd, single-level 'align' pooling, layer4 as the tail, both
        convolutions 1x1 / stride 1 / no bias and every BatchNorm of the block in eval mode."""
        if not PROJECT_BEFORE_POOLING or self._mode != 'TEST' or torch.is_grad_enabled() or cfg.ENABLE_CUSTOM_TAIL:
            return False
        # the hook protocol: a subclass that supplies its own pooling or tail (lib/nets/vgg16.py:49-59 overrides
        # _head_to_tail, mobilenet_v1.py likewise) must have its methods CALLED - the shortcut replaces exactly these
        cls = type(self)
        if (cls._crop_pool_layer is not Network._crop_pool_layer or cls._roi_align_layer is not Network._roi_align_layer
                or cls._head_to_tail is not Network._head_to_tail or cls._layer4 is not Network._layer4
                or any(k in self.__dict__ for k in ('_crop_pool_layer', '_roi_align_layer', '_head_to_tail', '_layer4'))):
            return False
        if self._pyramid is not None and cfg.POOLING_MODE == 'multiscale':
            return False
        blk = self.resnet.layer4[0]
        if blk.downsample is None:
            return False
        for conv in (blk.conv1, blk.downsample[0]):
            if (tuple(conv.kernel_size) != (1, 1) or tuple(conv.stride) != (1, 1) or tuple(conv.padding) != (0, 0)
                    or conv.bias is not None or conv.groups != 1):
                return False
        return not any(m.training for m in (blk.bn1, blk.bn2, blk.bn3, blk.downsample[1]))

    def _layer4_projected(self, net_conv, rois):
        """layer4 on the pooled RoIs (lib/nets/resnet.py:98-127 applied by ``_head_to_tail`` to pool5) with the block's two
        input-side 1x1 convolutions moved in front of the RoIAlign.

        RoIAlign is linear in the feature map and acts on the pixel axes; a bias-free 1x1 convolution is linear and acts on
        the channel axis: ``conv(roi_align(F)) == roi_align(conv(F))``.  The reference evaluates the left side on
        300 x 7 x 7 = 14 700 pooled pixels; the right side needs the convolution on the 38 x 63 = 2 394 pixels of the map
        once (6.1x fewer multiply-adds for conv1 and the downsample branch: 75.6 -> 12.3 GFLOP per 1000x600 frame).  The
        folded BatchNorm (an affine map with a shift - it does NOT commute with a pooling that drops out-of-map samples)
        and the ReLU stay behind the pooling, in the RoIAlign kernel's store epilogue, exactly where the reference applies
        them.  Same function, another rounding order (differences ~1e-6 relative, tests/test_gpu_parity.py)."""
        blk = self.resnet.layer4[0]
        bn = blk.batchnorm_en
        x = to_nhwc(net_conv)
        w1, s1, b1 = prepared_conv(blk.conv1, blk.bn1, bn)
        wd, sd, bd = prepared_conv(blk.downsample[0], blk.downsample[1])
        if x.shape[-1] != w1.shape[-1]:
            x = ops.pad_channels(x, w1.shape[-1])
        rois = rois.contiguous()
        count = self._predictions.get('rois_count')
        scale = 1.0 / self._feat_stride
        if FUSE_PROJECTIONS and w1.shape[0] % 256 == 0 and cfg.POOLING_SIZE == 7:
            wc, sc, bc = prepared_conv_concat(blk, '_fused_cache_l4proj', [(blk.conv1, blk.bn1, bn),
                                                                           (blk.downsample[0], blk.downsample[1], True)])
            a1, identity = ops.roi_align_split(ops.conv2d_nhwc(x, wc), rois, cfg.POOLING_SIZE, scale, w1.shape[0],
                                               ROI_ALIGN_SAMPLING_RATIO, roi_count=count, scale=sc, shift=bc, relu1=True,
                                               relu2=False)
        else:
            a1 = ops.roi_align_nhwc(ops.conv2d_nhwc(x, w1), rois, cfg.POOLING_SIZE, scale, ROI_ALIGN_SAMPLING_RATIO,
                                    roi_count=count, scale=s1, shift=b1, relu=True)
            identity = ops.roi_align_nhwc(ops.conv2d_nhwc(x, wd), rois, cfg.POOLING_SIZE, scale, ROI_ALIGN_SAMPLING_RATIO,
                                          roi_count=count, scale=sd, shift=bd, relu=False)
        out = conv_bn_act(a1, blk.conv2, blk.bn2, relu=True, use_bn=bn)
        y = conv_bn_act(out, blk.conv3, blk.bn3, relu=True, residual=identity, use_bn=bn)
        for later in list(self.resnet.layer4)[1:]:
            y = later(y)
        return y

    def _region_classification(self, fc7):
        """cls_score_net + softmax, bbox_pred_net.  Returns (cls_prob, bbox_pred) like the ancestor."""
        if uncertainty.enabled():
            if torch.is_grad_enabled() and self._mode == 'TRAIN':
                return uncertainty.classify_train(self, fc7)
            return uncertainty.classify_test(self, fc7.contiguous(), self._predictions['rois'])
        if torch.is_grad_enabled() and self._mode == 'TRAIN':
            r, c = fc7.shape
            k, e = self._num_classes, self._bbox_elem()
            out = fused_head_train(fc7.view(r, 1, 1, c), self, self.cls_score_net, self.bbox_pred_net,
                                   '_det_fused_cache').view(r, -1)
            self._predictions['cls_score'] = out[:, :k]
            self._predictions['bbox_pred'] = out[:, k:k + k * e]
            return None, self._predictions['bbox_pred']
        tail = self._predictions.get('_tail')
        if tail is None or tail['fc7'] is not fc7:
            r, c = fc7.shape
            tail = self._tail_kernel(fc7.contiguous().view(r, 1, 1, c), self._predictions['rois'])
        self._predictions['cls_score'] = tail['cls_score']
        self._predictions['cls_prob'] = tail['cls_prob']
        self._predictions['bbox_pred'] = tail['bbox_pred']
        self._predictions['pred_boxes'] = tail['pred_boxes']
        return tail['cls_prob'], tail['bbox_pred']

    def _predict(self):
        net_conv = self._image_to_head()
        rois = self._region_proposal(net_conv)
        if self._mode == 'TRAIN':
            rois = self._training_targets(rois)
        if self._projected_head_ok():
            fc7 = self._fc7_from_layer4(self._layer4_projected(net_conv, rois))
        else:
            pool5 = self._crop_pool_layer(net_conv, rois)
            fc7 = self._head_to_tail(pool5)
        cls_prob, bbox_pred = self._region_classification(fc7)
        return rois, cls_prob, bbox_pred

    # ------------------------------------------------------------------------------------------
    # training: targets and losses (anchor_target_layer.py, proposal_target_layer.py, loss_utils.py; _add_losses of
    # the missing network.py restated from the ancestor: the four terms are summed with unit weights)
    # ------------------------------------------------------------------------------------------
    def _training_targets(self, rois):
        """RPN anchor targets + second-stage RoI sampling.  ``self._target_override`` (tests) injects precomputed
        targets: dict(anchor=(labels, targets, inside, outside), proposal=dict(rois, labels, targets, inside, outside))."""
        ov = getattr(self, '_target_override', None) or {}
        # per-step sampling seeds: drawn on the host, or (model/train_graph.py) read on the device from self._seed_dev
        sd = getattr(self, '_seed_dev', None)
        gc = getattr(self, '_gt_count_dev', None)          # live rows of a gt buffer padded to capacity (captured steps)
        seed_a = dict(seed=0, seed_dev=sd[0:1], gt_count=gc) if sd is not None else {}
        seed_p = dict(seed=0, seed_dev=sd[1:2], gt_count=gc) if sd is not None else {}
        with torch.no_grad():
            if 'anchor' in ov:
                self._anchor_targets = dict(zip(('labels', 'targets', 'inside', 'outside'), ov['anchor']))
            else:
                lab, tgt, inw, outw, counts = anchor_target_layer_device(self._gt_boxes, self._info, self._anchors, **seed_a)
                self._anchor_targets = {'labels': lab, 'targets': tgt, 'inside': inw, 'outside': outw, 'counts': counts}
            if 'proposal' in ov:
                self._proposal_targets = dict(ov['proposal'])
            else:
                p = self._predictions
                if cfg.NET_TYPE == 'lidar':
                    self._proposal_targets = proposal_target_layer_device(
                        p['rois'], p['roi_scores'], self._gt_boxes, self._num_classes, roi_count=p['rois_count'],
                        anchors_3d=p['roi_anchors_3d'], true_gt_boxes=self._true_gt_boxes, gt_boxes_dc=self._gt_boxes_dc,
                        **seed_p)
                else:
                    self._proposal_targets = proposal_target_layer_device(p['rois'], p['roi_scores'], self._gt_boxes,
                                                                          self._num_classes, roi_count=p['rois_count'],
                                                                          gt_boxes_dc=self._gt_boxes_dc, **seed_p)
        if 'anchors_3d' in self._proposal_targets:
            self._predictions['roi_anchors_3d'] = self._proposal_targets['anchors_3d']   # follows the sampled rows
        self._predictions['rois_sampled'] = self._proposal_targets['rois']
        self._predictions['rois_count'] = None            # every sampled row is live
        return self._proposal_targets['rois']

    def _add_losses(self):
        p, at, pt = self._predictions, self._anchor_targets, self._proposal_targets
        rpn_out = p['rpn_out']
        hw, ld = rpn_out.shape[1] * rpn_out.shape[2], rpn_out.shape[3]
        if getattr(self, '_rpn_grad_src', None) is not None:
            rpn_l = self._rpn_losses_on_labelled_pixels(at)
        else:
            rpn_l = rpn_loss_train(rpn_out.view(hw, ld), at['labels'], at['targets'], at['inside'], at['outside'],
                                   self._num_anchors)
        lidar = (tuple(cfg.LIDAR.REG_LOSS_WEIGHT), bool(cfg.LIDAR.EN_RY_SIN)) if cfg.NET_TYPE == 'lidar' else None
        if cfg.UC.EN_BBOX_ALEATORIC or cfg.UC.EN_CLS_ALEATORIC:
            det_l = uncertainty.det_loss_uc(self, pt['labels'], pt['targets'], pt['inside'], pt['outside'], lidar)
        else:
            det_l = det_loss_train(p['cls_score'], p['bbox_pred'], pt['labels'], pt['targets'], pt['inside'], pt['outside'],
                                   lidar=lidar)
        self._losses = {'rpn_cross_entropy': rpn_l[0], 'rpn_loss_box': rpn_l[1], 'cross_entropy': det_l[0],
                        'loss_box': det_l[1]}
        self._losses['total_loss'] = rpn_l[0] + rpn_l[1] + det_l[0] + det_l[1]
        return self._losses['total_loss']

    def forward(self, image, info, gt_boxes=None, gt_boxes_dc=None, mode='TRAIN'):
        """image: (1,H,W,C) float32 numpy blob (lib/roi_data_layer/minibatch.py:670) or device tensor of
        that layout; info: 7-vector [x_min,x_max,y_min,y_max,z_min,z_max,scale]."""
        dev = torch.device(self._device)
        if dev.type != 'cuda':
            raise RuntimeError("faster_rcnn_pytorch_multimodal_amd runs on the MI355X only (net._device=%r); "
                               "the CPU path of the reference is not part of this package" % (self._device,))
        if isinstance(image, np.ndarray):
            image = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).to(dev, non_blocking=True)
        self._info = np.asarray(info, dtype=np.float32)
        self._frame_scale = float(self._info[6])
        # keep the blob NHWC and pad channels to a multiple of 4 (16-byte pixels for the stem conv);
        # self._image is the NCHW-shaped view the reference exposes
        self._image = to_nchw_view(ops.pad_channels(image.contiguous(), pad4(image.shape[-1])))
        self._mode = mode
        self._predictions = {}
        self._gt_boxes_dc = None
        if mode == 'TEST':
            with torch.no_grad():
                return self._predict()
        if isinstance(gt_boxes, tuple):
            # (BEV rectangles (G,5), 3-D boxes (G,8)) already on the device: the captured LiDAR step (model/train_graph.py
            # splits the blob's rows on the host before the replay)
            gt, self._true_gt_boxes = gt_boxes
        else:
            gt = np.asarray(gt_boxes, dtype=np.float32) if not isinstance(gt_boxes, torch.Tensor) else gt_boxes
        if cfg.NET_TYPE == 'lidar' and not isinstance(gt_boxes, tuple):
            # blobs['gt_boxes'] rows are [xc,yc,zc,l,w,h,ry,cls] in voxel-grid units (minibatch.py:147-167).  The target
            # layers take both forms (proposal_target_layer.py:174-175): the 3-D rows and their axis-aligned BEV
            # rectangles [x1,y1,x2,y2,cls]; the rectangle is the same bbaa_graphics_gems the 3-D anchors go through
            # (which helper the missing network.py used is unpinned).
            gt_np = gt.detach().cpu().numpy() if isinstance(gt, torch.Tensor) else gt
            if gt_np.ndim != 2 or gt_np.shape[1] != 8:
                raise ValueError("LiDAR gt_boxes must be (G, 8) [xc,yc,zc,l,w,h,ry,cls], got %s" % (gt_np.shape,))
            aabb = np.concatenate((bbaa_graphics_gems(gt_np[:, :7]), gt_np[:, 7:8]), 1).astype(np.float32)
            self._true_gt_boxes = torch.from_numpy(np.ascontiguousarray(gt_np, dtype=np.float32)).to(dev)
            gt = aabb
        self._gt_boxes = (torch.from_numpy(np.ascontiguousarray(gt)) if isinstance(gt, np.ndarray) else gt).to(
            dev, dtype=torch.float32)
        # don't-care boxes (minibatch.py:168-176,215-220): only read when cfg.TRAIN.IGNORE_DC.  The anchor target layer's
        # IGNORE_DC branch is a no-op in the reference (it writes -1 into labels that are still all -1,
        # anchor_target_layer.py:58-64), so only the proposal sampling uses them.
        self._gt_boxes_dc = None
        if cfg.TRAIN.IGNORE_DC and gt_boxes_dc is not None and len(gt_boxes_dc) > 0:
            dc = gt_boxes_dc if isinstance(gt_boxes_dc, torch.Tensor) else torch.from_numpy(
                np.ascontiguousarray(np.asarray(gt_boxes_dc, dtype=np.float32)))
            self._gt_boxes_dc = dc.to(dev, dtype=torch.float32)
        out = self._predict()
        self._add_losses()
        return out

    # ------------------------------------------------------------------------------------------
    # inference entry point used by lib/model/test.py:75
    # ------------------------------------------------------------------------------------------
    def frame_pool(self, **kw):
        """The pool of captured frames of this net (model/frame_graph.FramePool), created on first use with
        cfg.TEST.FRAMES_IN_FLIGHT streams; keyword arguments re-create it (tests, bench.py)."""
        from ..model.frame_graph import FramePool
        pool = self.__dict__.get('_frame_pool')
        if pool is None or kw or str(pool.dev) != str(torch.device(self._device)):
            args = dict(streams=int(cfg.TEST.FRAMES_IN_FLIGHT), max_keys=int(cfg.TEST.GRAPH_MAX_SHAPES),
                        autotune=bool(cfg.TEST.GRAPH_AUTOTUNE))
            args.update(kw)
            pool = self.__dict__['_frame_pool'] = FramePool(self, **args)
        return pool

    def enable_frame_graphs(self, enabled=True):
        """``test_frame`` / ``run_eval`` replay the forward pass of a frame as a captured hipGraph (one per frame problem,
        ``frame_pool``) instead of launching ~115 kernels from Python.  ``test_net`` uses the pool regardless
        (cfg.TEST.FRAME_GRAPHS); this switch is for callers of the per-frame API (lib/model/test.py:75)."""
        self._frame_graphs = bool(enabled)

    def test_frame(self, data, info):
        """Returns (cls_score, cls_prob, pred_boxes, rois, uncertainties) as device tensors with exactly
        ``n`` = number of proposals rows (one host sync to read n), like the reference."""
        runner = None
        if getattr(self, '_frame_graphs', False) and cfg.TEST.FRAME_GRAPHS and getattr(self, '_rpn_override', None) is None:
            pool = self.frame_pool()
            pool.sync_weights()
            runner = pool.runner(data.shape, info, with_filter=False)
        if runner is not None:
            runner.run(data)
            # the reference hands out fresh tensors: copies, because the graph's buffers are rewritten by the next frame
            p = self._predictions = dict(runner.predictions)
            self._info, self._mode = runner.info, 'TEST'
            n = int(p['rois_count'].item())
            out = {k: p[k][:n].clone() for k in ('cls_score', 'cls_prob', 'pred_boxes', 'rois')}
            uncertainties = {k: v[:n].clone() for k, v in p.get('uncertainties', {}).items()}
            return out['cls_score'], out['cls_prob'], out['pred_boxes'], out['rois'], uncertainties
        self.forward(data, info, None, None, mode='TEST')
        p = self._predictions
        n = int(p['rois_count'].item())
        uncertainties = {k: v[:n] for k, v in p.get('uncertainties', {}).items()}
        return p['cls_score'][:n], p['cls_prob'][:n], p['pred_boxes'][:n], p['rois'][:n], uncertainties

    def _clip_gradients(self):
        """Per-element clamp to +-cfg.GRAD_MAX_CLIP (lib/model/config.py:338; the ancestor clips by value)."""
        clip = float(cfg.GRAD_MAX_CLIP)
        for prm in self.parameters():
            if prm.grad is not None:
                prm.grad.clamp_(-clip, clip)

    def backward(self, loss):
        """loss.backward() plus the join of the filter-gradient side stream (autograd_ops.ASYNC_WGRAD): afterwards every
        ``param.grad`` is complete as seen from the current stream."""
        try:
            loss.backward()
        except BaseException:
            autograd_ops.drop_deferred_weight_grads()
            raise
        autograd_ops.join_weight_grads(self._device)

    def enable_train_graphs(self, enabled=True, max_graphs=8):
        """Run ``train_step`` as a replayed hipGraph per problem shape (model/train_graph.py): same arithmetic and the same
        random sub-sampling streams as the eager step, without the ~1000 host launches per frame."""
        self._train_graphs = {} if enabled else None
        self._train_graph_max = int(max_graphs)

    def _train_runner(self, blobs):
        graphs = getattr(self, '_train_graphs', None)
        if graphs is None:
            return None
        from ..model import train_graph
        why = train_graph.graphable(self, blobs)
        if why is None:
            data, info = blobs['data'], np.asarray(blobs['info'], dtype=np.float32)
            # one graph per frame geometry; the number of gt boxes only selects the capacity of its gt buffer (32, 64, ...)
            key = (int(data.shape[1]), int(data.shape[2]), int(data.shape[3]), train_graph.gt_capacity(len(blobs['gt_boxes'])),
                   tuple(float(v) for v in info), train_graph.capture_switches())
            runner = graphs.get(key)
            if runner is not None:
                return runner
            if len(graphs) < self._train_graph_max:
                runner = graphs[key] = train_graph.captured_step(self, key[0], key[1], key[2], len(blobs['gt_boxes']), info, blobs)
                return runner
            why = "more than %d distinct frame geometries are held as graphs (enable_train_graphs(max_graphs=...))" % self._train_graph_max
        warned = self.__dict__.setdefault('_train_graph_warned', set())
        if why not in warned:
            warned.add(why)
            import warnings
            warnings.warn("train_step: this frame runs EAGERLY (host-bound, ~1.5x slower), not as a captured step: " + why)
        return None

    def apply_update(self, optimizer, in_place=False):
        """The weight update that ends a pseudo batch (lib/model/train_val.py:379-382 inside the missing network.py's
        train_step): data-parallel average, per-element gradient clip, optimizer step, gradients cleared.  ``in_place``:
        captured training graphs accumulate into the gradient buffers and read the derived filters by ADDRESS, so the
        gradients are zeroed in place and the filters re-derived in place."""
        if hasattr(optimizer, 'reduce'):
            optimizer.reduce()        # data parallel: average over the ranks first, clip the batch gradient after
        self._clip_gradients()
        optimizer.step()
        if in_place:
            optimizer.zero_grad(set_to_none=False)
            from ..model.train_graph import after_optimizer_step
            after_optimizer_step(self)
        else:
            optimizer.zero_grad()

    def train_step(self, blobs, optimizer, update_weights=False):
        """One forward/backward on a frame (lib/model/train_val.py:458).  Gradients accumulate over calls and the
        optimizer steps only when ``update_weights`` (pseudo-batching, train_val.py:379-382).  Returns the loss."""
        graph_mode = getattr(self, '_train_graphs', None) is not None
        runner = self._train_runner(blobs) if graph_mode else None
        if runner is not None:
            loss, counts = runner.run(blobs)
            self._losses = runner.losses
        else:
            self.forward(blobs['data'], blobs['info'], blobs['gt_boxes'], blobs.get('gt_boxes_dc'), mode='TRAIN')
            counts = self._proposal_targets.get('counts') if isinstance(self._proposal_targets, dict) else None
            loss = self._losses['total_loss']
            self.backward(loss)
        # candidate counts are read AFTER the backward pass has been queued: the host wait overlaps the device work instead
        # of stalling between forward and backward
        if counts is not None:
            n_fg, n_bg = [int(v) for v in counts[:2].cpu()]
            if n_fg + n_bg == 0:
                # the reference stops here (pdb.set_trace, proposal_target_layer.py:232-235): no RoI is a foreground or
                # a background candidate (all masked by IGNORE_DC, or every IoU outside both bands).  Data parallel: the
                # fault is recorded in the gradient bucket and EVERY rank raises at the next all-reduce, so nobody is
                # left waiting in the collective.
                if not (hasattr(optimizer, 'mark_fault') and optimizer.mark_fault()):
                    raise RuntimeError(NO_CANDIDATES)
        if update_weights:
            self.apply_update(optimizer, in_place=graph_mode)
        value = float(loss.item())
        self._predictions = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in self._predictions.items()}
        return value

    def train_step_with_summary(self, blobs, optimizer, sum_size, update_weights=False):
        """Same step; the summary list carries the four loss terms instead of tensorboard protobufs."""
        loss = self.train_step(blobs, optimizer, update_weights)
        return loss, [(k, float(v.item())) for k, v in self._losses.items()]

    def run_eval(self, blobs, batch_size, update_summaries=False):
        """Validation forward (lib/model/train_val.py:411-412): returns (summary, rois, roi_labels, cls_prob,
        pred_boxes, uncertainties) — what ``filter_and_draw_prep`` consumes next (train_val.py:416-420).
        ``roi_labels`` = class of the best-overlapping gt box when IoU >= cfg.TRAIN.FG_THRESH, else 0.  The summary
        list carries (name, value) pairs of the detection counts instead of tensorboard protobufs."""
        was_training = self.training
        self.eval()
        try:
            cls_score, cls_prob, pred_boxes, rois, uncertainties = self.test_frame(blobs['data'], blobs['info'])
        finally:
            if was_training:
                self.train()
        gt = blobs.get('gt_boxes')
        roi_labels = torch.zeros((rois.shape[0],), dtype=torch.float32, device=rois.device)
        if gt is not None and len(gt) > 0 and rois.shape[0] > 0 and cfg.NET_TYPE == 'image':
            gt_t = torch.as_tensor(np.asarray(gt, dtype=np.float32)).to(rois.device)
            ov = ops.bbox_overlaps(rois[:, 1:5].contiguous(), gt_t[:, :4].contiguous())
            best, arg = ov.max(1)
            roi_labels = torch.where(best >= cfg.TRAIN.FG_THRESH, gt_t[arg, 4], roi_labels)
        summary = [('val_num_rois', float(rois.shape[0])), ('val_num_fg_rois', float((roi_labels > 0).sum().item()))] \
            if update_summaries else []
        return summary, rois, roi_labels, cls_prob, pred_boxes, uncertainties
