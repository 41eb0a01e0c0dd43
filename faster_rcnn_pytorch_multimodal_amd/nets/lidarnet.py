"""LiDAR-BEV detector (ResNet backbone on a voxelised bird's-eye-view blob) — counterpart of the reference's
lib/nets/lidarnet.py.

Differences from the image detector (lidarnet.py:29-68,104-151): ``conv1`` takes cfg.LIDAR.NUM_CHANNEL = 15
input planes (12 height slices + density + intensity + elongation, lib/roi_data_layer/minibatch.py:459-512);
without FPN ``_batchnorm_en`` is False, so layer4 runs WITHOUT BatchNorm (lib/nets/resnet.py:163-164,103-118);
every BatchNorm is trainable by default (``set_bn_var``, lidarnet.py:110) and blocks are frozen only up to
cfg.RESNET.FIXED_BLOCKS; boxes are 7-DoF [xc,yc,zc,l,w,h,ry] decoded from the RoI and its 3-D anchor.
With cfg.USE_FPN the backbone is the p2..p5 pyramid of the image detector (lidarnet.py:31-40,136-146).
The 15-channel blob is zero-padded to 16 channels on the device so the stem conv reads 16-byte pixels.
"""
import torch
import torch.nn as nn

from ..model.config import cfg
from ..utils.init_utils import const_init, normal_init, set_bn_eval, set_bn_fix, set_bn_train, set_bn_var
from .fpn import fpn
from .imagenet import _Head
from . import uncertainty
from .network import Network


class lidarnet(Network):
    def __init__(self, num_layers=50):
        Network.__init__(self)
        if cfg.USE_FPN:
            # lidarnet.py:31-40: same pyramid as the image detector (p2..p5, 256 channels, RPN on p2 at stride 4,
            # multi-scale pooling, custom tail); layer4 is part of the backbone and keeps its BatchNorm
            if cfg.POOLING_MODE == 'multiscale':
                self._feat_stride = 4
            self._fpn_en = True
            self._batchnorm_en = True
            self._net_conv_channels = 256
            self._roi_pooling_channels = cfg.POOLING_SIZE * cfg.POOLING_SIZE * self._net_conv_channels
        elif cfg.USE_LIDAR_FPN:
            # lidarnet.py:41-46 only sets attributes (_feat_stride = 8, _fpn_en, 1024 channels); _init_head_tail
            # (lidarnet.py:136-150) builds a pyramid for cfg.USE_FPN alone, so with this flag the reference's own
            # subclass still produces the stride-16 layer3 map: whatever made a stride-8 pyramid of it lived in the
            # missing network.py and is not recoverable from the snapshot
            raise NotImplementedError("cfg.USE_LIDAR_FPN: the stride-8 variant is defined only in the reference's missing "
                                      "lib/nets/network.py (lidarnet.py builds no pyramid for it); cfg.USE_FPN is supported")
        else:
            self._feat_stride = 16
            self._fpn_en = False
            self._net_conv_channels = 1024
            self._roi_pooling_channels = 1024
            self._batchnorm_en = False
        self._fc7_channels = 2048
        self.inplanes = 64
        self._num_resnet_layers = num_layers
        # lidarnet.py:56-67: with MC dropout the detection heads read fc7 / 4 features (nets/uncertainty.py builds them)
        epistemic = bool(cfg.UC.EN_BBOX_EPISTEMIC or cfg.UC.EN_CLS_EPISTEMIC)
        self._det_net_channels = self._fc7_channels // 4 if epistemic else self._fc7_channels
        self._dropout_en = epistemic
        self._cls_drop_rate, self._bbox_drop_rate, self._resnet_drop_rate = uncertainty.drop_rates(lidar=True)
        self.num_lidar_channels = cfg.LIDAR.NUM_CHANNEL

    def init_weights(self):
        # lidarnet.py:70-102
        normal_init(self.rpn_net, 0, 0.01, cfg.TRAIN.TRUNCATED)
        if cfg.USE_FPN:
            self._fpn.init()
        if cfg.ENABLE_CUSTOM_TAIL:
            for m in (self.t_fc1, self.t_fc2, self.t_fc3):
                normal_init(m, 0, 0.01, cfg.TRAIN.TRUNCATED)
        normal_init(self.rpn_cls_score_net, 0, 0.01, cfg.TRAIN.TRUNCATED)
        normal_init(self.rpn_bbox_pred_net, 0, 0.01, cfg.TRAIN.TRUNCATED)
        normal_init(self.cls_score_net, 0, 0.01, cfg.TRAIN.TRUNCATED)
        normal_init(self.bbox_pred_net, 0, 0.001, cfg.TRAIN.TRUNCATED)
        if uncertainty.enabled():
            uncertainty.init_weights(self, normal_init, const_init, cfg.TRAIN.TRUNCATED, lidar=True)

    def _init_head_tail(self):
        self.resnet = self._build_resnet()
        # lidarnet.py:107: the stem is rebuilt for the BEV planes (default torch init, not kaiming fan_out)
        self.resnet.conv1 = nn.Conv2d(self.num_lidar_channels, self.inplanes, kernel_size=7, stride=2, padding=3,
                                      bias=False)
        assert -1 <= cfg.RESNET.FIXED_BLOCKS < 4
        self.resnet.apply(set_bn_var)
        for n in (4, 3, 2, 1):
            if cfg.RESNET.FIXED_BLOCKS >= n:
                layer = getattr(self.resnet, 'layer%d' % n)
                layer.apply(set_bn_fix)
                if n == 1:
                    self.resnet.bn1.apply(set_bn_fix)
                for p in layer.parameters():
                    p.requires_grad = False
        if cfg.RESNET.FIXED_BLOCKS >= 0:
            for p in list(self.resnet.bn1.parameters()) + list(self.resnet.conv1.parameters()):
                p.requires_grad = False
        if cfg.USE_FPN:
            # lidarnet.py:136-146
            self._fpn = fpn(planes=self._net_conv_channels)
            self._layers['fpn'] = self._fpn
            self._layers['head'] = self.resnet.stem()
            for i in (1, 2, 3, 4):
                self._layers['layer%d' % i] = getattr(self.resnet, 'layer%d' % i)
        else:
            self._layers['head'] = _Head(self.resnet)

    def train(self, mode=True):
        nn.Module.train(self, mode)
        if mode:
            self.resnet.eval()
            if cfg.RESNET.FIXED_BLOCKS != -1:
                self.resnet.apply(set_bn_eval)
            else:
                self.resnet.train()
                self.resnet.apply(set_bn_train)
            for n in (3, 2, 1):
                if cfg.RESNET.FIXED_BLOCKS <= n:
                    layer = getattr(self.resnet, 'layer%d' % (n + 1))
                    layer.train()
                    layer.apply(set_bn_train)
            if cfg.RESNET.FIXED_BLOCKS <= 0:
                self.resnet.layer1.train()
                self.resnet.conv1.train()
                self.resnet.bn1.apply(set_bn_train)
                self.resnet.layer1.apply(set_bn_train)
        return self

    def eval(self):
        nn.Module.eval(self)
        if uncertainty.enabled():
            uncertainty.apply_eval_protocol(self)      # dropout modules stay stochastic (imagenet.py:165-172)
        return self

    # ---- checkpoint helpers (lidarnet.py:205-246): same key rules as the reference ---------------------
    @staticmethod
    def _copy_filtered(own_state, state_dict, wanted):
        for name, param in state_dict.items():
            if name not in own_state or not wanted(name):
                continue
            if isinstance(param, torch.nn.Parameter):
                param = param.data
            own_state[name].copy_(param)

    def load_pretrained_full(self, state_dict):
        self._copy_filtered(self.state_dict(), state_dict,
                            lambda n: not (('bbox' in n or 'cls' in n) and 'rpn' not in n))

    def load_pretrained_cnn(self, state_dict):
        self._copy_filtered(self.state_dict(), state_dict, lambda n: 'resnet' in n and 'layer4' not in n)

    load_pretrained_rpn = load_pretrained_cnn
