"""Image detector (ResNet backbone) — counterpart of the reference's lib/nets/imagenet.py.

Channel / stride constants (imagenet.py:30-63): without FPN ``_feat_stride=16``, net_conv has 1024
channels, RoIs are pooled to 7x7x1024 and layer4 lifts them to 2048 (``_fc7_channels``); with FPN the
stride is 4 and every pyramid level has 256 channels.  ``_layers['head']`` = stem + layer1..layer3
(imagenet.py:131-134).  conv1/bn1 and ``layerN`` for N <= cfg.RESNET.FIXED_BLOCKS are frozen, BatchNorm is
frozen and kept in eval mode unless FIXED_BLOCKS == -1 (imagenet.py:96-116, 138-163) — which is what lets
every BatchNorm be folded into its convolution's epilogue on the HIP path.
"""
import torch
import torch.nn as nn

from ..model.config import cfg
from ..utils.init_utils import const_init, normal_init, set_bn_eval, set_bn_fix, set_bn_train, set_bn_var
from .fpn import fpn
from . import uncertainty
from .network import Network


class _Head(nn.Module):
    """stem -> layer1 -> layer2 -> layer3 on NHWC tensors (the reference's nn.Sequential head)."""

    def __init__(self, resnet):
        super().__init__()
        self.stem = resnet.stem()
        self.stages = nn.ModuleList([resnet.layer1, resnet.layer2, resnet.layer3])

    def forward(self, x):
        x = self.stem(x)
        for stage in self.stages:
            x = stage(x)
        return x


class imagenet(Network):
    def __init__(self, num_layers=50):
        Network.__init__(self)
        if cfg.USE_FPN:
            if cfg.POOLING_MODE == 'multiscale':
                self._feat_stride = 4
            self._fpn_en = True
            self._batchnorm_en = True
            self._net_conv_channels = 256
            self._roi_pooling_channels = cfg.POOLING_SIZE * cfg.POOLING_SIZE * self._net_conv_channels
        else:
            self._feat_stride = 16
            self._fpn_en = False
            self._batchnorm_en = True
            self._net_conv_channels = 1024
            self._roi_pooling_channels = 1024
        self._fc7_channels = 2048
        self.inplanes = 64
        self._num_resnet_layers = num_layers
        # imagenet.py:52-63: with MC dropout the detection heads read fc7 / 4 features (nets/uncertainty.py builds them)
        epistemic = bool(cfg.UC.EN_BBOX_EPISTEMIC or cfg.UC.EN_CLS_EPISTEMIC)
        self._det_net_channels = self._fc7_channels // 4 if epistemic else self._fc7_channels
        self._dropout_en = epistemic
        self._cls_drop_rate, self._bbox_drop_rate, self._resnet_drop_rate = uncertainty.drop_rates(lidar=False)

    def init_weights(self):
        # imagenet.py:65-91
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
            uncertainty.init_weights(self, normal_init, const_init, cfg.TRAIN.TRUNCATED, lidar=False)

    def _init_head_tail(self):
        self.resnet = self._build_resnet()
        frozen = [self.resnet.bn1, self.resnet.conv1]
        assert -1 <= cfg.RESNET.FIXED_BLOCKS < 4
        for n in (1, 2, 3):
            if cfg.RESNET.FIXED_BLOCKS >= n:
                frozen.append(getattr(self.resnet, 'layer%d' % n))
        for m in frozen:
            for p in m.parameters():
                p.requires_grad = False
        self.resnet.apply(set_bn_var if cfg.RESNET.FIXED_BLOCKS == -1 else set_bn_fix)
        if cfg.USE_FPN:
            # imagenet.py:119-129: the pyramid is built from c2..c5, so layer4 belongs to the backbone here
            self._fpn = fpn(planes=self._net_conv_channels)
            self._layers['fpn'] = self._fpn
            self._layers['fpn_downsample'] = nn.MaxPool2d(2)      # present upstream; unused on this path
            self._layers['head'] = self.resnet.stem()
            for i in (1, 2, 3, 4):
                self._layers['layer%d' % i] = getattr(self.resnet, 'layer%d' % i)
        else:
            self._layers['head'] = _Head(self.resnet)

    def train(self, mode=True):
        nn.Module.train(self, mode)
        if mode:
            self.resnet.eval()
            for n, attr in ((3, 'layer4'), (2, 'layer3'), (1, 'layer2')):
                if cfg.RESNET.FIXED_BLOCKS <= n:
                    getattr(self.resnet, attr).train()
            if cfg.RESNET.FIXED_BLOCKS <= 0:
                self.resnet.layer1.train()
                self.resnet.conv1.train()
            if cfg.RESNET.FIXED_BLOCKS == -1:
                self.resnet.train()
                self.resnet.apply(set_bn_train)
            else:
                self.resnet.apply(set_bn_eval)
        return self

    def eval(self):
        nn.Module.eval(self)
        if uncertainty.enabled():
            uncertainty.apply_eval_protocol(self)      # dropout modules stay stochastic (imagenet.py:165-172)
        return self

    # ---- checkpoint helpers (imagenet.py:199-244): same key rules as the reference -------------------
    @staticmethod
    def _copy_matching(own_state, items):
        for name, param in items:
            if name not in own_state:
                continue
            if isinstance(param, torch.nn.Parameter):
                param = param.data
            own_state[name].copy_(param)

    def load_pretrained_rpn(self, state_dict):
        self._copy_matching(self.state_dict(), state_dict.items())

    def load_pretrained_full(self, state_dict):
        def wanted(name):
            det_head = ('bbox' in name or 'cls' in name) and 'rpn' not in name
            return not det_head
        self._copy_matching(self.state_dict(), ((k, v) for k, v in state_dict.items() if wanted(k)))

    def load_pretrained_cnn(self, state_dict):
        renamed = ((k if 'resnet' in k else 'resnet.' + k, v) for k, v in state_dict.items())
        self._copy_matching(self.state_dict(), renamed)
