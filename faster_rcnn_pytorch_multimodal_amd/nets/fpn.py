"""Feature pyramid top-down path — counterpart of the reference's lib/nets/fpn.py:23-68.

p5 = lat5(c5); p4 = up(p5) + lat4(c4); p3 = aa3(up(p4) + lat3(c3)); p2 = aa2(up(p3) + lat2(c2)) with
``up`` = F.interpolate(size=(H,W), mode='bilinear', align_corners=False).  Only p3 and p2 are smoothed, and p2
is built from the SMOOTHED p3 (fpn.py:56-68); ``aalayer4`` and ``subsample`` exist (state-dict keys) but are
unused, like upstream.  Lateral / smoothing convolutions run on the MFMA implicit-GEMM kernel, the
upsample-add on ``frcnn_upsample_bilinear_add_fwd``; tensors are NHWC.
"""
import torch
import torch.nn as nn

from .. import ops
from ..model.config import cfg
from ..utils.init_utils import normal_init
from .autograd_ops import conv_bn_act_train, upsample_add_train
from .hip_modules import conv_bn_act


class fpn(nn.Module):
    def __init__(self, c2_inplanes=256, c3_inplanes=512, c4_inplanes=1024, c5_inplanes=2048, planes=1024):
        super().__init__()
        self.latlayer2 = nn.Conv2d(c2_inplanes, planes, kernel_size=1, stride=1, padding=0)
        self.latlayer3 = nn.Conv2d(c3_inplanes, planes, kernel_size=1, stride=1, padding=0)
        self.latlayer4 = nn.Conv2d(c4_inplanes, planes, kernel_size=1, stride=1, padding=0)
        self.latlayer5 = nn.Conv2d(c5_inplanes, planes, kernel_size=1, stride=1, padding=0)
        self.aalayer2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1)
        self.aalayer3 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1)
        self.aalayer4 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1)
        self.subsample = nn.AvgPool2d(2, stride=2)

    def init(self):
        for m in (self.latlayer2, self.latlayer3, self.latlayer4, self.latlayer5, self.aalayer2, self.aalayer3,
                  self.aalayer4):
            normal_init(m, 0, 0.01, cfg.TRAIN.TRUNCATED)

    def forward(self, c2, c3, c4, c5):
        grad = torch.is_grad_enabled()
        conv = conv_bn_act_train if grad else conv_bn_act
        up_add = upsample_add_train if grad else ops.upsample_bilinear_add
        p5 = conv(c5, self.latlayer5)
        p4 = up_add(p5, conv(c4, self.latlayer4))
        p3 = conv(up_add(p4, conv(c3, self.latlayer3)), self.aalayer3)
        p2 = conv(up_add(p3, conv(c2, self.latlayer2)), self.aalayer2)
        return p2, p3, p4, p5
