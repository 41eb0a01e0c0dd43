"""ResNet backbone with the caffe stride placement of the reference (lib/nets/resnet.py), executed by the
HIP implicit-GEMM kernel.

Module tree and parameter names equal the reference's so checkpoints load unchanged:
``conv1, bn1, layer1..layer4`` with ``Bottleneck{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}``.
Strides: layer2/3 down-sample on the FIRST 1x1 of their first block (resnet.py:232-234); without FPN
layer4 runs at stride 1 (resnet.py:236-238); with FPN layer4[0] keeps stride 2 on its 3x3.
``batchnorm_en=False`` (LiDAR, non-FPN) removes BN from layer4 only (resnet.py:163-164).
Every conv is fused with its folded BatchNorm, the residual add and the ReLU in one kernel launch.
Tensors between modules are NHWC.
"""
import torch
import torch.nn as nn

from ..model.config import cfg
from .hip_modules import MaxPool3x3s2, conv_bn_act


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, batchnorm_en=True):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.downsample = downsample
        self.batchnorm_en = batchnorm_en

    def _needs_grad(self, x):
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))

    def forward(self, x):
        if self._needs_grad(x):
            from .autograd_ops import bottleneck_train     # one autograd node per block, HIP kernels both ways
            return bottleneck_train(x, self)
        bn = self.batchnorm_en
        if self.downsample is not None:
            identity = conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False)
        else:
            identity = x
        out = conv_bn_act(x, self.conv1, self.bn1, relu=True, use_bn=bn)
        out = conv_bn_act(out, self.conv2, self.bn2, relu=True, use_bn=bn)
        return conv_bn_act(out, self.conv3, self.bn3, relu=True, residual=identity, use_bn=bn)


class Stage(nn.Sequential):
    """A ``layerN``: a chain of Bottlenecks (nn.Sequential so keys read ``layerN.<i>.*``)."""


class Stem(nn.Module):
    """conv1 + bn1 + relu (one kernel) + 3x3/2 max-pool — lib/nets/resnet.py:152-156."""

    def __init__(self, conv1, bn1, maxpool):
        super().__init__()
        self.conv1, self.bn1, self.maxpool = conv1, bn1, maxpool

    def forward(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in list(self.conv1.parameters()) + list(self.bn1.parameters())):
            # cfg.RESNET.FIXED_BLOCKS == -1: the stem trains too (conv1 filter, bn1 on batch statistics)
            from .autograd_ops import conv_bn_act_train, maxpool_train
            return maxpool_train(conv_bn_act_train(x, self.conv1, self.bn1, relu=True))
        return self.maxpool(conv_bn_act(x, self.conv1, self.bn1, relu=True))


class ResNetWrapper(nn.Module):
    def __init__(self, layers, in_channels=3, batchnorm_en=True):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)     # parameter-free; kept so the attribute exists like upstream
        self.maxpool = MaxPool3x3s2()
        self.layer1 = self._make_layer(64, layers[0], 1, True)
        self.layer2 = self._make_layer(128, layers[1], 2, True)
        self.layer3 = self._make_layer(256, layers[2], 2, True)
        self.layer4 = self._make_layer(512, layers[3], 2, batchnorm_en)
        for i in (2, 3):
            first = getattr(self, 'layer%d' % i)[0]
            first.conv1.stride = (2, 2)
            first.conv2.stride = (1, 1)
        if not cfg.USE_FPN:
            self.layer4[0].conv2.stride = (1, 1)
            self.layer4[0].downsample[0].stride = (1, 1)
        self._reset_parameters()

    def _make_layer(self, planes, blocks, stride, batchnorm_en):
        down = None
        if stride != 1 or self.inplanes != planes * Bottleneck.expansion:
            down = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * Bottleneck.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * Bottleneck.expansion))
        chain = [Bottleneck(self.inplanes, planes, stride, down, batchnorm_en)]
        self.inplanes = planes * Bottleneck.expansion
        chain += [Bottleneck(self.inplanes, planes, batchnorm_en=batchnorm_en) for _ in range(1, blocks)]
        return Stage(*chain)

    def _reset_parameters(self):
        # lib/nets/resnet.py:168-173
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def stem(self):
        return Stem(self.conv1, self.bn1, self.maxpool)

    def forward(self, x):
        x = self.stem()(x)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


_DEPTHS = {50: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}


def _resnet(depth, pretrained=False, dropout_en=False, drop_rate=0.0, batchnorm_en=True, in_channels=3):
    if pretrained:
        raise NotImplementedError("no network access: load ImageNet weights with load_pretrained_cnn(state_dict)")
    # dropout_en / drop_rate are accepted like upstream; the reference never forwards them to its blocks
    # (lib/nets/resnet.py:157-164), so backbone dropout is dead code there and absent here.
    return ResNetWrapper(_DEPTHS[depth], in_channels=in_channels, batchnorm_en=batchnorm_en)


def resnet50(pretrained=False, dropout_en=False, drop_rate=0.0, batchnorm_en=True):
    return _resnet(50, pretrained, dropout_en, drop_rate, batchnorm_en)


def resnet101(pretrained=False, dropout_en=False, drop_rate=0.0, batchnorm_en=True):
    return _resnet(101, pretrained, dropout_en, drop_rate, batchnorm_en)


def resnet152(pretrained=False, dropout_en=False, drop_rate=0.0, batchnorm_en=True):
    return _resnet(152, pretrained, dropout_en, drop_rate, batchnorm_en)
