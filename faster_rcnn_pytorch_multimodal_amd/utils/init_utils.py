"""Weight initialisers and BatchNorm freeze helpers (reference: lib/utils/init_utils.py:16-77)."""
import math

import torch
import torch.nn as nn


def normal_init(m, mean, stddev, truncated=False, bias=0.0):
    """N(mean, stddev) weights; ``truncated`` folds the tails with fmod(2) like the reference (:24-36)."""
    if truncated:
        m.weight.data.normal_().fmod_(2).mul_(stddev).add_(mean)
    else:
        m.weight.data.normal_(mean, stddev)
    m.bias.data.fill_(bias)


def uniform_init(m, min_v, max_v, bias=0.0):
    m.weight.data.uniform_(min_v, max_v)
    m.bias.data.fill_(bias)


def const_init(m, weight, bias):
    nn.init.constant_(m.weight, weight)
    nn.init.constant_(m.bias, bias)


def xaiver_init(m, mean, stddev, truncated=False, bias=0.0):
    if truncated:
        m.weight.data.xavier_normal_().fmod_(2).mul_(stddev).add_(mean)
    else:
        nn.init.xavier_normal_(m.weight)
    m.bias.data.fill_(bias)


def _is_bn(m):
    return m.__class__.__name__.find('BatchNorm') != -1


def set_bn_fix(m):
    if _is_bn(m):
        for p in m.parameters():
            p.requires_grad = False


def set_bn_var(m):
    if _is_bn(m):
        for p in m.parameters():
            p.requires_grad = True


def set_bn_eval(m):
    if _is_bn(m):
        m.eval()


def set_bn_train(m):
    if _is_bn(m):
        m.train()


def seeded_state_dict(module, seed, bn_mode="identity", all_backbone=False):
    """Deterministic weights keyed by module NAME (independent of construction order), following the
    reference's init rules: kaiming-normal fan_out for backbone convs (resnet.py:168-173), N(0,0.01) for
    rpn_net / rpn_* / cls_score_net / FPN convs and N(0,0.001) for bbox_pred_net (imagenet.py:65-91,
    fpn.py:47-54), zero biases; BN affine (1,0) with running stats (0,1) for bn_mode='identity', random
    affine + stats for bn_mode='random' (exercises the BN fold), and 'tame' = 'random' with bn3 damped.  all_backbone=True treats every conv as
    a backbone conv (stand-alone ResNet101)."""
    import zlib
    out = {}
    for name, m in module.named_modules():
        g = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 63))
        pre = name + "." if name else ""
        if isinstance(m, nn.Conv2d):
            if all_backbone or name.startswith("resnet."):
                std = math.sqrt(2.0 / (m.out_channels * m.kernel_size[0] * m.kernel_size[1]))
            else:
                std = 0.01
            out[pre + "weight"] = torch.randn(m.weight.shape, generator=g) * std
            if m.bias is not None:
                out[pre + "bias"] = torch.zeros_like(m.bias)
        elif isinstance(m, nn.Linear):
            std = 0.001 if name == "bbox_pred_net" else 0.01
            out[pre + "weight"] = torch.randn(m.weight.shape, generator=g) * std
            out[pre + "bias"] = torch.zeros_like(m.bias)
        elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):      # BatchNorm1d: the LiDAR uncertainty heads
            c = m.num_features
            if bn_mode == "identity":
                out[pre + "weight"], out[pre + "bias"] = torch.ones(c), torch.zeros(c)
                out[pre + "running_mean"], out[pre + "running_var"] = torch.zeros(c), torch.ones(c)
            else:
                # 'tame': like 'random' but the residual-branch BN (bn3) is damped so that activations
                # stay O(input) through the 33 blocks instead of growing to 1e7 (workload)
                damp = 0.25 if (bn_mode == "tame" and name.endswith("bn3")) else 1.0
                out[pre + "weight"] = (torch.rand(c, generator=g) + 0.5) * damp
                out[pre + "bias"] = torch.randn(c, generator=g) * 0.1
                out[pre + "running_mean"] = torch.randn(c, generator=g) * 0.1
                out[pre + "running_var"] = torch.rand(c, generator=g) + 0.5
            out[pre + "num_batches_tracked"] = torch.zeros((), dtype=torch.long)
    return out
