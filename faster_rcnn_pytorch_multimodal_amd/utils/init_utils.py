"""Weight initialisers and BatchNorm freeze helpers (reference: lib/utils/init_utils.py:16-77)."""
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
