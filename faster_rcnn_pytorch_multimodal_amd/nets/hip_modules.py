"""Glue between ``torch.nn`` parameter containers and the HIP kernels.

``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.Linear`` objects are kept ONLY as owners of parameters so that
``state_dict()`` keys, ``named_parameters()`` and ``.requires_grad`` behave exactly like the reference's
modules (lib/model/train_val.py:188-206 builds its optimizer groups from them).  Their torch ``forward``
is never used on the hot path: ``conv_bn_act`` reads the parameters, lays the filter out as KRSC with the
input channels padded to a multiple of 4, folds an eval-mode BatchNorm into a per-channel scale/shift and
launches ``frcnn_conv2d_fwd``.  Prepared tensors are cached on the module and keyed by the parameters'
version counters, so ``load_state_dict`` / optimizer steps invalidate them automatically.

Activations between modules are NHWC ``(N, H, W, C)`` contiguous tensors.
"""
import torch
import torch.nn as nn

from .. import ops


def pad4(c):
    return (c + 3) // 4 * 4


def _versions(*tensors):
    return tuple((t._version, t.data_ptr()) if t is not None else None for t in tensors)


def prepared_conv(conv, bn=None, use_bn=True):
    """Returns (w_krsc, scale, shift) device tensors for ``conv`` (+ eval-mode ``bn`` folded in)."""
    bn = bn if (bn is not None and use_bn) else None
    key_tensors = [conv.weight, conv.bias]
    if bn is not None:
        key_tensors += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
    # the batch-norm training kernel updates the running statistics through raw pointers (no torch version bump):
    # autograd_ops._BnTrainFn counts those updates on the module instead
    stats_version = bn.__dict__.get('_frcnn_stats_version', 0) if bn is not None else 0
    key = (_versions(*key_tensors), str(conv.weight.device), stats_version)
    cache = conv.__dict__.get('_frcnn_prepared')
    if cache is not None and cache[0] == key:
        return cache[1]
    with torch.no_grad():
        w = conv.weight.detach()
        k, c, r, s = w.shape
        cp = pad4(c)
        w_krsc = torch.zeros((k, r, s, cp), dtype=torch.float32, device=w.device)
        w_krsc[..., :c] = w.permute(0, 2, 3, 1)
        scale = shift = None
        if bn is not None:
            # F.batch_norm(eval): (x - mean) / sqrt(var + eps) * weight + bias  ->  x*scale + shift
            scale = (bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)).contiguous()
            shift = (bn.bias.detach() - bn.running_mean.detach() * scale).contiguous()
            if conv.bias is not None:
                shift = (shift + conv.bias.detach() * scale).contiguous()
        elif conv.bias is not None:
            shift = conv.bias.detach().contiguous()
    prepared = (w_krsc.contiguous(), scale, shift)
    conv.__dict__['_frcnn_prepared'] = (key, prepared)
    return prepared


def conv_bn_act(x, conv, bn=None, relu=False, residual=None, use_bn=True):
    """NHWC in, NHWC out: act(bn(conv(x)) + residual) on the fp32 matrix cores."""
    if bn is not None and use_bn and bn.training:
        raise NotImplementedError("BatchNorm in training mode (batch statistics) is not on the HIP path yet")
    w, scale, shift = prepared_conv(conv, bn, use_bn)
    if x.shape[-1] != w.shape[-1]:
        x = ops.pad_channels(x, w.shape[-1])
    stride = conv.stride[0] if isinstance(conv.stride, (tuple, list)) else conv.stride
    pad = conv.padding[0] if isinstance(conv.padding, (tuple, list)) else conv.padding
    u = None
    if residual is None:
        n, h, wd, c = x.shape
        u = _winograd_filter(conv, w, (n, h, wd), stride, pad)
    return ops.conv2d_nhwc(x, w, scale, shift, residual, stride=stride, pad=pad, relu=relu, w_winograd=u)


def _winograd_filter(conv, w_krsc, nhw, stride, pad):
    """The Winograd-transformed filter U of an eligible 3x3 layer, cached next to the KRSC filter it was made from
    (``prepared_conv`` makes a new one per parameter version), so the inference path does not redo the transform per frame.
    U (1.78x the filter) is only built - and only kept - for a layer whose plan reads it (``ops.winograd_filter_wanted``: a
    cached Winograd plan, forced Winograd, or a shape the autotuner is about to time); under set_conv_algo(1) or an
    implicit-GEMM plan nothing is transformed or held.  A cache miss inside a stream capture would launch the transform
    into the graph and pin a graph-pool tensor on the module: it raises instead (run one eager frame first)."""
    k, r, s, c = w_krsc.shape
    n, h, w = nhw
    if not ops.winograd_filter_wanted(n, h, w, c, k, r, s, stride, pad):
        conv.__dict__.pop('_frcnn_winograd', None)
        return None
    cache = conv.__dict__.get('_frcnn_winograd')
    if cache is not None and cache[0] is w_krsc:
        return cache[1]
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("Winograd filter of a %dx%dx3x3 layer is not prepared: run an eager frame before capturing" % (k, c))
    u = ops.winograd_filter(w_krsc)
    conv.__dict__['_frcnn_winograd'] = (w_krsc, u)
    return u


def to_nhwc(t):
    """NCHW-shaped tensor (any strides) -> contiguous NHWC tensor; free for channels_last inputs."""
    v = t.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def to_nchw_view(t):
    """NHWC contiguous tensor -> NCHW-shaped view (channels_last strides), no copy."""
    return t.permute(0, 3, 1, 2)


class MaxPool3x3s2(nn.Module):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the stem (lib/nets/resnet.py:156), NHWC."""

    def forward(self, x):
        return ops.maxpool3x3s2_nhwc(x)
