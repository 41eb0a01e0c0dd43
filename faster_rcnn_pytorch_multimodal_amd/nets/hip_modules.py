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


def stable_store(holder, name, key, fresh, refresh=None):
    """Cache ``fresh`` (a tuple of tensors / None) under ``holder.__dict__[name]`` keyed by ``key`` - INTO THE STORAGE OF THE
    PREVIOUS ENTRY when there is one of the same shapes.  Weight-derived tensors (KRSC filters, folded BatchNorm terms,
    transposed and Winograd filters, fused head filters) therefore keep their device addresses across optimizer steps, which
    is what a captured training graph (model/train_graph.py) reads.  ``refresh``: zero-argument callable that re-derives
    this entry from the live parameters (``refresh_derived_weights`` calls it after an optimizer step, since a replayed
    graph runs no Python that would notice the parameters' new version)."""
    d = holder.__dict__
    old = d.get(name)
    value = fresh
    if old is not None and len(old[1]) == len(fresh) and all(
            (o is None) == (f is None) and (o is None or (o.shape == f.shape and o.device == f.device))
            for o, f in zip(old[1], fresh)):
        with torch.no_grad():
            for o, f in zip(old[1], fresh):
                if o is not None and o.data_ptr() != f.data_ptr():
                    o.copy_(f)
        value = old[1]
    d[name] = (key, value)
    if refresh is not None:
        d[name + '_refresh'] = refresh
    return value


# derived entries that read other derived entries (the KRSC filter) come second
_REFRESH_ORDER = ('_frcnn_prepared', '_frcnn_perm', '_fused_cache', '_heads_cache', '_frcnn_wt', '_frcnn_winograd',
                  '_frcnn_dgrad_winograd')


def refresh_derived_weights(net):
    """Re-derive every cached weight-dependent tensor of ``net`` in place (see ``stable_store``).  Returns the number of
    entries visited."""
    holders = [net] + [m for m in net.modules() if m is not net]
    for h in list(holders):
        holders += [v for k, v in h.__dict__.items() if k.endswith('_holder')]
    todo = []
    for h in holders:
        for k, fn in list(h.__dict__.items()):
            if k.endswith('_refresh') and callable(fn):
                rank = next((i for i, tag in enumerate(_REFRESH_ORDER) if tag in k), len(_REFRESH_ORDER))
                todo.append((rank, fn))
    for _, fn in sorted(todo, key=lambda t: t[0]):
        fn()
    return len(todo)


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
    return stable_store(conv, '_frcnn_prepared', key, (w_krsc.contiguous(), scale, shift),
                        refresh=lambda: prepared_conv(conv, bn, use_bn))


def prepared_conv_concat(owner, cache_name, parts):
    """Several convolutions that read the SAME input with the same geometry as ONE: ``parts`` = [(conv, bn, use_bn), ...] ->
    (w_krsc (K1 + K2 + .., R, S, C), scale (K,), shift (K,)) with the folded BatchNorm terms concatenated (1 / 0 where a
    part has none).  Cached on ``owner`` in storage-stable tensors like every derived weight (captured frames read them by
    address); the entry's name must contain '_fused_cache' so that refresh_derived_weights visits it after the parts."""
    prepared = [prepared_conv(c, b, u) for c, b, u in parts]
    key = tuple((t.data_ptr(), t._version) if t is not None else None for p in prepared for t in p)
    cache = owner.__dict__.get(cache_name)
    if cache is not None and cache[0] == key:
        return cache[1]
    with torch.no_grad():
        w = torch.cat([p[0] for p in prepared], 0).contiguous()
        scale = torch.cat([p[1] if p[1] is not None else torch.ones(p[0].shape[0], device=w.device) for p in prepared]).contiguous()
        shift = torch.cat([p[2] if p[2] is not None else torch.zeros(p[0].shape[0], device=w.device) for p in prepared]).contiguous()
    return stable_store(owner, cache_name, key, (w, scale, shift), refresh=lambda: prepared_conv_concat(owner, cache_name, parts))


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
    U (1.78x the filter) is only built for a layer and shape whose plan reads it (``ops.winograd_filter_wanted``: a cached
    Winograd plan, forced Winograd, or a shape the autotuner is about to time); under set_conv_algo(1) or an implicit-GEMM
    plan nothing is transformed.  Once built it is kept for the lifetime of the module (a captured graph may have baked
    its address in).  A cache miss inside a stream capture would launch the transform
    into the graph and pin a graph-pool tensor on the module: it raises instead (run one eager frame first)."""
    k, r, s, c = w_krsc.shape
    n, h, w = nhw
    if not ops.winograd_filter_wanted(n, h, w, c, k, r, s, stride, pad):
        # this SHAPE's plan does not read U.  An entry made for another shape of the same layer stays, with its refresh
        # hook: a captured graph of that shape (FrameRunner / TrainStepRunner) reads U by device address, so it must
        # neither be freed nor fall out of refresh_derived_weights (two frame sizes whose plans differ, e.g. KITTI).
        return None
    return _winograd_entry(conv, w_krsc)


def _winograd_entry(conv, w_krsc):
    """U of ``conv`` for the current contents of ``w_krsc``: the cached tensor, re-derived IN PLACE when the filter changed.
    Also the entry's refresh hook - which therefore does not depend on the plan tables or the algorithm switch at the time
    an optimizer step / load_state_dict is noticed."""
    k, _, _, c = w_krsc.shape
    cache = conv.__dict__.get('_frcnn_winograd')
    key = (w_krsc.data_ptr(), w_krsc._version)
    if cache is not None and cache[0] == key:
        return cache[1][0]
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("Winograd filter of a %dx%dx3x3 layer is not prepared: run an eager frame before capturing" % (k, c))
    return stable_store(conv, '_frcnn_winograd', key, (ops.winograd_filter(w_krsc),),
                        refresh=lambda: _winograd_entry(conv, w_krsc))[0]


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
