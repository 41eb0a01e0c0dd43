"""Autograd nodes of the training path (``Network.train_step`` -> ``loss.backward()``,
lib/model/train_val.py:458): each node's forward AND backward are libfrcnn_hip.so launches
(``frcnn_conv2d_fwd / _bwd_data / _bwd_weight``, ``frcnn_act_bwd``, ``frcnn_upsample_bilinear_*``,
``frcnn_roi_align_fwd / _bwd``, ``frcnn_rpn_loss``, ``frcnn_det_loss``).  torch.autograd only orders the nodes
and owns the ``.grad`` buffers; gradient accumulation inside a residual block is fused into the data-gradient
kernel's epilogue (``add=``), so a Bottleneck is ONE node.

BatchNorm of the image detector is frozen (lib/nets/imagenet.py:110-116): a per-channel scale/shift of the
convolution output that receives no gradient.  The LiDAR backbone trains its BatchNorm layers with batch statistics
(lib/nets/lidarnet.py:110,152-175): ``_BnTrainFn`` = ``frcnn_bn_train_fwd / _bwd``.  Activations are NHWC; parameters keep the reference's layouts
(Conv2d (K,C,R,S), Linear (out,in)) and are re-laid out as KRSC through ``hip_modules.prepared_conv`` caches.
"""
import os

import torch

from .. import ops
from .hip_modules import _winograd_filter, pad4, prepared_conv, stable_store


# Filter gradients are off the critical path of backward (only the data gradient feeds the next node), so they CAN
# run on a side HIP stream and be accumulated into ``param.grad`` there; ``join_weight_grads()`` (called by
# Network.backward before clipping / the optimizer) makes the main stream wait for them.  Measured on the FPN train
# step (round 1): no gain — the step is bound by GPU throughput, not by the dependency chain — so the default keeps
# the plain autograd flow.
ASYNC_WGRAD = False
_SIDE = {}
_KEEP = []      # operands of side-stream launches, held until the main stream has joined the side stream


# True (default): filter gradients run on a side stream next to the data-gradient chain (a captured step is a forked graph).
# False: in line on the step's own stream - a captured step is then ONE chain of kernel nodes, the form the runtime overlaps
# when several replays are in flight on different streams (model/train_graph.TrainPipeline: two replays of a forked graph were
# measured not to overlap at all).  ONLY with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment before the HIP runtime starts:
# with ROCm 7's packet-captured replay of single-chain graphs this step's graph adds wrong filter gradients from its second
# replay on (train_graph.inline_graphs_supported / TrainPipeline's replay check guard it).
WGRAD_ON_SIDE_STREAM = True


# Side streams per device.  Every parameter is pinned to ONE of them (first come, round robin): launches that accumulate into the
# same gradient buffer - the RPN head applied to five pyramid levels - stay ordered, launches of different parameters may overlap.
WGRAD_SIDE_STREAMS = 1
# captured steps: the BatchNorm backward adds d_gamma / d_beta into param.grad itself (False: temporaries + add_ launches; A/B)
BN_GRADS_IN_KERNEL = os.environ.get('FRCNN_BN_GRADS_IN_KERNEL', '1') != '0'
_STREAM_OF = {}    # (device, id of the gradient's owner) -> index


def _side_stream(device, owner=None):
    if not WGRAD_ON_SIDE_STREAM:
        return torch.cuda.current_stream(device)
    key = str(device)
    pool = _SIDE.setdefault(key, [])
    want = max(1, int(WGRAD_SIDE_STREAMS))
    while len(pool) < want:
        pool.append(torch.cuda.Stream(device=device))
    if want == 1 or owner is None:
        return pool[0]
    idx = _STREAM_OF.get((key, owner))
    if idx is None:
        idx = _STREAM_OF[(key, owner)] = len(_STREAM_OF) % want
    return pool[idx % want]


# Deferred filter gradients (ASYNC_WGRAD and GROUP_WGRAD): repeated layers of identical shape - the 22 equal Bottlenecks of
# layer3 - are collected per shape and launched TOGETHER (ops.conv2d_bwd_weight_acc_grouped): one launch pair per shape instead
# of a pixel-split launch + a slab reduction per layer.  A group is flushed to the side stream when it is full, when its shape
# has not come up for _STALE calls (its stage's backward is over), and at join_weight_grads().
# OFF for the eager step; ON in every captured step since round 5 (model/train_graph.TrainStepRunner).  The grouped launches
# are far more efficient (one unsplit launch per shape and stage) but run AFTER their stage's data-gradient chain instead of
# next to it: with the register-staged kernel and its accumulation kernel they lost (15.3-15.4 against 15.0-15.2 ms per replayed
# step, round 3); with conv_wgrad_dma_f32, which adds into param.grad itself, they win (14.7 -> 14.3 ms, round 5).
GROUP_WGRAD = False
# {id(BatchNorm module): (mean buffer, variance buffer)} while a training step with DEFERRED running statistics is captured
# (model/train_graph.TrainStepRunner(defer_bn_stats=True)), else None: see _BnTrainFn.forward
BN_STAT_SINK = None
GROUP_WGRAD_SIZE = 24  # layers per grouped launch (<= ops.WGRAD_MAX_GROUPS): a whole ResNet stage
_DEFER = {}          # key -> list of (x, d_conv, grad)
_DEFER_AGE = {}      # key -> _wgrad calls since the key last came up
_STALE = 8


def _flush_group(key):
    entries = _DEFER.pop(key, [])
    _DEFER_AGE.pop(key, None)
    if not entries:
        return
    x0 = entries[0][0]
    r, s, stride, pad = key[2:6]
    main = torch.cuda.current_stream(x0.device)
    side = _side_stream(x0.device, id(entries[0][2]))
    if WGRAD_ON_SIDE_STREAM:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        if len(entries) == 1:
            ops.conv2d_bwd_weight_acc(entries[0][0], entries[0][1], r, s, entries[0][2], None, stride=stride, pad=pad)
        else:
            ops.conv2d_bwd_weight_acc_grouped([e[0] for e in entries], [e[1] for e in entries], r, s, [e[2] for e in entries],
                                              stride=stride, pad=pad)
    for x, d, _ in entries:
        _KEEP.append((x, d))
        if WGRAD_ON_SIDE_STREAM:
            x.record_stream(side)
            d.record_stream(side)


def _defer_wgrad(x, d_conv, r, s, stride, pad, grad):
    key = (tuple(x.shape), tuple(d_conv.shape), r, s, stride, pad, tuple(grad.shape), str(x.device))
    for k in list(_DEFER_AGE):
        _DEFER_AGE[k] += 1
    _DEFER.setdefault(key, []).append((x, d_conv, grad))
    _DEFER_AGE[key] = 0
    if len(_DEFER[key]) >= min(GROUP_WGRAD_SIZE, ops.WGRAD_MAX_GROUPS):
        _flush_group(key)
    for k in [k for k, age in _DEFER_AGE.items() if age >= _STALE]:
        _flush_group(k)


def drop_deferred_weight_grads():
    """Forget collected-but-unlaunched filter gradients (a backward pass that raised must not leak its tensors into the
    next step's groups)."""
    _DEFER.clear()
    _DEFER_AGE.clear()


def join_weight_grads(device=None):
    """Main stream waits for every filter-gradient launch issued so far (no-op when none were issued); deferred groups are
    launched first."""
    for k in list(_DEFER):
        if device is None or k[-1] == str(torch.device(device)):
            _flush_group(k)
    for key, pool in _SIDE.items():
        if device is None or key == str(torch.device(device)):
            for side in pool:
                torch.cuda.current_stream(side.device).wait_stream(side)
    _KEEP.clear()


def _accumulate(param, grad):
    if param.grad is None:
        param.grad = grad
    else:
        param.grad.add_(grad)


def _wgrad(x, d_conv, r, s, stride, pad, targets):
    """Filter (and bias) gradient of one convolution.  ``targets`` = list of (param, fn) where fn maps
    (dw_krsc, db) to that parameter's gradient.  Synchronous mode returns the list of gradients; asynchronous mode
    launches on the side stream, accumulates into param.grad there and returns None for each."""
    want_bias = any(kind == 'b' for _, _, kind in targets)
    if not ASYNC_WGRAD:
        dw_krsc, db = ops.conv2d_bwd_weight(x, d_conv, r, s, stride=stride, pad=pad, want_bias=want_bias)
        return [fn(dw_krsc, db) for _, fn, _ in targets]
    kinds = [kind for _, _, kind in targets]
    direct = (kinds in (['w'], ['w', 'b']) and all(p.grad is not None and p.grad.is_contiguous() for p, _, _ in targets)
              and targets[0][0].dim() in (2, 4) and targets[0][0].numel() == d_conv.shape[-1] * targets[0][0].shape[1] * r * s)
    if GROUP_WGRAD and direct and kinds == ['w'] and targets[0][0].dim() == 4:
        _defer_wgrad(x, d_conv, r, s, stride, pad, targets[0][0].grad)
        return [None]
    main = torch.cuda.current_stream(x.device)
    side = _side_stream(x.device, id(targets[0][0]))
    if WGRAD_ON_SIDE_STREAM:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        # plain Conv2d / Linear targets ('w' [+ 'b'] of one module, gradient buffers in place): one launch chain sums the
        # pixel-split slabs, changes the layout and adds into param.grad - no temporaries, no permute copy, no add_
        if direct:
            ops.conv2d_bwd_weight_acc(x, d_conv, r, s, targets[0][0].grad, targets[1][0].grad if len(targets) > 1 else None,
                                      stride=stride, pad=pad)
        else:
            dw_krsc, db = ops.conv2d_bwd_weight(x, d_conv, r, s, stride=stride, pad=pad, want_bias=want_bias)
            for param, fn, _ in targets:
                _accumulate(param, fn(dw_krsc, db))
    # the side stream still reads x / d_conv when this function's caller drops them, and the allocator would hand their
    # blocks to the next main-stream allocation.  Tensor.record_stream covers that in eager mode; inside a stream capture it
    # was observed not to (run-to-run different gradients from the replayed graph), so the operands are simply kept alive
    # until join_weight_grads() has made the main stream wait for the side stream.
    _KEEP.append((x, d_conv))
    if WGRAD_ON_SIDE_STREAM:
        x.record_stream(side)
        d_conv.record_stream(side)
    return [None for _ in targets]


def _transposed_filter(conv_like, w_krsc):
    """Cached (C,R,S,K) flipped filter of the data-gradient convolution, keyed like the forward filter."""
    cache = conv_like.__dict__.get('_frcnn_wt')
    key = (w_krsc.data_ptr(), w_krsc._version, tuple(w_krsc.shape))
    if cache is not None and cache[0] == key:
        return cache[1][0]
    return stable_store(conv_like, '_frcnn_wt', key, (ops.conv2d_transpose_filter(w_krsc),),
                        refresh=lambda: _transposed_filter(conv_like, w_krsc))[0]


FUSE_ACT_BWD = True             # False: a separate frcnn_act_bwd pass after each data gradient inside a Bottleneck
DGRAD_WINOGRAD_CACHE = True     # False: the data-gradient convolution transforms its filter on every call


def _dgrad_winograd(conv_like, w_t, x_shape, stride, pad):
    """Winograd transform of the data-gradient filter ``w_t`` (C,3,3,K), cached next to it: the weights change once per
    optimizer step, the data gradient runs once per frame.  None when that convolution's plan is not a Winograd plan (see
    hip_modules._winograd_filter for the rules; a miss inside a stream capture raises)."""
    c, r, s, k = w_t.shape
    if not DGRAD_WINOGRAD_CACHE or r != 3 or stride != 1:
        return None
    if not ops.dgrad_winograd_wanted(x_shape, k, r, s, stride, pad):
        return None          # an entry made for another shape stays (captured graphs read it by address), see _winograd_filter
    return _dgrad_winograd_entry(conv_like, w_t)


def _dgrad_winograd_entry(conv_like, w_t):
    """Cached Winograd form of ``w_t``, re-derived in place when ``w_t`` changed; also the entry's refresh hook."""
    c, _, _, k = w_t.shape
    cache = conv_like.__dict__.get('_frcnn_dgrad_winograd')
    key = (w_t.data_ptr(), w_t._version)
    if cache is not None and cache[0] == key:
        return cache[1][0]
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("Winograd data-gradient filter of a %dx%dx3x3 layer is not prepared: run an eager step before capturing"
                           % (k, c))
    return stable_store(conv_like, '_frcnn_dgrad_winograd', key, (ops.winograd_filter(w_t),),
                        refresh=lambda: _dgrad_winograd_entry(conv_like, w_t))[0]


def _param_grad_from_krsc(dw_krsc, param):
    """(K,R,S,Cpad) -> the parameter's own layout ((K,C,R,S) conv, (out,in) linear)."""
    k, r, s, _ = dw_krsc.shape
    if param.dim() == 2:
        return dw_krsc.view(k, -1)[:, :param.shape[1]].contiguous()
    return dw_krsc[..., :param.shape[1]].permute(0, 3, 1, 2).contiguous()


def _stride_pad(conv):
    st = conv.stride[0] if isinstance(conv.stride, (tuple, list)) else conv.stride
    pd = conv.padding[0] if isinstance(conv.padding, (tuple, list)) else conv.padding
    return st, pd


class _ConvFn(torch.autograd.Function):
    """y = act(conv(x, w) * scale + shift [+ residual]) for ONE parameter set (weight [, bias])."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, pack):
        w_krsc, scale, shift, stride, pad, relu, owner = pack
        u = None
        if residual is None and w_krsc.shape[1] == 3 and isinstance(owner, torch.nn.Module):
            u = _winograd_filter(owner, w_krsc, tuple(x.shape[:3]), stride, pad)
        y = ops.conv2d_nhwc(x, w_krsc, scale, shift, residual, stride=stride, pad=pad, relu=relu, w_winograd=u)
        ctx.pack = pack
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, y if relu else None, weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, bias = ctx.saved_tensors
        w_krsc, scale, _, stride, pad, relu, owner = ctx.pack
        dy = dy.contiguous()
        need_x, need_res, need_w, need_b = ctx.needs_input_grad[:4]
        if relu or scale is not None or (ctx.has_res and need_res):
            d_conv, d_res = ops.act_bwd(dy, y, scale, relu=relu, want_res=ctx.has_res and need_res)
        else:
            d_conv, d_res = dy, None
        dx = dw = db = None
        r, s = w_krsc.shape[1], w_krsc.shape[2]
        targets = []
        if need_w:
            targets.append((weight, lambda dwk, dbk, w=weight: _param_grad_from_krsc(dwk, w), 'w'))
        if bias is not None and need_b:
            targets.append((bias, lambda dwk, dbk: dbk, 'b'))
        if targets:
            grads = _wgrad(x, d_conv, r, s, stride, pad, targets)
            if need_w:
                dw = grads[0]
            if bias is not None and need_b:
                db = grads[-1]
        if need_x:
            w_t = _transposed_filter(owner, w_krsc)
            dx = ops.conv2d_bwd_data(d_conv, w_t, tuple(x.shape), stride=stride, pad=pad,
                                     w_winograd=_dgrad_winograd(owner, w_t, tuple(x.shape), stride, pad))
        return dx, d_res, dw, db, None


class _BnTrainFn(torch.autograd.Function):
    """act(batch_norm(y, batch statistics) [+ residual]) - frcnn_bn_train_fwd / _bwd.  The module's running statistics
    are updated in place by the forward launch (F.batch_norm(training=True))."""

    @staticmethod
    def forward(ctx, y, residual, gamma, beta, bn, relu):
        track = bn.track_running_stats and bn.running_mean is not None
        momentum = bn.momentum
        sink = BN_STAT_SINK.get(id(bn)) if (track and BN_STAT_SINK is not None) else None
        if sink is not None:
            # deferred statistics (model/train_graph.TrainPipeline): this launch leaves the frame's batch mean / unbiased
            # variance in the slot's private buffers - momentum 1 turns the kernel's update (1 - m) * old + m * stat into
            # 0 * old + stat - and the running statistics are folded afterwards, in frame order
            run_mean, run_var, momentum = sink[0], sink[1], 1.0
            sink[2] = True          # this module really runs in the captured step (layer4's BatchNorms of the LiDAR net do not)
        else:
            run_mean, run_var = (bn.running_mean, bn.running_var) if track else (None, None)
            if track and bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
                if momentum is None:                       # cumulative moving average (torch.nn.modules.batchnorm)
                    momentum = 1.0 / float(bn.num_batches_tracked)
        out, mean, invstd = ops.bn_train_fwd(y, gamma.detach() if gamma is not None else None,
                                             beta.detach() if beta is not None else None, bn.eps, momentum or 0.0,
                                             run_mean, run_var, residual, relu)
        if track and sink is None:
            bn.__dict__['_frcnn_stats_version'] = bn.__dict__.get('_frcnn_stats_version', 0) + 1
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.affine = (gamma, beta)
        ctx.save_for_backward(y, out if relu else None, gamma, mean, invstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, out, gamma, mean, invstd = ctx.saved_tensors
        wprm, bprm = ctx.affine
        if (ASYNC_WGRAD and BN_GRADS_IN_KERNEL and ctx.needs_input_grad[2] and ctx.needs_input_grad[3] and wprm is not None and bprm is not None
                and wprm.grad is not None and bprm.grad is not None and wprm.grad.is_contiguous() and bprm.grad.is_contiguous()):
            # captured steps: d_gamma / d_beta are added into the parameters' gradient buffers by the BatchNorm backward's own
            # statistics pass (no temporaries, no add_ launches: 168 of the LiDAR step's launches)
            dy, dres, _, _ = ops.bn_train_bwd(dout.contiguous(), out, y, gamma.detach() if gamma is not None else None, mean, invstd,
                                              relu=ctx.relu, want_res=ctx.has_res and ctx.needs_input_grad[1],
                                              grad_gamma=wprm.grad.view(-1), grad_beta=bprm.grad.view(-1))
            return dy if ctx.needs_input_grad[0] else None, dres, None, None, None, None
        dy, dres, dgamma, dbeta = ops.bn_train_bwd(dout.contiguous(), out, y, gamma.detach() if gamma is not None else None,
                                                   mean, invstd, relu=ctx.relu,
                                                   want_res=ctx.has_res and ctx.needs_input_grad[1])
        if ASYNC_WGRAD:
            # captured steps (model/train_graph.py) accumulate parameter gradients themselves, in place, on the step's stream:
            # autograd's AccumulateGrad nodes run on the parameters' creation stream, outside the capture
            with torch.no_grad():
                for prm, g, need in ((ctx.affine[0], dgamma, ctx.needs_input_grad[2]), (ctx.affine[1], dbeta, ctx.needs_input_grad[3])):
                    if need and prm is not None and g is not None:
                        _accumulate(prm, g.view_as(prm))
            dgamma = dbeta = None
        return (dy if ctx.needs_input_grad[0] else None, dres, dgamma if ctx.needs_input_grad[2] else None,
                dbeta if ctx.needs_input_grad[3] else None, None, None)


def conv_bn_act_train(x, conv, bn=None, relu=False, residual=None, use_bn=True):
    """Differentiable counterpart of hip_modules.conv_bn_act.  An eval-mode (frozen) BatchNorm is folded into the
    convolution's epilogue; a BatchNorm in train() mode (LiDAR backbone, lib/nets/lidarnet.py:152-175) runs as its own
    node on the raw convolution output with batch statistics."""
    stride, pad = _stride_pad(conv)
    if bn is not None and use_bn and bn.training:
        w_krsc, _, shift = prepared_conv(conv, None, False)
        if x.shape[-1] != w_krsc.shape[-1]:
            raise NotImplementedError("differentiable conv needs C % 4 == 0 inputs (only the frozen stem pads its input)")
        y = _ConvFn.apply(x, None, conv.weight, conv.bias, (w_krsc, None, shift, stride, pad, False, conv))
        return _BnTrainFn.apply(y, residual, bn.weight, bn.bias, bn, relu)
    if bn is not None and use_bn and any(p.requires_grad for p in bn.parameters()):
        raise NotImplementedError("trainable BatchNorm affine parameters with frozen (eval-mode) statistics are not on "
                                  "the HIP path: the reference either freezes both or trains both")
    w_krsc, scale, shift = prepared_conv(conv, bn, use_bn)
    if x.shape[-1] != w_krsc.shape[-1]:
        raise NotImplementedError("differentiable conv needs C % 4 == 0 inputs (only the frozen stem pads its input)")
    # with a folded BN the effective bias is shift = bn.bias - mean*scale (+ conv.bias*scale); conv.bias itself only
    # exists on the BN-free convolutions (RPN, FPN), where shift == conv.bias and d(bias) = sum(d_conv)
    return _ConvFn.apply(x, residual, conv.weight, conv.bias, (w_krsc, scale, shift, stride, pad, relu, conv))


def linear_train(x2d, lin, relu=False, weight_nhwc_from=None):
    """act(x W^T + b) as a 1x1 convolution over R "pixels".  ``weight_nhwc_from=(C,P)`` says the Linear consumes an
    NCHW-flattened (C,P,P) map while ``x2d`` is the NHWC flattening: the filter columns are permuted once (cached)."""
    r, cin = x2d.shape
    if weight_nhwc_from is None:
        w_krsc = lin.weight.detach().view(lin.out_features, 1, 1, cin)
        return _ConvFn.apply(x2d.view(r, 1, 1, cin), None, lin.weight, lin.bias,
                             (w_krsc, None, lin.bias.detach(), 1, 0, relu, lin)).view(r, -1)
    return _PermutedLinearFn.apply(x2d, lin.weight, lin.bias, lin, weight_nhwc_from, relu)


class _PermutedLinearFn(torch.autograd.Function):
    @staticmethod
    def _prepared(lin, c, p):
        key = (lin.weight._version, lin.weight.data_ptr())
        cache = lin.__dict__.get('_frcnn_perm')
        if cache is not None and cache[0] == key:
            return cache[1][0]
        w = lin.weight.detach().view(lin.out_features, c, p, p).permute(0, 2, 3, 1).reshape(lin.out_features, 1, 1, -1)
        return stable_store(lin, '_frcnn_perm', key, (w.contiguous(),),
                            refresh=lambda: _PermutedLinearFn._prepared(lin, c, p))[0]

    @staticmethod
    def forward(ctx, x2d, weight, bias, lin, cp, relu):
        c, p = cp
        w_krsc = _PermutedLinearFn._prepared(lin, c, p)
        r = x2d.shape[0]
        y = ops.conv2d_nhwc(x2d.view(r, 1, 1, -1), w_krsc, None, bias.detach(), None, relu=relu)
        ctx.meta = (lin, cp, relu, w_krsc)
        ctx.save_for_backward(x2d, y if relu else None)
        return y.view(r, -1)

    @staticmethod
    def backward(ctx, dy):
        x2d, y = ctx.saved_tensors
        lin, (c, p), relu, w_krsc = ctx.meta
        r = x2d.shape[0]
        dy4 = dy.contiguous().view(r, 1, 1, -1)
        d_conv = ops.act_bwd(dy4, y, None, relu=True)[0] if relu else dy4
        x4 = x2d.view(r, 1, 1, -1)
        to_w = lambda dwk, dbk: dwk.view(lin.out_features, p, p, c).permute(0, 3, 1, 2).reshape(lin.out_features, -1)
        dw, db = _wgrad(x4, d_conv, 1, 1, 1, 0, [(lin.weight, to_w, 'w_perm'), (lin.bias, lambda dwk, dbk: dbk, 'b')])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_bwd_data(d_conv, _transposed_filter(lin, w_krsc), tuple(x4.shape)).view(r, -1)
        return dx, dw, db, None, None, None


class _FusedHeadFn(torch.autograd.Function):
    """Two sibling layers that share their input (rpn_cls_score_net + rpn_bbox_pred_net, cls_score_net +
    bbox_pred_net) as ONE convolution with the filters concatenated along K and zero-padded to a multiple of 4."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, pack):
        w_krsc, bias_cat, owner = pack
        y = ops.conv2d_nhwc(x, w_krsc, None, bias_cat, None)
        ctx.pack = pack
        ctx.biases = (b1, b2)
        ctx.save_for_backward(x, w1, w2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w1, w2 = ctx.saved_tensors
        w_krsc, _, owner = ctx.pack
        dy = dy.contiguous()
        k1, k2 = w1.shape[0], w2.shape[0]
        b1, b2 = ctx.biases
        dw1, db1, dw2, db2 = _wgrad(x, dy, w_krsc.shape[1], w_krsc.shape[2], 1, 0, [
            (w1, lambda dwk, dbk: _param_grad_from_krsc(dwk[:k1], w1), 'w_part'),
            (b1, lambda dwk, dbk: dbk[:k1].contiguous(), 'b'),
            (w2, lambda dwk, dbk: _param_grad_from_krsc(dwk[k1:k1 + k2], w2), 'w_part'),
            (b2, lambda dwk, dbk: dbk[k1:k1 + k2].contiguous(), 'b')])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_bwd_data(dy, _transposed_filter(owner, w_krsc), tuple(x.shape))
        return dx, dw1, db1, dw2, db2, None


def fused_head_weights(owner, m1, m2, cache_name):
    """(w_krsc (K1+K2 padded to %4, 1, 1, C), bias) of two 1x1-conv / Linear siblings, cached on ``owner``."""
    tensors = (m1.weight, m1.bias, m2.weight, m2.bias)
    key = tuple((t._version, t.data_ptr()) for t in tensors)
    cache = owner.__dict__.get(cache_name)
    if cache is not None and cache[0] == key:
        return cache[1]
    with torch.no_grad():
        w1 = m1.weight.detach().reshape(m1.weight.shape[0], -1)
        w2 = m2.weight.detach().reshape(m2.weight.shape[0], -1)
        k = w1.shape[0] + w2.shape[0]
        kp = pad4(k)
        w = torch.zeros((kp, w1.shape[1]), dtype=torch.float32, device=w1.device)
        w[:w1.shape[0]] = w1
        w[w1.shape[0]:k] = w2
        b = torch.zeros((kp,), dtype=torch.float32, device=w1.device)
        b[:w1.shape[0]] = m1.bias.detach()
        b[w1.shape[0]:k] = m2.bias.detach()
        w = w.view(kp, 1, 1, -1).contiguous()
    return stable_store(owner, cache_name, key, (w, b), refresh=lambda: fused_head_weights(owner, m1, m2, cache_name))


def fused_head_train(x, owner, m1, m2, cache_name):
    w, b = fused_head_weights(owner, m1, m2, cache_name)
    return _FusedHeadFn.apply(x, m1.weight, m1.bias, m2.weight, m2.bias, (w, b, _Holder.of(owner, cache_name)))


class _Holder(object):
    """Small attribute bag that owns the transposed-filter cache of a fused head."""
    @staticmethod
    def of(owner, name):
        h = owner.__dict__.get(name + '_holder')
        if h is None:
            h = _Holder()
            owner.__dict__[name + '_holder'] = h
        return h


class _BottleneckFn(torch.autograd.Function):
    """lib/nets/resnet.py:98-127 as one node: three (four with the projection shortcut) fused conv launches forward;
    backward walks them in reverse and accumulates the two gradients of the block input inside the last
    data-gradient launch."""

    @staticmethod
    def forward(ctx, x, block, use_bn, *weights):
        p1 = prepared_conv(block.conv1, block.bn1, use_bn)
        p2 = prepared_conv(block.conv2, block.bn2, use_bn)
        p3 = prepared_conv(block.conv3, block.bn3, use_bn)
        s1, _ = _stride_pad(block.conv1)
        s2, _ = _stride_pad(block.conv2)
        if block.downsample is not None:
            pd = prepared_conv(block.downsample[0], block.downsample[1], True)
            sd, _ = _stride_pad(block.downsample[0])
            identity = ops.conv2d_nhwc(x, pd[0], pd[1], pd[2], None, stride=sd, pad=0, relu=False)
        else:
            pd, sd, identity = None, 1, x
        o1 = ops.conv2d_nhwc(x, p1[0], p1[1], p1[2], None, stride=s1, pad=0, relu=True)
        # the 3x3 layer with its cached Winograd-transformed filter when its tuned plan reads one (re-derived only when the
        # weights change, i.e. once per optimizer step instead of inside every call)
        u2 = _winograd_filter(block.conv2, p2[0], tuple(o1.shape[:3]), s2, 1)
        o2 = ops.conv2d_nhwc(o1, p2[0], p2[1], p2[2], None, stride=s2, pad=1, relu=True, w_winograd=u2)
        out = ops.conv2d_nhwc(o2, p3[0], p3[1], p3[2], identity, stride=1, pad=0, relu=True)
        ctx.block = block
        ctx.meta = (p1, p2, p3, pd, s1, s2, sd)
        ctx.save_for_backward(x, o1, o2, out)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, o1, o2, out = ctx.saved_tensors
        blk = ctx.block
        p1, p2, p3, pd, s1, s2, sd = ctx.meta
        need_w = [w.requires_grad for w in (blk.conv1.weight, blk.conv2.weight, blk.conv3.weight)]

        def wg(inp, dz, conv, r, stride, pad):
            w = conv.weight
            return _wgrad(inp, dz, r, r, stride, pad, [(w, lambda dwk, dbk: _param_grad_from_krsc(dwk, w), 'w')])[0]

        dz3, d_id = ops.act_bwd(dy.contiguous(), out, p3[1], relu=True, want_res=True)
        dw3 = wg(o2, dz3, blk.conv3, 1, 1, 0) if need_w[2] else None
        # the ReLU / folded-BatchNorm backward of conv2 (conv1) is applied by the data-gradient launch of conv3 (conv2) as it
        # stores its result (frcnn_conv2d_bwd_data_act): no separate pass over d_o2 / d_o1
        w3_t = _transposed_filter(blk.conv3, p3[0])
        w2_t = _transposed_filter(blk.conv2, p2[0])
        u2_t = _dgrad_winograd(blk.conv2, w2_t, tuple(o1.shape), s2, 1)
        if FUSE_ACT_BWD:
            dz2 = ops.conv2d_bwd_data(dz3, w3_t, tuple(o2.shape), act_y=o2, act_scale=p2[1])
        else:
            dz2, _ = ops.act_bwd(ops.conv2d_bwd_data(dz3, w3_t, tuple(o2.shape)), o2, p2[1], relu=True)
        dw2 = wg(o1, dz2, blk.conv2, 3, s2, 1) if need_w[1] else None
        if FUSE_ACT_BWD:
            dz1 = ops.conv2d_bwd_data(dz2, w2_t, tuple(o1.shape), stride=s2, pad=1, w_winograd=u2_t, act_y=o1, act_scale=p1[1])
        else:
            dz1, _ = ops.act_bwd(ops.conv2d_bwd_data(dz2, w2_t, tuple(o1.shape), stride=s2, pad=1, w_winograd=u2_t), o1, p1[1],
                                 relu=True)
        dw1 = wg(x, dz1, blk.conv1, 1, s1, 0) if need_w[0] else None
        dwd = None
        dx = None
        if pd is not None:
            dzd, _ = ops.act_bwd(d_id, None, pd[1], relu=False)
            if blk.downsample[0].weight.requires_grad:
                dwd = wg(x, dzd, blk.downsample[0], 1, sd, 0)
            if ctx.needs_input_grad[0]:
                dx = ops.conv2d_bwd_data(dzd, _transposed_filter(blk.downsample[0], pd[0]), tuple(x.shape), stride=sd)
                dx = ops.conv2d_bwd_data(dz1, _transposed_filter(blk.conv1, p1[0]), tuple(x.shape), stride=s1, add=dx)
        elif ctx.needs_input_grad[0]:
            dx = ops.conv2d_bwd_data(dz1, _transposed_filter(blk.conv1, p1[0]), tuple(x.shape), stride=s1, add=d_id)
        grads = [dw1, dw2, dw3]     # parameter-layout gradients, or None when they were accumulated on the side stream
        if pd is not None:
            grads.append(dwd)
        return (dx, None, None) + tuple(grads)


def bottleneck_train(x, block):
    """Differentiable Bottleneck.  With every BatchNorm frozen (image detector) or absent (LiDAR layer4) the block is
    ONE autograd node; a block holding a BatchNorm in train() mode is the chain conv -> batch-norm node per layer."""
    bn_on = block.batchnorm_en
    live = [bn for bn in (block.bn1, block.bn2, block.bn3) if bn_on and bn.training]
    if block.downsample is not None and block.downsample[1].training:
        live.append(block.downsample[1])
    if live:
        if block.downsample is not None:
            identity = conv_bn_act_train(x, block.downsample[0], block.downsample[1], relu=False)
        else:
            identity = x
        out = conv_bn_act_train(x, block.conv1, block.bn1, relu=True, use_bn=bn_on)
        out = conv_bn_act_train(out, block.conv2, block.bn2, relu=True, use_bn=bn_on)
        return conv_bn_act_train(out, block.conv3, block.bn3, relu=True, residual=identity, use_bn=bn_on)
    weights = [block.conv1.weight, block.conv2.weight, block.conv3.weight]
    if block.downsample is not None:
        weights.append(block.downsample[0].weight)
    return _BottleneckFn.apply(x, block, block.batchnorm_en, *weights)


class _MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) of the stem (lib/nets/resnet.py:156) when the stem trains (FIXED_BLOCKS == -1)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.maxpool3x3s2_nhwc(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool3x3s2_bwd(x, dy.contiguous())


def maxpool_train(x):
    return _MaxPoolFn.apply(x)


class _UpsampleAddFn(torch.autograd.Function):
    """lib/nets/fpn.py:42-45."""

    @staticmethod
    def forward(ctx, x, lateral):
        ctx.in_hw = (x.shape[1], x.shape[2])
        return ops.upsample_bilinear_add(x, lateral)

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        dx = ops.upsample_bilinear_bwd(dout, ctx.in_hw) if ctx.needs_input_grad[0] else None
        return dx, (dout if ctx.needs_input_grad[1] else None)


def upsample_add_train(x, lateral):
    return _UpsampleAddFn.apply(x, lateral)


class _MultiLevelRoIAlignFn(torch.autograd.Function):
    """MultiScaleRoIAlign.forward (lib/utils/torchpoolers.py:137-200): every level writes the RoIs mapped to it."""

    @staticmethod
    def forward(ctx, rois, levels, meta, *feats):
        pooled, scales, sampling = meta
        out = torch.zeros((rois.shape[0], pooled, pooled, feats[0].shape[-1]), dtype=torch.float32, device=rois.device)
        for lvl, (f, sc) in enumerate(zip(feats, scales)):
            ops.roi_align_nhwc(f, rois, pooled, sc, sampling, level_of_roi=levels, level=lvl if levels is not None else -1,
                               out=out)
        ctx.meta = meta
        ctx.shapes = [tuple(f.shape) for f in feats]
        ctx.save_for_backward(rois, levels)
        return out

    @staticmethod
    def backward(ctx, dout):
        rois, levels = ctx.saved_tensors
        pooled, scales, sampling = ctx.meta
        dout = dout.contiguous()
        grads = []
        for lvl, (shape, sc) in enumerate(zip(ctx.shapes, scales)):
            if not ctx.needs_input_grad[3 + lvl]:
                grads.append(None)
                continue
            grads.append(ops.roi_align_bwd(dout, shape, rois, sc, sampling, level_of_roi=levels,
                                           level=lvl if levels is not None else -1))
        return (None, None, None) + tuple(grads)


def roi_align_train(feats, rois, levels, pooled, scales, sampling):
    return _MultiLevelRoIAlignFn.apply(rois, levels, (pooled, tuple(scales), sampling), *feats)


class _SpatialMeanFn(torch.autograd.Function):
    """fc7 = layer4(pool5).mean(3).mean(2) (tail of the non-FPN detector)."""

    @staticmethod
    def forward(ctx, x):
        ctx.pooled = x.shape[1]
        return ops.spatial_mean(x)

    @staticmethod
    def backward(ctx, dout):
        return ops.spatial_mean_bwd(dout.contiguous(), ctx.pooled)


def spatial_mean_train(x):
    return _SpatialMeanFn.apply(x)


class _GatherPatchesFn(torch.autograd.Function):
    """The r x s windows of a (1,H,W,C) map around a device-side list of pixels (ops.gather_patches); backward scatters the
    window gradients back into a zero map (float atomics, like the RoIAlign backward)."""

    @staticmethod
    def forward(ctx, x, idx, count, r, s, pad):
        ctx.meta = (x.shape[1], x.shape[2], pad)
        ctx.save_for_backward(idx, count)
        return ops.gather_patches(x, idx, count, r, s, pad)

    @staticmethod
    def backward(ctx, d):
        idx, count = ctx.saved_tensors
        h, w, pad = ctx.meta
        return ops.scatter_add_patches(d.contiguous(), idx, count, h, w, pad), None, None, None, None, None


def conv_on_patches_train(x, idx, count, conv, holder, relu=True):
    """act(conv(x)) at the listed pixels ONLY, differentiable: a stride-1 padded convolution evaluated at pixel p is the VALID
    convolution of the r x s window around p, so the result (cap, 1, 1, K) equals rows ``idx`` of the dense output and its
    backward costs cap output pixels instead of H x W.  ``holder`` owns the caches of this form (the dense form of the same
    module keeps its own on the module)."""
    r, s = conv.kernel_size
    stride, pad = _stride_pad(conv)
    if stride != 1:
        raise NotImplementedError("conv_on_patches_train: stride-1 convolutions only")
    patches = _GatherPatchesFn.apply(x, idx, count, r, s, pad)
    w_krsc, _, shift = prepared_conv(conv, None, False)
    return _ConvFn.apply(patches, None, conv.weight, conv.bias, (w_krsc, None, shift, 1, 0, relu, holder))


class _RpnLossFn(torch.autograd.Function):
    """cross_entropy over labelled anchors + smooth_l1_loss('RPN', ...) on the fused RPN head output."""

    @staticmethod
    def forward(ctx, rpn2d, labels, targets, inside, outside, num_anchors):
        losses, drpn = ops.rpn_loss(rpn2d, num_anchors, labels, targets, inside, outside, 1.0, 1.0, want_grad=True)
        ctx.a = num_anchors
        ctx.save_for_backward(drpn)
        return losses[:2].clone()

    @staticmethod
    def backward(ctx, g):
        (drpn,) = ctx.saved_tensors
        a = ctx.a
        d = drpn.clone()
        d[:, :2 * a] *= g[0]          # cross-entropy gradient lives in the logit columns,
        d[:, 2 * a:6 * a] *= g[1]     # the box-loss gradient in the delta columns
        return d, None, None, None, None, None


def rpn_loss_train(rpn2d, labels, targets, inside, outside, num_anchors):
    return _RpnLossFn.apply(rpn2d, labels, targets, inside, outside, num_anchors)


class _DetLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cls_score, bbox_pred, labels, targets, inside, outside, lidar):
        losses, dcls, dbox = ops.det_loss(cls_score, labels, bbox_pred, targets, inside, outside, 4, 1.0, 1.0, lidar=lidar)
        ctx.save_for_backward(dcls, dbox)
        return losses

    @staticmethod
    def backward(ctx, g):
        dcls, dbox = ctx.saved_tensors
        return dcls * g[0], dbox * g[1], None, None, None, None, None


def det_loss_train(cls_score, bbox_pred, labels, targets, inside, outside, lidar=None):
    """``lidar=(REG_LOSS_WEIGHT, EN_RY_SIN)`` selects the 7-element form of lib/utils/loss_utils.py:61-77."""
    return _DetLossFn.apply(cls_score.contiguous(), bbox_pred.contiguous(), labels, targets, inside, outside, lidar)
