"""Uncertainty heads of the detectors (cfg.UC.*) — the fork's research feature.

The heads are defined in the reference's missing ``lib/nets/network.py``; what the snapshot pins is
  * the module NAMES and initialisers: ``bbox_fc1/2``, ``cls_fc1/2``, ``bbox_drop1/2``, ``cls_drop1/2``
    (lib/nets/imagenet.py:75-81,167-172), the LiDAR detector's ``bbox_bn1/2``, ``cls_bn1/2`` (lib/nets/lidarnet.py:82-92),
    ``bbox_al_var_net`` / ``cls_al_var_net`` (imagenet.py:88-91),
  * the sizes and rates: with an epistemic flag the heads read ``_det_net_channels = _fc7_channels / 4`` features and
    the dropout rates are 0.1 (box) / 0.3 (class) for the image detector, 0.5 / 0.2 for LiDAR (imagenet.py:52-63,
    lidarnet.py:56-67),
  * the Monte-Carlo protocol: ``net.set_e_num_sample(cfg.UC.E_NUM_SAMPLE)`` around ``test_frame``
    (lib/model/test.py:74-77), dropout modules left in train() mode by ``eval()`` (imagenet.py:165-172),
  * the arithmetic: ``compute_bbox_var``, ``categorical_entropy`` / ``categorical_mutual_information``,
    ``logit_distort``, ``bayesian_cross_entropy``, the aleatoric attenuation of ``smooth_l1_loss``
    (lib/utils/loss_utils.py:82-85,114-169),
  * the consumers: the uncertainty dict keys and their per-detection gather (lib/utils/filter_predictions.py:23-43,
    113-124) and the column stacking of lib/model/test.py:151-159,260-270.
Everything the snapshot leaves open is a NAMED constant below (DESIGN.md section 1 lists them).

Device side: the stochastic passes are ONE batched launch per layer — the first dropout expands the deterministic
activation to T = E_NUM_SAMPLE copies (``frcnn_dropout_fwd`` with ``repeat``), the following Linear layers run on
T x R rows on the matrix cores.  ResNet layer4 is shared by the T passes: the reference builds it without dropout
(``dropout_en`` never reaches ``_make_layer``, lib/nets/resnet.py:157-164).  Random draws are counter-based
(csrc/rng.h) so the CPU oracle replays them.
"""
import torch
import torch.nn as nn

from .. import ops
from ..model.config import cfg
from .autograd_ops import (_BnTrainFn, _Holder, _param_grad_from_krsc, _transposed_filter, _wgrad, linear_train)
from .hip_modules import pad4

# ---- reconstruction constants (the missing network.py) ---------------------------------------------------------------
UC_FC1_DIVISOR = 2                 # *_fc1: fc7 -> fc7 / 2; *_fc2: fc7 / 2 -> _det_net_channels (= fc7 / 4)
CLS_VAR_IS_LOG = True              # cls_al_var_net predicts log-variances like the box head ("x = log(bbox_var)", test.py:82)
E_BBOX_INV_INPUT_IS_STD = True     # EN_BBOX_EPISTEMIC_INV_TRANSFORM: lidar_3d_uncertainty_transform_inv receives sqrt(e_bbox_var)
                                   # (it scales by box sizes, takes exp(u) - 1 and squares: a standard-deviation-like input);
                                   # the call site lived in the missing network.py
BBOX_VAR_ON_DENORMALISED = True    # both box variances describe deltas * STDS + MEANS (what bbox_transform_inv consumes):
                                   # e_bbox_var = variance over the T passes, a_bbox_var = exp(log-variance) * STDS^2
UNCERTAINTY_ORDER = ('a_entropy', 'a_mutual_info', 'a_cls_var', 'e_entropy', 'e_mutual_info', 'e_cls_var',
                     'a_bbox_var', 'e_bbox_var')     # filter_predictions.py:113-124
# counter-based RNG streams (csrc/rng.h): one per stochastic module
STREAM = {'bbox_drop1': 11, 'bbox_drop2': 12, 'cls_drop1': 13, 'cls_drop2': 14, 'logit_distort': 15, 'bayes_ce': 16}


def enabled():
    u = cfg.UC
    return bool(u.EN_BBOX_ALEATORIC or u.EN_CLS_ALEATORIC or u.EN_BBOX_EPISTEMIC or u.EN_CLS_EPISTEMIC)


def check_flags():
    u = cfg.UC
    if u.EN_RPN_BBOX_ALEATORIC or u.EN_RPN_CLS_ALEATORIC or u.EN_RPN_BBOX_EPISTEMIC or u.EN_RPN_CLS_EPISTEMIC:
        raise NotImplementedError("cfg.UC.EN_RPN_*: the reference's detectors define no RPN uncertainty modules "
                                  "(lib/nets/imagenet.py:65-91 initialises none)")
    if bool(u.EN_BBOX_EPISTEMIC) != bool(u.EN_CLS_EPISTEMIC):
        raise NotImplementedError("cfg.UC.EN_BBOX_EPISTEMIC and EN_CLS_EPISTEMIC are switched together by the reference's "
                                  "CLIs (tools/test_net.py:203-204): with one of them both heads read _det_net_channels = "
                                  "fc7/4 features but only one branch has the layers that produce them")
    if u.EN_BBOX_EPISTEMIC_INV_TRANSFORM:
        if not u.EN_BBOX_EPISTEMIC:
            raise NotImplementedError("cfg.UC.EN_BBOX_EPISTEMIC_INV_TRANSFORM transforms the epistemic box variance: it needs "
                                      "cfg.UC.EN_BBOX_EPISTEMIC")
        if cfg.NET_TYPE != 'lidar':
            raise NotImplementedError("cfg.UC.EN_BBOX_EPISTEMIC_INV_TRANSFORM with the image detector: uncertainty_transform_inv "
                                      "(lib/model/bbox_transform.py:107-130) reads 7-element rows [x,y,z,l,w,h,ry] "
                                      "(columns 0::7, 1::7, 3::7, 4::7); the image head predicts 4 elements per class")


def num_uncertainty_pos(num_classes, bbox_elem):
    """Extra columns per detection row (lib/model/test.py:151-159; the class terms are entropy, mutual information and
    one variance per class = 4 for the reference's two classes)."""
    u, n = cfg.UC, 0
    if u.EN_BBOX_ALEATORIC:
        n += bbox_elem
    if u.EN_BBOX_EPISTEMIC:
        n += bbox_elem
    if u.EN_CLS_ALEATORIC:
        n += 2 + num_classes
    if u.EN_CLS_EPISTEMIC:
        n += 2 + num_classes
    return n


def drop_rates(lidar):
    """(cls_drop_rate, bbox_drop_rate, resnet_drop_rate) — imagenet.py:52-63 / lidarnet.py:56-67."""
    if cfg.UC.EN_BBOX_EPISTEMIC or cfg.UC.EN_CLS_EPISTEMIC:
        return (0.2, 0.5, 0.5) if lidar else (0.3, 0.1, 0.5)
    return 0.0, 0.0, 0.0


def build_modules(net, lidar):
    """Creates the modules on ``net`` (state-dict names = the reference's attribute names)."""
    u = cfg.UC
    c, d = net._fc7_channels, net._det_net_channels
    h = c // UC_FC1_DIVISOR
    k, e = net._num_classes, net._bbox_elem()
    for prefix, on, rate in (('bbox', u.EN_BBOX_EPISTEMIC, net._bbox_drop_rate), ('cls', u.EN_CLS_EPISTEMIC, net._cls_drop_rate)):
        if not on:
            continue
        setattr(net, prefix + '_fc1', nn.Linear(c, h))
        setattr(net, prefix + '_fc2', nn.Linear(h, d))
        setattr(net, prefix + '_drop1', nn.Dropout(rate))
        setattr(net, prefix + '_drop2', nn.Dropout(rate))
        if lidar:                                             # lidarnet.py:85-86,91-92
            setattr(net, prefix + '_bn1', nn.BatchNorm1d(h))
            setattr(net, prefix + '_bn2', nn.BatchNorm1d(d))
    if u.EN_BBOX_ALEATORIC:
        net.bbox_al_var_net = nn.Linear(d, k * e)
    if u.EN_CLS_ALEATORIC:
        net.cls_al_var_net = nn.Linear(d, k)


def init_weights(net, normal_init, const_init, truncated, lidar):
    """imagenet.py:74-91 / lidarnet.py:81-102."""
    u = cfg.UC
    for prefix, on in (('bbox', u.EN_BBOX_EPISTEMIC), ('cls', u.EN_CLS_EPISTEMIC)):
        if on:
            normal_init(getattr(net, prefix + '_fc1'), 0, 0.01, truncated)
            normal_init(getattr(net, prefix + '_fc2'), 0, 0.01, truncated)
            if lidar:
                const_init(getattr(net, prefix + '_bn1'), 1.0, 0.0)
                const_init(getattr(net, prefix + '_bn2'), 1.0, 0.0)
    if u.EN_BBOX_ALEATORIC:
        normal_init(net.bbox_al_var_net, 0, 0.001, True)
    if u.EN_CLS_ALEATORIC:
        normal_init(net.cls_al_var_net, 0, 0.01, truncated if lidar else True)


def apply_eval_protocol(net):
    """eval() of the detectors (imagenet.py:165-172): the dropout modules stay stochastic when the full net is on."""
    if cfg.ENABLE_FULL_NET is True and cfg.UC.EN_BBOX_EPISTEMIC:
        net.bbox_drop1.train()
        net.bbox_drop2.train()
    if cfg.ENABLE_FULL_NET is True and cfg.UC.EN_CLS_EPISTEMIC:
        net.cls_drop1.train()
        net.cls_drop2.train()


# ---------------------------------------------------------------------------------------------------------------------
# inference
# ---------------------------------------------------------------------------------------------------------------------
def _linear(x2d, lin, bn=None, relu=False):
    """act(bn(x W^T + b)) on the matrix cores; an eval-mode BatchNorm1d is folded into the epilogue."""
    n, c = x2d.shape
    w = lin.weight.detach().view(lin.out_features, 1, 1, c)
    scale, shift = None, lin.bias.detach()
    if bn is not None:
        if bn.training:
            raise NotImplementedError("BatchNorm1d of the uncertainty heads in train() mode at test time")
        scale = (bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)).contiguous()
        shift = (bn.bias.detach() - bn.running_mean.detach() * scale + lin.bias.detach() * scale).contiguous()
    return ops.conv2d_nhwc(x2d.contiguous().view(n, 1, 1, c), w.contiguous(), scale, shift, None, relu=relu).view(n, -1)


def _branch(net, prefix, fc7, epistemic, samples, seed, seed_dev=None):
    """Features feeding ``<prefix>_pred`` heads: (T * R, D) and T.  T > 1 only when the dropout modules are stochastic."""
    if not epistemic:
        return fc7, 1
    fc1, fc2 = getattr(net, prefix + '_fc1'), getattr(net, prefix + '_fc2')
    d1, d2 = getattr(net, prefix + '_drop1'), getattr(net, prefix + '_drop2')
    bn1, bn2 = getattr(net, prefix + '_bn1', None), getattr(net, prefix + '_bn2', None)
    h = _linear(fc7, fc1, bn1, relu=True)                                   # deterministic: shared by the T passes
    t = samples if (d1.training or d2.training) else 1
    if d1.training:
        h = ops.dropout(h, d1.p, seed, STREAM[prefix + '_drop1'], repeat=t, seed_dev=seed_dev).view(t * fc7.shape[0], -1)
    elif t > 1:
        h = h.repeat(t, 1)
    h = _linear(h, fc2, bn2, relu=True)
    if d2.training:
        h = ops.dropout(h, d2.p, seed, STREAM[prefix + '_drop2'], seed_dev=seed_dev)
    return h, t


def _norm_consts(net, key, k, device):
    """(stds, means) of the box targets tiled per class, as device tensors cached on the net: a host->device copy is not
    allowed inside a stream capture (model/frame_graph.py), and the values only change with cfg."""
    vals = (tuple(float(v) for v in cfg.TRAIN[key].BBOX_NORMALIZE_STDS), tuple(float(v) for v in cfg.TRAIN[key].BBOX_NORMALIZE_MEANS))
    tag = (key, k, str(device), vals)
    cache = net.__dict__.get('_uc_norm_consts')
    if cache is None or cache[0] != tag:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("uncertainty heads: normalisation constants are not on the device yet (run an eager frame first)")
        cache = (tag, torch.tensor(vals[0], dtype=torch.float32, device=device).repeat(k),
                 torch.tensor(vals[1], dtype=torch.float32, device=device).repeat(k))
        net.__dict__['_uc_norm_consts'] = cache
    return cache[1], cache[2]


def classify_test(net, fc7, rois):
    """Test-time heads with cfg.UC.*: fills net._predictions (cls_score, cls_prob, bbox_pred, pred_boxes,
    uncertainties) and returns (cls_prob, bbox_pred)."""
    u = cfg.UC
    r = fc7.shape[0]
    k, e = net._num_classes, net._bbox_elem()
    lidar = cfg.NET_TYPE == 'lidar'
    seed, seed_dev = net.uc_seed_args()       # (host counter, None) eagerly; (0, device word) inside a captured frame
    t_req = max(int(net._e_num_sample), 1)
    unc = {}
    # ---- class branch ----
    feat_c, tc = _branch(net, 'cls', fc7, u.EN_CLS_EPISTEMIC, t_req, seed, seed_dev)
    score_s = _linear(feat_c, net.cls_score_net).view(tc, r, k)
    if u.EN_CLS_EPISTEMIC:
        cls_prob, e_ent, e_mi, e_var = ops.mc_cls_stats(score_s, want_var=True)
    else:
        cls_prob, _, _ = ops.mc_cls_stats(score_s)
    cls_score = ops.mc_mean(score_s) if tc > 1 else score_s[0]
    if u.EN_CLS_ALEATORIC:
        logvar = _linear(feat_c, net.cls_al_var_net).view(tc, r, k)
        logvar = ops.mc_mean(logvar) if tc > 1 else logvar[0]
        dist, a_var = ops.logit_distort(cls_score.contiguous(), logvar.contiguous(), int(u.A_NUM_CE_SAMPLE), seed,
                                        STREAM['logit_distort'], var_is_log=CLS_VAR_IS_LOG, seed_dev=seed_dev)
        _, a_ent, a_mi = ops.mc_cls_stats(dist)
        unc['a_entropy'], unc['a_mutual_info'], unc['a_cls_var'] = a_ent, a_mi, a_var
    if u.EN_CLS_EPISTEMIC:
        unc['e_entropy'], unc['e_mutual_info'], unc['e_cls_var'] = e_ent, e_mi, e_var
    # ---- box branch ----
    feat_b, tb = _branch(net, 'bbox', fc7, u.EN_BBOX_EPISTEMIC, t_req, seed, seed_dev)
    box_s = _linear(feat_b, net.bbox_pred_net).view(tb, r, k * e)
    bbox_pred = ops.mc_mean(box_s) if tb > 1 else box_s[0]
    stds, means = _norm_consts(net, 'LIDAR' if lidar else 'IMAGE', k, fc7.device)
    if u.EN_BBOX_ALEATORIC:
        lv = _linear(feat_b, net.bbox_al_var_net).view(tb, r, k * e)
        a_var = ops.exp((ops.mc_mean(lv) if tb > 1 else lv[0]).contiguous())
        unc['a_bbox_var'] = a_var * stds * stds if BBOX_VAR_ON_DENORMALISED else a_var
    if u.EN_BBOX_EPISTEMIC:
        var = ops.mc_bbox_var(box_s.contiguous()) if tb > 1 else torch.zeros((r, k * e), dtype=torch.float32, device=fc7.device)
        unc['e_bbox_var'] = var * stds * stds if BBOX_VAR_ON_DENORMALISED else var
        if u.EN_BBOX_EPISTEMIC_INV_TRANSFORM:
            # variance of the deltas -> variance-like terms in box space (lib/model/bbox_transform.py:132-169)
            unc['e_bbox_var'] = ops.uncertainty_transform_inv(rois[:, 1:5].contiguous(), unc['e_bbox_var'].contiguous(),
                                                              net._predictions['roi_anchors_3d'], net._frame_scale, lidar=True,
                                                              input_is_variance=E_BBOX_INV_INPUT_IS_STD)
    deltas = (bbox_pred * stds + means).contiguous()
    if lidar:
        pred_boxes = ops.lidar_bbox_transform_inv(rois[:, 1:5].contiguous(), net._predictions['roi_anchors_3d'], deltas,
                                                  net._frame_scale)
    else:
        pred_boxes = ops.bbox_transform_inv(rois[:, 1:5].contiguous(), deltas, net._frame_scale)
    p = net._predictions
    p['cls_score'], p['cls_prob'], p['bbox_pred'], p['pred_boxes'] = cls_score, cls_prob, bbox_pred.contiguous(), pred_boxes
    p['uncertainties'] = {key_: unc[key_] for key_ in UNCERTAINTY_ORDER if key_ in unc}
    p['fc7_uc'] = fc7                    # input of the heads (parity tests run the oracle's heads on it)
    return cls_prob, p['bbox_pred']


def stack_uncertainty_columns(uncertainties, det_roi, num_classes):
    """Per-detection uncertainty columns in UNCERTAINTY_ORDER (nms_hstack_var_torch, filter_predictions.py:23-43):
    det_roi (K, max_out) int32 RoI row of each detection (-1 = padding) -> (K, max_out, U) float32, zeros on padding.
    Box variances take the detection's class columns, class terms are shared by all classes."""
    k, m = det_roi.shape
    idx = det_roi.clamp(min=0).long()
    valid = (det_roi >= 0).unsqueeze(-1).float()
    cols = []
    for name in UNCERTAINTY_ORDER:
        if name not in uncertainties:
            continue
        v = uncertainties[name]
        if name.endswith('bbox_var'):
            e = v.shape[1] // num_classes
            per_cls = torch.stack([v[idx[j], j * e:(j + 1) * e] for j in range(k)], 0)         # (K, max_out, E)
            cols.append(per_cls * valid)
        else:
            v2 = v if v.dim() == 2 else v.unsqueeze(1)
            cols.append(v2[idx] * valid)                                                       # (K, max_out, 1 or K)
    return torch.cat(cols, 2) if cols else None


# ---------------------------------------------------------------------------------------------------------------------
# training
# ---------------------------------------------------------------------------------------------------------------------
class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed, stream, seed_dev=None):
        ctx.meta = (p, seed, stream, seed_dev)
        return ops.dropout(x.contiguous(), p, seed, stream, seed_dev=seed_dev)

    @staticmethod
    def backward(ctx, dy):
        p, seed, stream, seed_dev = ctx.meta
        return ops.dropout_bwd(dy.contiguous(), p, seed, stream, seed_dev=seed_dev), None, None, None, None


def _branch_train(net, prefix, fc7, epistemic, seed, seed_dev=None):
    if not epistemic:
        return fc7
    fc1, fc2 = getattr(net, prefix + '_fc1'), getattr(net, prefix + '_fc2')
    d1, d2 = getattr(net, prefix + '_drop1'), getattr(net, prefix + '_drop2')
    bn1, bn2 = getattr(net, prefix + '_bn1', None), getattr(net, prefix + '_bn2', None)

    def layer(x, lin, bn, drop, stream):
        r = x.shape[0]
        if bn is not None and bn.training:
            y = linear_train(x, lin, relu=False)
            y = _BnTrainFn.apply(y.view(r, 1, 1, -1), None, bn.weight, bn.bias, bn, True).view(r, -1)
        elif bn is not None:
            raise NotImplementedError("BatchNorm1d of the uncertainty heads in eval() mode inside a training step")
        else:
            y = linear_train(x, lin, relu=True)
        return _DropoutFn.apply(y, drop.p, seed, stream, seed_dev) if drop.training and drop.p > 0 else y

    h = layer(fc7, fc1, bn1, d1, STREAM[prefix + '_drop1'])
    return layer(h, fc2, bn2, d2, STREAM[prefix + '_drop2'])


class _MultiHeadFn(torch.autograd.Function):
    """Sibling Linear layers that share their input as ONE convolution: filters concatenated along K and zero-padded to
    a multiple of 4 (the matrix-core kernels' output granularity), like autograd_ops._FusedHeadFn for N >= 1 layers."""

    @staticmethod
    def forward(ctx, x, pack, *params):
        w_krsc, bias_cat, owner, sizes = pack
        ctx.pack = pack
        ctx.save_for_backward(x, *params)
        return ops.conv2d_nhwc(x, w_krsc, None, bias_cat, None)

    @staticmethod
    def backward(ctx, dy):
        x, *params = ctx.saved_tensors
        w_krsc, _, owner, sizes = ctx.pack
        dy = dy.contiguous()
        targets, lo = [], 0
        for i, k in enumerate(sizes):
            w, b = params[2 * i], params[2 * i + 1]
            targets.append((w, lambda dwk, dbk, lo=lo, k=k, w=w: _param_grad_from_krsc(dwk[lo:lo + k], w), 'w'))
            targets.append((b, lambda dwk, dbk, lo=lo, k=k: dbk[lo:lo + k].contiguous(), 'b'))
            lo += k
        grads = _wgrad(x, dy, 1, 1, 1, 0, targets)
        dx = ops.conv2d_bwd_data(dy, _transposed_filter(owner, w_krsc), tuple(x.shape)) if ctx.needs_input_grad[0] else None
        return (dx, None) + tuple(grads)


def multi_head_train(x2d, net, modules, cache_name):
    """[lin(x) for lin in modules] through one fused, differentiable launch."""
    tensors = [t for m in modules for t in (m.weight, m.bias)]
    key = tuple((t._version, t.data_ptr()) for t in tensors)
    cache = net.__dict__.get(cache_name)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            sizes = [m.out_features for m in modules]
            kp = pad4(sum(sizes))
            w = torch.zeros((kp, x2d.shape[1]), dtype=torch.float32, device=x2d.device)
            b = torch.zeros((kp,), dtype=torch.float32, device=x2d.device)
            lo = 0
            for m in modules:
                w[lo:lo + m.out_features] = m.weight.detach()
                b[lo:lo + m.out_features] = m.bias.detach()
                lo += m.out_features
            cache = (key, w.view(kp, 1, 1, -1).contiguous(), b, tuple(sizes))
        net.__dict__[cache_name] = cache
    _, w, b, sizes = cache
    r = x2d.shape[0]
    y = _MultiHeadFn.apply(x2d.contiguous().view(r, 1, 1, -1), (w, b, _Holder.of(net, cache_name), sizes), *tensors).view(r, -1)
    outs, lo = [], 0
    for k in sizes:
        outs.append(y[:, lo:lo + k])
        lo += k
    return outs


def classify_train(net, fc7):
    """Training-time heads with cfg.UC.*: one stochastic pass (dropout masks of this step), raw log-variance outputs."""
    u = cfg.UC
    seed, seed_dev = net.uc_seed_args()
    p = net._predictions
    feat_c = _branch_train(net, 'cls', fc7, u.EN_CLS_EPISTEMIC, seed, seed_dev)
    feat_b = _branch_train(net, 'bbox', fc7, u.EN_BBOX_EPISTEMIC, seed, seed_dev)
    cls_heads = [net.cls_score_net] + ([net.cls_al_var_net] if u.EN_CLS_ALEATORIC else [])
    box_heads = [net.bbox_pred_net] + ([net.bbox_al_var_net] if u.EN_BBOX_ALEATORIC else [])
    if feat_c is feat_b:                                   # no epistemic stacks: every head reads fc7
        outs = multi_head_train(fc7, net, cls_heads + box_heads, '_uc_heads_cache')
        cls_out, box_out = outs[:len(cls_heads)], outs[len(cls_heads):]
    else:
        cls_out = multi_head_train(feat_c, net, cls_heads, '_uc_cls_heads_cache')
        box_out = multi_head_train(feat_b, net, box_heads, '_uc_box_heads_cache')
    p['cls_score'], p['bbox_pred'] = cls_out[0], box_out[0]
    p['cls_var'] = cls_out[1] if u.EN_CLS_ALEATORIC else None
    p['bbox_var'] = box_out[1] if u.EN_BBOX_ALEATORIC else None
    p['uc_seed'], p['uc_seed_dev'] = seed, seed_dev
    return None, p['bbox_pred']


class _DetLossUcFn(torch.autograd.Function):
    """Second-stage losses with the aleatoric terms: box loss attenuated by the predicted log-variance
    (loss_utils.py:82-85) and / or bayesian_cross_entropy over A_NUM_CE_SAMPLE distorted logits (:149-169)."""

    @staticmethod
    def forward(ctx, cls_score, bbox_pred, bbox_var, cls_var, labels, targets, inside, outside, meta):
        bbox_elem, lidar, (seed, seed_dev), num_ce = meta
        weights, ry_sin = (lidar if lidar is not None else (None, False))
        dvar = dcvar = None
        if bbox_var is not None:
            losses, dcls, dbox, dvar = ops.det_loss_aleatoric(cls_score, labels, bbox_pred, bbox_var, targets, inside, outside,
                                                              bbox_elem, weights, ry_sin)
        else:
            losses, dcls, dbox = ops.det_loss(cls_score, labels, bbox_pred, targets, inside, outside, bbox_elem, 1.0, 1.0,
                                              lidar=lidar)
        losses = losses.clone()
        if cls_var is not None:
            ce, dcls, dcvar = ops.bayesian_cross_entropy(cls_score, cls_var, labels, num_ce, seed, STREAM['bayes_ce'],
                                                         var_is_log=CLS_VAR_IS_LOG, seed_dev=seed_dev)
            losses[0] = ce[0]
        ctx.save_for_backward(dcls, dbox, dvar, dcvar)
        return losses

    @staticmethod
    def backward(ctx, g):
        dcls, dbox, dvar, dcvar = ctx.saved_tensors
        return (dcls * g[0], dbox * g[1], dvar * g[1] if dvar is not None else None,
                dcvar * g[0] if dcvar is not None else None, None, None, None, None, None)


def det_loss_uc(net, labels, targets, inside, outside, lidar):
    p = net._predictions
    meta = (net._bbox_elem(), lidar, (p['uc_seed'], p.get('uc_seed_dev')), int(cfg.UC.A_NUM_CE_SAMPLE))
    c = lambda t: t.contiguous() if t is not None else None
    return _DetLossUcFn.apply(c(p['cls_score']), c(p['bbox_pred']), c(p['bbox_var']), c(p['cls_var']), labels, targets,
                              inside, outside, meta)
