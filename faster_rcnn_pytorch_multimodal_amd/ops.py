"""Tensor-level wrappers over the C ABI (include/frcnn_hip.h).

torch is used here only as the owner of device memory and of the HIP stream; every computation is a
libfrcnn_hip.so kernel launched on ``torch.cuda.current_stream()``.  Nothing synchronises with the host,
so a sequence of these calls can be captured into one hipGraph.  CPU tensors are rejected: there is no
non-HIP execution path in this package.

Layouts: activations NHWC ``(N, H, W, C)`` contiguous fp32, filters KRSC ``(K, R, S, C)`` contiguous fp32.
"""
import ctypes

import os

import torch

from . import _hip

# Shape log for bench.py / tuning: when PROFILE is a list, every conv2d_nhwc call appends its shape dict, in call order -
# the order of the call numbers conv_profile_end() reports with the per-dispatch durations.
PROFILE = None
# FLOP accounting for bench.py's training / LiDAR rooflines: when FLOPS is a dict, every forward / data-gradient /
# filter-gradient convolution call adds its direct-form FLOPs under 'fwd' / 'dgrad' / 'wgrad' and what the matrix pipe
# executes under '<kind>_executed' (a Winograd F(2x2,3x3) plan multiplies 16 values per 2x2 output tile instead of 36).
FLOPS = None


def flops_begin():
    global FLOPS
    FLOPS = {k: 0.0 for k in ('fwd', 'dgrad', 'wgrad', 'fwd_executed', 'dgrad_executed', 'wgrad_executed')}


def flops_end():
    global FLOPS
    out, FLOPS = FLOPS, None
    return out


def _log_flops(kind, direct, winograd=False):
    if FLOPS is not None:
        FLOPS[kind] += direct
        FLOPS[kind + '_executed'] += direct * (16.0 / 36.0 if winograd else 1.0)


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev_f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _hip.HipError("%s must be a tensor on the MI355X (got %s); this package has no CPU path"
                            % (name, getattr(t, "device", type(t))))
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise _hip.HipError("%s must be contiguous float32 (got %s, contiguous=%s)" % (name, t.dtype, t.is_contiguous()))
    return t


def _workspace(nbytes, device):
    """Scratch buffer; a fresh allocation keeps graph capture simple (the caching allocator pools it)."""
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# what set_conv_autotune(True) means: 1 = every candidate plan is timed alone on an idle chip, 2 = under load (four launches
# of the candidate in flight on four streams: ranked by throughput, the objective of the four-frames-in-flight schedule)
AUTOTUNE_LEVEL = int(os.environ.get('FRCNN_AUTOTUNE_LEVEL', '2'))


def set_conv_autotune(enable):
    """Turn the convolution plan autotuner on/off (frcnn_conv2d_set_autotune): tune during eager warm-up frames,
    the cached plans are then used inside captured graphs.  ``True`` = AUTOTUNE_LEVEL; 1 / 2 select the level."""
    global _CONV_AUTOTUNE
    level = (AUTOTUNE_LEVEL if enable is True else int(enable)) if enable else 0
    _hip.check(_hip.load().frcnn_conv2d_set_autotune(level), "frcnn_conv2d_set_autotune")
    _CONV_AUTOTUNE = bool(enable)


def set_conv_algo(mode):
    """0 = the autotuner may choose Winograd F(2x2, 3x3) for the eligible 3x3 layers, 1 = implicit GEMM only,
    2 = Winograd wherever it applies (frcnn_conv2d_set_algo).  Flags: +16 never fuse the Winograd input transform into the
    64x64 GEMM's tile load, +32 forced Winograd (2) uses that fused form wherever C % 32 == 0 (tests)."""
    global _CONV_ALGO_MODE, _CONV_ALGO_FLAGS
    _hip.check(_hip.load().frcnn_conv2d_set_algo(int(mode)), "frcnn_conv2d_set_algo")
    _CONV_ALGO_MODE = int(mode) & 3
    _CONV_ALGO_FLAGS = int(mode) & ~3


_CONV_ALGO_MODE = 0
_CONV_ALGO_FLAGS = 0
_CONV_AUTOTUNE = False


def conv_plan_algo(n, h, w, c, k, r, s, stride, pad, has_residual=False):
    """Form of the cached plan of this forward shape: -1 none, 0 implicit GEMM, 1 Winograd (frcnn_conv2d_plan_algo)."""
    return int(_hip.load().frcnn_conv2d_plan_algo(n, h, w, c, k, r, s, stride, pad, int(bool(has_residual))))


def winograd_filter_wanted(n, h, w, c, k, r, s, stride, pad):
    """Whether a residual-free call of this shape can read a pre-transformed Winograd filter NOW: its cached plan is a
    Winograd plan, or Winograd is forced (algo mode 2), or the shape is about to be tuned (autotune on, no plan yet: the tuner
    times the Winograd form too)."""
    if _CONV_ALGO_MODE == 1 or not winograd_eligible(k, r, s, c, stride, pad):
        return False
    if _CONV_ALGO_MODE == 2:
        return True
    algo = conv_plan_algo(n, h, w, c, k, r, s, stride, pad, False)
    return algo == 1 or (algo < 0 and _CONV_AUTOTUNE)


def conv_profile_begin():
    """Start per-dispatch timing of the convolution kernels (frcnn_conv2d_profile_begin)."""
    _hip.check(_hip.load().frcnn_conv2d_profile_begin(), "frcnn_conv2d_profile_begin")


def conv_profile_end(capacity=1 << 16):
    """Stop it; returns a list of (microseconds, conv2d_fwd call number, kind) per dispatch, kind 0 = main kernel,
    1 = split-K second pass."""
    us = (ctypes.c_float * capacity)()
    call = (ctypes.c_int * capacity)()
    kind = (ctypes.c_int * capacity)()
    n = _hip.load().frcnn_conv2d_profile_end(us, call, kind, capacity)
    if n > capacity:
        raise _hip.HipError("conv_profile_end: %d dispatches > capacity %d" % (n, capacity))
    return [(float(us[i]), int(call[i]), int(kind[i])) for i in range(n)]


def export_conv_plans():
    """The tuned convolution plans as a list of 13-int rows (frcnn_conv2d_export_plans)."""
    lib = _hip.load()
    n = lib.frcnn_conv2d_export_plans(None, 0)
    buf = (ctypes.c_int * (13 * max(n, 1)))()
    n = min(n, lib.frcnn_conv2d_export_plans(buf, n))
    return [[int(buf[13 * e + i]) for i in range(13)] for e in range(n)]


def import_conv_plans(rows):
    """Install plans saved by export_conv_plans (e.g. to profile exactly the kernels a timed run used)."""
    flat = [int(v) for row in rows for v in row]
    if len(flat) % 13:
        raise _hip.HipError("import_conv_plans: rows must have 13 ints")
    buf = (ctypes.c_int * max(len(flat), 1))(*flat)
    _hip.check(_hip.load().frcnn_conv2d_import_plans(buf, len(flat) // 13), "frcnn_conv2d_import_plans")


def conv_out_hw(h, w, r, s, stride, pad):
    return (h + 2 * pad - r) // stride + 1, (w + 2 * pad - s) // stride + 1


def conv2d_nhwc(x, w_krsc, scale=None, shift=None, residual=None, stride=1, pad=0, relu=False, split_k=0, out=None,
                w_winograd=None):
    """y = act(conv(x, w) * scale + shift + residual)  —  frcnn_conv2d_fwd (frcnn_conv2d_fwd_pre when the caller supplies
    the Winograd-transformed filter of a 3x3 / stride 1 / pad 1 layer, see ``winograd_filter``)."""
    lib = _hip.load()
    _dev_f32(x, "x"); _dev_f32(w_krsc, "w")
    n, h, w, c = x.shape
    k, r, s, c2 = w_krsc.shape
    if c2 != c:
        raise _hip.HipError("conv2d_nhwc: input has %d channels, filter expects %d" % (c, c2))
    for nm, t in (("scale", scale), ("shift", shift)):
        if t is not None:
            _dev_f32(t, nm)
            if t.numel() != k:
                raise _hip.HipError("conv2d_nhwc: %s has %d elements, expected %d" % (nm, t.numel(), k))
    ho, wo = conv_out_hw(h, w, r, s, stride, pad)
    if out is None:
        out = torch.empty((n, ho, wo, k), dtype=torch.float32, device=x.device)
    else:
        _dev_f32(out, "out")
        if tuple(out.shape) != (n, ho, wo, k):
            raise _hip.HipError("conv2d_nhwc: out has shape %s, expected %s" % (tuple(out.shape), (n, ho, wo, k)))
    if residual is not None:
        _dev_f32(residual, "residual")
        if tuple(residual.shape) != (n, ho, wo, k):
            raise _hip.HipError("conv2d_nhwc: residual shape %s != output shape %s" % (tuple(residual.shape), (n, ho, wo, k)))
    ws_bytes = lib.frcnn_conv2d_fwd_ws_bytes(n, h, w, c, k, r, s, stride, pad, split_k)
    ws = _workspace(ws_bytes, x.device) if ws_bytes else None
    if w_winograd is not None:
        _dev_f32(w_winograd, "w_winograd")
        if tuple(w_winograd.shape) != (16, k, c):
            raise _hip.HipError("conv2d_nhwc: w_winograd has shape %s, expected %s" % (tuple(w_winograd.shape), (16, k, c)))
        _hip.check(lib.frcnn_conv2d_fwd_pre(_ptr(x), _ptr(w_krsc), _ptr(w_winograd), _ptr(scale), _ptr(shift), _ptr(residual),
                                            _ptr(out), n, h, w, c, k, r, s, stride, pad, int(bool(relu)), split_k, _ptr(ws),
                                            ws_bytes, _stream()), "frcnn_conv2d_fwd_pre")
    else:
        _hip.check(lib.frcnn_conv2d_fwd(_ptr(x), _ptr(w_krsc), _ptr(scale), _ptr(shift), _ptr(residual), _ptr(out), n, h, w,
                                        c, k, r, s, stride, pad, int(bool(relu)), split_k, _ptr(ws), ws_bytes, _stream()),
                   "frcnn_conv2d_fwd")
    if PROFILE is not None:
        PROFILE.append({"n": n, "h": h, "w": w, "c": c, "k": k, "r": r, "s": s, "stride": stride, "pad": pad,
                        "residual": residual is not None, "relu": bool(relu), "flops": 2.0 * n * ho * wo * k * r * s * c})
    if FLOPS is not None:
        _log_flops('fwd', 2.0 * n * ho * wo * k * r * s * c,
                   residual is None and winograd_eligible(k, r, s, c, stride, pad) and _CONV_ALGO_MODE != 1 and
                   (_CONV_ALGO_MODE == 2 or conv_plan_algo(n, h, w, c, k, r, s, stride, pad, False) == 1))
    return out


def winograd_eligible(k, r, s, c, stride, pad):
    """The layers frcnn_conv2d_set_algo's Winograd F(2x2, 3x3) form applies to (no residual operand)."""
    return r == 3 and s == 3 and stride == 1 and pad == 1 and c % 4 == 0 and k % 4 == 0


def winograd_filter(w_krsc):
    """(K,3,3,C) filter -> its Winograd transform U (16,K,C) (frcnn_conv2d_winograd_filter); constant while the weights are."""
    lib = _hip.load()
    _dev_f32(w_krsc, "w")
    k, r, s, c = w_krsc.shape
    if not winograd_eligible(k, r, s, c, 1, 1):
        raise _hip.HipError("winograd_filter: need a (k,3,3,c) filter with k%4 == 0 and c%4 == 0")
    u = torch.empty((16, k, c), dtype=torch.float32, device=w_krsc.device)
    _hip.check(lib.frcnn_conv2d_winograd_filter(_ptr(w_krsc), _ptr(u), k, c, _stream()), "frcnn_conv2d_winograd_filter")
    return u


def conv2d_transpose_filter(w_krsc):
    """(K,R,S,C) -> (C,R,S,K) with flipped taps: the filter of the data-gradient convolution."""
    lib = _hip.load()
    _dev_f32(w_krsc, "w")
    k, r, s, c = w_krsc.shape
    out = torch.empty((c, r, s, k), dtype=torch.float32, device=w_krsc.device)
    _hip.check(lib.frcnn_conv2d_transpose_filter(_ptr(w_krsc), _ptr(out), k, r, s, c, _stream()),
               "frcnn_conv2d_transpose_filter")
    return out


def dgrad_winograd_wanted(x_shape, k, r, s, stride, pad):
    """Whether the data-gradient convolution of this layer (a stride-1 3x3 convolution of dy with the transposed filter) can
    read a pre-transformed Winograd filter now (see ``winograd_filter_wanted``)."""
    n, h, w, c = x_shape
    if stride != 1 or r != 3 or s != 3 or r - 1 - pad != 1:
        return False
    ho, wo = conv_out_hw(h, w, r, s, stride, pad)
    return winograd_filter_wanted(n, ho, wo, k, c, r, s, 1, 1)


def conv2d_bwd_data(dy, w_t, x_shape, stride=1, pad=0, add=None, w_winograd=None, act_y=None, act_scale=None):
    """dx (n,h,w,c) = conv_transpose(dy, w) [+ add].  w_t = conv2d_transpose_filter(w); x_shape = forward input shape.
    ``w_winograd`` = winograd_filter(w_t) (16, c, k): frcnn_conv2d_bwd_data_pre.  ``act_y`` (shape of dx) [, ``act_scale``
    (c,)]: the result is masked / scaled like ``act_bwd(dx, act_y, act_scale, relu=True)`` would in a second pass
    (frcnn_conv2d_bwd_data_act; not for the strided 1x1 form)."""
    lib = _hip.load()
    _dev_f32(dy, "dy"); _dev_f32(w_t, "w_t")
    n, h, w, c = x_shape
    c2, r, s, k = w_t.shape
    if c2 != c or dy.shape[-1] != k:
        raise _hip.HipError("conv2d_bwd_data: filter (C=%d,K=%d) does not match x C=%d / dy K=%d" % (c2, k, c, dy.shape[-1]))
    if tuple(dy.shape[:3]) != (n,) + conv_out_hw(h, w, r, s, stride, pad):
        raise _hip.HipError("conv2d_bwd_data: dy shape %s does not match the forward output" % (tuple(dy.shape),))
    if add is not None:
        _dev_f32(add, "add")
        if tuple(add.shape) != tuple(x_shape):
            raise _hip.HipError("conv2d_bwd_data: add shape %s != x shape %s" % (tuple(add.shape), tuple(x_shape)))
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    if FLOPS is not None:
        _log_flops('dgrad', 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * k * r * s * c,
                   add is None and dgrad_winograd_wanted(x_shape, k, r, s, stride, pad))
    ws_bytes = lib.frcnn_conv2d_bwd_data_ws_bytes(n, h, w, c, k, r, s, stride, pad)
    ws = _workspace(ws_bytes, dy.device) if ws_bytes else None
    if act_y is not None:
        _dev_f32(act_y, "act_y")
        if tuple(act_y.shape) != tuple(x_shape):
            raise _hip.HipError("conv2d_bwd_data: act_y shape %s != x shape %s" % (tuple(act_y.shape), tuple(x_shape)))
        if act_scale is not None:
            _dev_f32(act_scale, "act_scale")
            if act_scale.numel() != c:
                raise _hip.HipError("conv2d_bwd_data: act_scale has %d elements, expected %d" % (act_scale.numel(), c))
        if w_winograd is not None:
            _dev_f32(w_winograd, "w_winograd")
            if tuple(w_winograd.shape) != (16, c, k) or add is not None:
                w_winograd = None
        _hip.check(lib.frcnn_conv2d_bwd_data_act(_ptr(dy), _ptr(w_t), _ptr(w_winograd), _ptr(add), _ptr(act_y), _ptr(act_scale),
                                                 _ptr(dx), n, h, w, c, k, r, s, stride, pad, _ptr(ws), ws_bytes, _stream()),
                   "frcnn_conv2d_bwd_data_act")
        return dx
    if w_winograd is not None and add is None:
        _dev_f32(w_winograd, "w_winograd")
        if tuple(w_winograd.shape) != (16, c, k):
            raise _hip.HipError("conv2d_bwd_data: w_winograd has shape %s, expected %s" % (tuple(w_winograd.shape), (16, c, k)))
        _hip.check(lib.frcnn_conv2d_bwd_data_pre(_ptr(dy), _ptr(w_t), _ptr(w_winograd), None, _ptr(dx), n, h, w, c, k, r, s,
                                                 stride, pad, _ptr(ws), ws_bytes, _stream()), "frcnn_conv2d_bwd_data_pre")
        return dx
    _hip.check(lib.frcnn_conv2d_bwd_data(_ptr(dy), _ptr(w_t), _ptr(add), _ptr(dx), n, h, w, c, k, r, s, stride, pad,
                                         _ptr(ws), ws_bytes, _stream()), "frcnn_conv2d_bwd_data")
    return dx


# Tile counters of the filter-gradient and BatchNorm kernels (`counters` of frcnn_conv2d_bwd_weight[_acc], frcnn_bn_train_fwd /
# _bwd: the last workgroup of a tile / column block finishes the reduction): device ints that are zero when a
# launch starts and zero again when it ends; launches that may be in flight together must not share them.
#   * eager launches take consecutive ranges of a per-device ring: a range comes round again only after WGRAD_COUNTER_RING
#     ints of later launches - far more launches than can be in flight;
#   * a holder of a captured graph installs its OWN arena for its warm-up and capture (``wgrad_counter_arena``; the ranges
#     are baked into the graph, and no other graph or eager launch ever gets them) and rewinds it at the start of every
#     pass over its launch sequence;
#   * a capture WITHOUT an arena zero-fills fresh counters inside the capture (private to the graph, one fill kernel per call).
WGRAD_COUNTER_RING = 1 << 20
_COUNTER_RINGS = {}
WGRAD_ARENA = None


class CounterArena:
    def __init__(self, device, count=1 << 19):
        self.ints = torch.zeros(int(count), dtype=torch.int32, device=device)
        self.cursor = 0

    def rewind(self):
        self.cursor = 0

    def take(self, count):
        if self.cursor + count > self.ints.numel():
            raise _hip.HipError("filter-gradient counter arena exhausted (%d + %d > %d ints)" % (self.cursor, count, self.ints.numel()))
        start = self.cursor
        self.cursor += count
        return self.ints[start:start + count]


class wgrad_counter_arena:
    """``with ops.wgrad_counter_arena(arena):`` - filter-gradient launches inside take their tile counters from ``arena``."""

    def __init__(self, arena):
        self.arena = arena

    def __enter__(self):
        global WGRAD_ARENA
        self.prev, WGRAD_ARENA = WGRAD_ARENA, self.arena
        return self.arena

    def __exit__(self, *exc):
        global WGRAD_ARENA
        WGRAD_ARENA = self.prev
        return False


def _tile_counters(device, count):
    if WGRAD_ARENA is not None:
        if WGRAD_ARENA.ints.device != device:
            raise _hip.HipError("filter-gradient counter arena is on %s, the launch on %s" % (WGRAD_ARENA.ints.device, device))
        return WGRAD_ARENA.take(count)
    if torch.device(device).type == 'cuda' and torch.cuda.is_current_stream_capturing():
        return torch.zeros(count, dtype=torch.int32, device=device)
    ring = _COUNTER_RINGS.get(str(device))
    if ring is None:
        ring = _COUNTER_RINGS[str(device)] = CounterArena(device, WGRAD_COUNTER_RING)
    if ring.cursor + count > ring.ints.numel():
        ring.rewind()
    return ring.take(count)


def set_wgrad_variant(variant):
    """frcnn_conv2d_wgrad_set_variant: 0 every filter-gradient kernel, 1 register-staged only, 2 LDS-DMA wherever it applies."""
    _hip.check(_hip.load().frcnn_conv2d_wgrad_set_variant(int(variant)), "frcnn_conv2d_wgrad_set_variant")


def set_wgrad_plan(kernel, splits=1):
    """frcnn_conv2d_wgrad_set_plan: force (kernel 1..4, pixel splits) for every following filter gradient; kernel 0 = off."""
    _hip.check(_hip.load().frcnn_conv2d_wgrad_set_plan(int(kernel), int(splits)), "frcnn_conv2d_wgrad_set_plan")


def conv2d_bwd_weight(x, dy, r, s, stride=1, pad=0, want_bias=False):
    """Returns (dw (K,R,S,C), db (K,) or None)."""
    lib = _hip.load()
    _dev_f32(x, "x"); _dev_f32(dy, "dy")
    n, h, w, c = x.shape
    k = dy.shape[-1]
    if tuple(dy.shape[:3]) != (n,) + conv_out_hw(h, w, r, s, stride, pad):
        raise _hip.HipError("conv2d_bwd_weight: dy shape %s does not match the forward output" % (tuple(dy.shape),))
    dw = torch.empty((k, r, s, c), dtype=torch.float32, device=x.device)
    db = torch.empty((k,), dtype=torch.float32, device=x.device) if want_bias else None
    _log_flops('wgrad', 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * k * r * s * c)
    ws_bytes = lib.frcnn_conv2d_bwd_weight_ws_bytes(n, h, w, c, k, r, s, stride, pad)
    ws = _workspace(ws_bytes, x.device) if ws_bytes else None
    counters = _tile_counters(x.device, lib.frcnn_conv2d_bwd_weight_counters(c, k, r, s))
    _hip.check(lib.frcnn_conv2d_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(db), n, h, w, c, k, r, s, stride, pad,
                                           _ptr(ws), ws_bytes, counters.data_ptr(), _stream()), "frcnn_conv2d_bwd_weight")
    return dw, db


def conv2d_bwd_weight_acc(x, dy, r, s, grad_w, grad_b=None, stride=1, pad=0):
    """grad_w (K, C_real, R, S) [or (K, C_real) with r = s = 1] += filter gradient, grad_b (K,) += bias gradient: the
    parameter's own gradient buffers, accumulated in place (frcnn_conv2d_bwd_weight_acc)."""
    lib = _hip.load()
    _dev_f32(x, "x"); _dev_f32(dy, "dy"); _dev_f32(grad_w, "grad_w")
    n, h, w, c = x.shape
    k = dy.shape[-1]
    if tuple(dy.shape[:3]) != (n,) + conv_out_hw(h, w, r, s, stride, pad):
        raise _hip.HipError("conv2d_bwd_weight_acc: dy shape %s does not match the forward output" % (tuple(dy.shape),))
    c_real = grad_w.shape[1]
    if grad_w.shape[0] != k or c_real > c or grad_w.numel() != k * c_real * r * s:
        raise _hip.HipError("conv2d_bwd_weight_acc: grad_w %s does not fit k=%d c<=%d r=%d s=%d" % (tuple(grad_w.shape), k, c, r, s))
    if grad_b is not None:
        _dev_f32(grad_b, "grad_b")
    _log_flops('wgrad', 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * k * r * s * c)
    ws_bytes = lib.frcnn_conv2d_bwd_weight_ws_bytes(n, h, w, c, k, r, s, stride, pad)
    ws = _workspace(ws_bytes, x.device)
    counters = _tile_counters(x.device, lib.frcnn_conv2d_bwd_weight_counters(c, k, r, s))
    _hip.check(lib.frcnn_conv2d_bwd_weight_acc(_ptr(x), _ptr(dy), _ptr(grad_w), c_real, _ptr(grad_b), n, h, w, c, k, r, s,
                                               stride, pad, _ptr(ws), ws_bytes, counters.data_ptr(), _stream()),
               "frcnn_conv2d_bwd_weight_acc")


WGRAD_MAX_GROUPS = 24


def conv2d_bwd_weight_acc_grouped(xs, dys, r, s, grads, stride=1, pad=0):
    """``grads[g]`` += filter gradient of convolution g for up to WGRAD_MAX_GROUPS convolutions of identical shape
    (frcnn_conv2d_bwd_weight_acc_grouped): one launch pair, no pixel split, no slab reduction."""
    import ctypes
    lib = _hip.load()
    groups = len(xs)
    if not (0 < groups <= WGRAD_MAX_GROUPS) or len(dys) != groups or len(grads) != groups:
        raise _hip.HipError("conv2d_bwd_weight_acc_grouped: 1..%d groups, same number of x / dy / grad tensors" % WGRAD_MAX_GROUPS)
    n, h, w, c = xs[0].shape
    k = dys[0].shape[-1]
    c_real = grads[0].shape[1]
    for x, dy, gw in zip(xs, dys, grads):
        _dev_f32(x, "x"); _dev_f32(dy, "dy"); _dev_f32(gw, "grad_w")
        if tuple(x.shape) != (n, h, w, c) or tuple(dy.shape) != (n,) + conv_out_hw(h, w, r, s, stride, pad) + (k,):
            raise _hip.HipError("conv2d_bwd_weight_acc_grouped: every group must have the shape of the first")
        if gw.shape[0] != k or gw.shape[1] != c_real or c_real > c or gw.numel() != k * c_real * r * s:
            raise _hip.HipError("conv2d_bwd_weight_acc_grouped: grad_w %s does not fit k=%d c<=%d r=%d s=%d"
                                % (tuple(gw.shape), k, c, r, s))
    arr = ctypes.c_void_p * groups
    _log_flops('wgrad', 2.0 * groups * dys[0].shape[0] * dys[0].shape[1] * dys[0].shape[2] * k * r * s * c)
    ws_bytes = lib.frcnn_conv2d_bwd_weight_acc_grouped_ws_bytes(groups, c, k, r, s)
    ws = _workspace(ws_bytes, xs[0].device)
    _hip.check(lib.frcnn_conv2d_bwd_weight_acc_grouped(arr(*[t.data_ptr() for t in xs]), arr(*[t.data_ptr() for t in dys]),
                                                       arr(*[t.data_ptr() for t in grads]), groups, c_real, n, h, w, c, k, r, s,
                                                       stride, pad, _ptr(ws), ws_bytes, _stream()),
               "frcnn_conv2d_bwd_weight_acc_grouped")


def maxpool3x3s2_nhwc(x):
    lib = _hip.load()
    _dev_f32(x, "x")
    n, h, w, c = x.shape
    out = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), dtype=torch.float32, device=x.device)
    _hip.check(lib.frcnn_maxpool3x3s2_fwd(_ptr(x), _ptr(out), n, h, w, c, _stream()), "frcnn_maxpool3x3s2_fwd")
    return out


def maxpool3x3s2_bwd(x, dy):
    """Backward of maxpool3x3s2_nhwc: x (N,H,W,C) the pooled input, dy (N,Ho,Wo,C) -> dx like x."""
    lib = _hip.load()
    _dev_f32(x, "x"); _dev_f32(dy, "dy")
    n, h, w, c = x.shape
    if tuple(dy.shape) != (n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c):
        raise _hip.HipError("maxpool3x3s2_bwd: dy %s does not match x %s" % (tuple(dy.shape), tuple(x.shape)))
    dx = torch.empty_like(x)
    _hip.check(lib.frcnn_maxpool3x3s2_bwd(_ptr(x), _ptr(dy), _ptr(dx), n, h, w, c, _stream()), "frcnn_maxpool3x3s2_bwd")
    return dx


def pad_channels(x, c_pad):
    """(N,H,W,C) -> (N,H,W,c_pad) zero-padded channels."""
    lib = _hip.load()
    _dev_f32(x, "x")
    n, h, w, c = x.shape
    if c == c_pad:
        return x
    out = torch.empty((n, h, w, c_pad), dtype=torch.float32, device=x.device)
    _hip.check(lib.frcnn_pad_channels(_ptr(x), _ptr(out), n * h * w, c, c_pad, _stream()), "frcnn_pad_channels")
    return out


def generate_anchors(base_f64, height, width, feat_stride):
    """base_f64: (A,4) float64 DEVICE tensor -> (H*W*A, 4) float32 anchors."""
    lib = _hip.load()
    if not base_f64.is_cuda or base_f64.dtype != torch.float64 or not base_f64.is_contiguous():
        raise _hip.HipError("generate_anchors: base must be a contiguous float64 device tensor")
    a = base_f64.shape[0]
    out = torch.empty((height * width * a, 4), dtype=torch.float32, device=base_f64.device)
    _hip.check(lib.frcnn_generate_anchors(_ptr(base_f64), a, height, width, feat_stride, _ptr(out), _stream()),
               "frcnn_generate_anchors")
    return out


def rpn_decode_clip(anchors, info, num_anchors, rpn=None, probs=None, deltas=None):
    """Front half of proposal_layer. Either `rpn` (H*W, >=6A) fused head output or (probs, deltas).
    info: host sequence [x_min, x_max, y_min, y_max, ...]. Returns (scores (N,), proposals (N,4))."""
    lib = _hip.load()
    _dev_f32(anchors, "anchors")
    total = anchors.shape[0]
    hw = total // num_anchors
    ld = 0
    if rpn is not None:
        _dev_f32(rpn, "rpn")
        ld = rpn.shape[-1]
        if rpn.numel() != hw * ld:
            raise _hip.HipError("rpn_decode_clip: rpn has %d elements, expected %d x %d" % (rpn.numel(), hw, ld))
    if probs is not None:
        _dev_f32(probs, "probs")
        if probs.numel() != total:
            raise _hip.HipError("rpn_decode_clip: probs has %d elements, expected %d" % (probs.numel(), total))
    if deltas is not None:
        _dev_f32(deltas, "deltas")
        if deltas.numel() != total * 4:
            raise _hip.HipError("rpn_decode_clip: deltas has %d elements, expected %d" % (deltas.numel(), total * 4))
    scores = torch.empty((total,), dtype=torch.float32, device=anchors.device)
    props = torch.empty((total, 4), dtype=torch.float32, device=anchors.device)
    info_arr = _hip.float_array([float(v) for v in info[:4]])
    _hip.check(lib.frcnn_rpn_decode_clip(_ptr(rpn), ld, _ptr(probs), _ptr(deltas), _ptr(anchors), info_arr, hw,
                                         num_anchors, _ptr(scores), _ptr(props), _stream()), "frcnn_rpn_decode_clip")
    return scores, props


def bbox_transform_inv(boxes, deltas, scale=None):
    """boxes (N, >=4) [x1,y1,x2,y2,...], deltas (N, 4K) -> (N, 4K) decoded boxes."""
    lib = _hip.load()
    _dev_f32(boxes, "boxes"); _dev_f32(deltas, "deltas")
    n = boxes.shape[0]
    if n == 0:
        return deltas.detach() * 0  # reference: bbox_transform.py:79-80
    k = deltas.shape[1] // 4
    out = torch.empty((n, 4 * k), dtype=torch.float32, device=boxes.device)
    _hip.check(lib.frcnn_bbox_transform_inv(_ptr(boxes), boxes.shape[1], _ptr(deltas), n, k,
                                            float(scale) if scale is not None else 0.0, _ptr(out), _stream()),
               "frcnn_bbox_transform_inv")
    return out


def lidar_bbox_transform_inv(rois, anchors_3d, deltas, scale=None):
    """rois (N, >=4) [x1,y1,x2,y2,...], anchors_3d (N,7), deltas (N,7K) -> (N,7K) [xc,yc,zc,l,w,h,ry] per class."""
    lib = _hip.load()
    _dev_f32(rois, "rois"); _dev_f32(anchors_3d, "anchors_3d"); _dev_f32(deltas, "deltas")
    n = rois.shape[0]
    if n == 0:
        return deltas.detach() * 0  # reference: bbox_transform.py:181-182
    k = deltas.shape[1] // 7
    out = torch.empty((n, 7 * k), dtype=torch.float32, device=rois.device)
    _hip.check(lib.frcnn_lidar_bbox_transform_inv(_ptr(rois), rois.shape[1], _ptr(anchors_3d), _ptr(deltas), n, k,
                                                  float(scale) if scale is not None else 0.0, _ptr(out), _stream()),
               "frcnn_lidar_bbox_transform_inv")
    return out


def uncertainty_transform_inv(rois, uncertainty, anchors_3d=None, scale=None, lidar=False, input_is_variance=False):
    """uncertainty (N, 7K) of the 7-element deltas -> squared box-space terms: (N, 4K) [x,y,l,w] (lidar False) or (N, 7K)."""
    lib = _hip.load()
    _dev_f32(rois, "rois"); _dev_f32(uncertainty, "uncertainty")
    n = rois.shape[0]
    if uncertainty.shape[0] != n or uncertainty.shape[1] % 7:
        raise _hip.HipError("uncertainty_transform_inv: uncertainty must be (N, 7K), got %s" % (tuple(uncertainty.shape),))
    k = uncertainty.shape[1] // 7
    out = torch.empty((n, (7 if lidar else 4) * k), dtype=torch.float32, device=rois.device)
    if lidar:
        _dev_f32(anchors_3d, "anchors_3d")
    if n:
        _hip.check(lib.frcnn_uncertainty_transform_inv(_ptr(rois), rois.shape[1], _ptr(anchors_3d) if lidar else None,
                                                       _ptr(uncertainty), n, k, float(scale) if scale is not None else 0.0,
                                                       1 if lidar else 0, 1 if input_is_variance else 0, _ptr(out), _stream()),
                   "frcnn_uncertainty_transform_inv")
    return out


def clip_boxes(boxes, info):
    lib = _hip.load()
    _dev_f32(boxes, "boxes")
    out = torch.empty_like(boxes)
    if boxes.numel() == 0:
        return out
    _hip.check(lib.frcnn_clip_boxes(_ptr(boxes), boxes.numel() // 4, _hip.float_array([float(v) for v in info[:4]]),
                                    _ptr(out), _stream()), "frcnn_clip_boxes")
    return out


def sort_topk_desc(scores, top_n):
    """Returns (order int64 (top_n,), sorted_scores (top_n,), count int32 (1,)) — (score desc, index asc)."""
    lib = _hip.load()
    _dev_f32(scores, "scores")
    n = scores.numel()
    m = min(n, top_n)
    order = torch.empty((m,), dtype=torch.int64, device=scores.device)
    sout = torch.empty((m,), dtype=torch.float32, device=scores.device)
    count = torch.empty((1,), dtype=torch.int32, device=scores.device)
    ws_bytes = lib.frcnn_sort_topk_desc_ws_bytes(n, top_n)
    ws = _workspace(ws_bytes, scores.device) if ws_bytes else None
    _hip.check(lib.frcnn_sort_topk_desc(_ptr(scores), n, top_n, _ptr(order), _ptr(sout), _ptr(count), _ptr(ws), ws_bytes,
                                        _stream()), "frcnn_sort_topk_desc")
    return order, sout, count


def gather_rows(rows, order, count=None):
    lib = _hip.load()
    _dev_f32(rows, "rows")
    width = rows.shape[1] if rows.dim() > 1 else 1
    m = order.numel()
    out = torch.empty((m, width), dtype=torch.float32, device=rows.device)
    _hip.check(lib.frcnn_gather_rows(_ptr(rows), _ptr(order), _ptr(count), m, width, _ptr(out), _stream()),
               "frcnn_gather_rows")
    return out


def set_nms_suppress_at_equal(on):
    """IoU == threshold exactly: True (default) suppresses like torchvision 0.4.0's CPU kernel (`>=`), False keeps the box
    like its CUDA kernel (`>`).  Returns the previous setting."""
    lib = _hip.load()
    old = bool(lib.frcnn_nms_get_suppress_at_equal())
    _hip.check(lib.frcnn_nms_set_suppress_at_equal(1 if on else 0), "frcnn_nms_set_suppress_at_equal")
    return old


def nms_suppress_at_equal():
    """The current rule at IoU == threshold exactly (frcnn_nms_get_suppress_at_equal).  The rule is read at LAUNCH time, so
    a captured frame keeps the rule of its capture: model/frame_graph.cfg_fingerprint keys the captured frames by it."""
    return bool(_hip.load().frcnn_nms_get_suppress_at_equal())


def nms_sorted(boxes, thresh, max_keep=None, n_dev=None, want_mask=False):
    """NMS over boxes already in descending-score order.
    Returns (keep_idx int64 (max_keep,), keep_count int32 (1,), keep_mask uint8 (n,) or None)."""
    lib = _hip.load()
    _dev_f32(boxes, "boxes")
    n = boxes.shape[0]
    max_keep = n if max_keep is None or max_keep <= 0 else min(max_keep, n)
    keep_idx = torch.empty((max_keep,), dtype=torch.int64, device=boxes.device)     # the kernel zeroes the unused tail
    keep_count = torch.empty((1,), dtype=torch.int32, device=boxes.device)
    keep_mask = torch.empty((n,), dtype=torch.uint8, device=boxes.device) if want_mask else None
    ws_bytes = lib.frcnn_nms_ws_bytes(n)
    ws = _workspace(ws_bytes, boxes.device)
    _hip.check(lib.frcnn_nms(_ptr(boxes), _ptr(n_dev), n, float(thresh), max_keep, _ptr(keep_idx), _ptr(keep_mask),
                             _ptr(keep_count), _ptr(ws), ws_bytes, _stream()), "frcnn_nms")
    return keep_idx, keep_count, keep_mask


def make_rois(sorted_boxes, sorted_scores, keep_idx, keep_count):
    lib = _hip.load()
    m = keep_idx.numel()
    rois = torch.empty((m, 5), dtype=torch.float32, device=sorted_boxes.device)
    roi_scores = torch.empty((m, 1), dtype=torch.float32, device=sorted_boxes.device)
    _hip.check(lib.frcnn_make_rois(_ptr(sorted_boxes), _ptr(sorted_scores), _ptr(keep_idx), _ptr(keep_count), m,
                                   _ptr(rois), _ptr(roi_scores), _stream()), "frcnn_make_rois")
    return rois, roi_scores


def roi_align_nhwc(feat, rois, pooled, spatial_scale, sampling_ratio=0, roi_count=None, level_of_roi=None, level=-1,
                   out=None, rois_per_image=0, scale=None, shift=None, relu=False):
    """feat (N,H,W,C), rois (R,5) -> (R, P, P, C).  ``rois_per_image`` > 0: rows [i*rpi, (i+1)*rpi) belong to image i and
    ``roi_count`` holds N live counts (frames batched into one call); 0: the image is the RoI's batch column.
    ``scale`` / ``shift`` (C,) / ``relu``: per-channel epilogue on the pooled values, out = act(pooled * scale + shift)
    (frcnn_roi_align_fwd_affine)."""
    lib = _hip.load()
    _dev_f32(feat, "feat"); _dev_f32(rois, "rois")
    n, h, w, c = feat.shape
    r = rois.shape[0]
    for nm, t in (("scale", scale), ("shift", shift)):
        if t is not None:
            _dev_f32(t, nm)
            if t.numel() != c:
                raise _hip.HipError("roi_align_nhwc: %s has %d elements, expected %d" % (nm, t.numel(), c))
    if out is None:
        out = torch.empty((r, pooled, pooled, c), dtype=torch.float32, device=feat.device)
    ws_bytes = lib.frcnn_roi_align_fwd_ws_bytes(h, w, c, r, pooled)
    ws = _workspace(ws_bytes, feat.device) if ws_bytes else None
    if scale is not None or shift is not None or relu:
        _hip.check(lib.frcnn_roi_align_fwd_affine(_ptr(feat), n, h, w, c, _ptr(rois), _ptr(roi_count), r, int(rois_per_image),
                                                  pooled, float(spatial_scale), int(sampling_ratio), _ptr(level_of_roi), level,
                                                  _ptr(out), _ptr(scale), _ptr(shift), int(bool(relu)), _ptr(ws), ws_bytes,
                                                  _stream()), "frcnn_roi_align_fwd_affine")
        return out
    _hip.check(lib.frcnn_roi_align_fwd(_ptr(feat), n, h, w, c, _ptr(rois), _ptr(roi_count), r, int(rois_per_image), pooled,
                                       float(spatial_scale), int(sampling_ratio), _ptr(level_of_roi), level, _ptr(out),
                                       _ptr(ws), ws_bytes, _stream()), "frcnn_roi_align_fwd")
    return out


def roi_align_split(feat, rois, pooled, spatial_scale, split_c, sampling_ratio=0, roi_count=None, scale=None, shift=None,
                    relu1=False, relu2=False):
    """(out1 (R,P,P,split_c), out2 (R,P,P,C - split_c)) = RoIAlign of feat (1,H,W,C) over the two channel ranges with one
    plan (frcnn_roi_align_fwd_split); ``scale`` / ``shift`` (C,) are applied after the pooling, ReLU per output."""
    lib = _hip.load()
    _dev_f32(feat, "feat"); _dev_f32(rois, "rois")
    n, h, w, c = feat.shape
    if n != 1 or rois.shape[1] != 5:
        raise _hip.HipError("roi_align_split: one image, rois (R,5)")
    for nm, t in (("scale", scale), ("shift", shift)):
        if t is not None:
            _dev_f32(t, nm)
            if t.numel() != c:
                raise _hip.HipError("roi_align_split: %s has %d elements, expected %d" % (nm, t.numel(), c))
    r = rois.shape[0]
    out1 = torch.empty((r, pooled, pooled, split_c), dtype=torch.float32, device=feat.device)
    out2 = torch.empty((r, pooled, pooled, c - split_c), dtype=torch.float32, device=feat.device)
    ws_bytes = lib.frcnn_roi_align_fwd_ws_bytes(h, w, 4, r, pooled)
    ws = _workspace(ws_bytes, feat.device)
    _hip.check(lib.frcnn_roi_align_fwd_split(_ptr(feat), h, w, c, _ptr(rois), _ptr(roi_count), r, pooled, float(spatial_scale),
                                             int(sampling_ratio), int(split_c), _ptr(out1), _ptr(out2), _ptr(scale), _ptr(shift),
                                             int(bool(relu1)), int(bool(relu2)), _ptr(ws), ws_bytes, _stream()),
               "frcnn_roi_align_fwd_split")
    return out1, out2


def head_fc_softmax_decode(x, w_cls, b_cls, w_box, b_box, rois, stds, means, scale, roi_anchors_3d=None):
    """x (R,P,P,C) -> dict(fc7, cls_score, cls_prob, bbox_pred, pred_boxes).  With ``roi_anchors_3d`` (R,7) the
    boxes are the 7-DoF LiDAR boxes (frcnn_head_fc_softmax_decode_lidar), otherwise 4-DoF image boxes."""
    lib = _hip.load()
    for nm, t in (("x", x), ("w_cls", w_cls), ("b_cls", b_cls), ("w_box", w_box), ("b_box", b_box), ("rois", rois)):
        _dev_f32(t, nm)
    r, p, _, c = x.shape
    k = w_cls.shape[0]
    e = 4 if roi_anchors_3d is None else 7
    if w_box.shape[0] != e * k or w_cls.shape[1] != c or w_box.shape[1] != c or len(stds) != e or len(means) != e:
        raise _hip.HipError("head_fc_softmax_decode: inconsistent head weights / normalisation for %d-DoF boxes" % e)
    dev = x.device
    fc7 = torch.empty((r, c), dtype=torch.float32, device=dev)
    cls_score = torch.empty((r, k), dtype=torch.float32, device=dev)
    cls_prob = torch.empty((r, k), dtype=torch.float32, device=dev)
    bbox_pred = torch.empty((r, e * k), dtype=torch.float32, device=dev)
    pred_boxes = torch.empty((r, e * k), dtype=torch.float32, device=dev)
    if e == 4:
        _hip.check(lib.frcnn_head_fc_softmax_decode(
            _ptr(x), r, p, c, _ptr(w_cls), _ptr(b_cls), _ptr(w_box), _ptr(b_box), k, _ptr(rois),
            _hip.float_array(stds), _hip.float_array(means), float(scale), _ptr(fc7), _ptr(cls_score), _ptr(cls_prob),
            _ptr(bbox_pred), _ptr(pred_boxes), _stream()), "frcnn_head_fc_softmax_decode")
    else:
        _dev_f32(roi_anchors_3d, "roi_anchors_3d")
        if tuple(roi_anchors_3d.shape) != (r, 7):
            raise _hip.HipError("head_fc_softmax_decode: roi_anchors_3d must be (%d, 7)" % r)
        _hip.check(lib.frcnn_head_fc_softmax_decode_lidar(
            _ptr(x), r, p, c, _ptr(w_cls), _ptr(b_cls), _ptr(w_box), _ptr(b_box), k, _ptr(rois), _ptr(roi_anchors_3d),
            _hip.float_array(stds), _hip.float_array(means), float(scale), _ptr(fc7), _ptr(cls_score), _ptr(cls_prob),
            _ptr(bbox_pred), _ptr(pred_boxes), _stream()), "frcnn_head_fc_softmax_decode_lidar")
    return {"fc7": fc7, "cls_score": cls_score, "cls_prob": cls_prob, "bbox_pred": bbox_pred, "pred_boxes": pred_boxes}


def generate_anchors_3d(base, height, width, feat_stride):
    """base: DEVICE float32 (T, 9) anchor-type table -> (anchors_3d (H*W*T, 7), anchors_2d (H*W*T, 4))."""
    lib = _hip.load()
    _dev_f32(base, "base")
    t = base.shape[0]
    a3 = torch.empty((height * width * t, 7), dtype=torch.float32, device=base.device)
    a2 = torch.empty((height * width * t, 4), dtype=torch.float32, device=base.device)
    _hip.check(lib.frcnn_generate_anchors_3d(_ptr(base), t, height, width, feat_stride, _ptr(a3), _ptr(a2), _stream()),
               "frcnn_generate_anchors_3d")
    return a3, a2


def filter_per_class_lidar(pred_boxes, cls_prob, thresh, nms_thresh, max_dets, max_out=None, roi_count=None,
                           want_rois=False):
    """7-DoF form: returns (dets (K, max_out, 8) [xc,yc,zc,l,w,h,ry,score], det_count int32 (K,)[, det_roi int32
    (K, max_out)])."""
    lib = _hip.load()
    _dev_f32(pred_boxes, "pred_boxes"); _dev_f32(cls_prob, "cls_prob")
    r, k = cls_prob.shape
    if pred_boxes.shape[1] != 7 * k:
        raise _hip.HipError("filter_per_class_lidar: pred_boxes must be (R, 7K)")
    max_out = r if max_out is None else max_out
    dets = torch.empty((k, max_out, 8), dtype=torch.float32, device=cls_prob.device)    # every row is written by the call
    det_count = torch.empty((k,), dtype=torch.int32, device=cls_prob.device)
    ws_bytes = lib.frcnn_filter_per_class_ws_bytes(r, k)
    ws = _workspace(ws_bytes, cls_prob.device)
    det_roi = torch.empty((k, max_out), dtype=torch.int32, device=cls_prob.device) if want_rois else None   # -1 = no detection
    _hip.check(lib.frcnn_filter_per_class_lidar(_ptr(pred_boxes), _ptr(cls_prob), _ptr(roi_count), r, k, float(thresh),
                                                float(nms_thresh), int(max_dets), int(max_out), _ptr(dets),
                                                _ptr(det_count), _ptr(det_roi), _ptr(ws), ws_bytes, _stream()),
               "frcnn_filter_per_class_lidar")
    return (dets, det_count, det_roi) if want_rois else (dets, det_count)


def filter_per_class(pred_boxes, cls_prob, frame_w, frame_h, scale, thresh, nms_thresh, max_dets, max_out=None,
                     roi_count=None, want_rois=False):
    """In-place clamp of pred_boxes + per-class threshold/NMS/max_dets.
    Returns (dets (K, max_out, 5), det_count int32 (K,)[, det_roi int32 (K, max_out): RoI row of each detection])."""
    lib = _hip.load()
    _dev_f32(pred_boxes, "pred_boxes"); _dev_f32(cls_prob, "cls_prob")
    r, k = cls_prob.shape
    max_out = r if max_out is None else max_out
    dets = torch.empty((k, max_out, 5), dtype=torch.float32, device=cls_prob.device)    # every row is written by the call
    det_count = torch.empty((k,), dtype=torch.int32, device=cls_prob.device)
    ws_bytes = lib.frcnn_filter_per_class_ws_bytes(r, k)
    ws = _workspace(ws_bytes, cls_prob.device)
    det_roi = torch.empty((k, max_out), dtype=torch.int32, device=cls_prob.device) if want_rois else None   # -1 = no detection
    _hip.check(lib.frcnn_filter_per_class(_ptr(pred_boxes), _ptr(cls_prob), _ptr(roi_count), r, k, float(frame_w),
                                          float(frame_h), float(scale), float(thresh), float(nms_thresh), int(max_dets),
                                          int(max_out), _ptr(dets), _ptr(det_count), _ptr(det_roi), _ptr(ws), ws_bytes,
                                          _stream()), "frcnn_filter_per_class")
    return (dets, det_count, det_roi) if want_rois else (dets, det_count)


def bev_voxelize(points, pc_range, voxel_size, z_shift, max_points, max_voxels, num_slices, num_meta,
                 elongation_col=-1):
    """Point cloud (N, F>=4) device tensor -> BEV map (gy, gx, num_slices+num_meta) and the device count of occupied
    cells (frcnn_bev_voxelize; lib/roi_data_layer/minibatch.py:434-512)."""
    lib = _hip.load()
    _dev_f32(points, "points")
    if points.dim() != 2 or points.shape[1] < 4 or points.shape[0] == 0:
        raise _hip.HipError("bev_voxelize: points must be (N>0, F>=4), got %s" % (tuple(points.shape),))
    rng, vs = _hip.float_array(pc_range), _hip.float_array(voxel_size)
    grid = (ctypes.c_int * 3)()
    _hip.check(lib.frcnn_bev_voxelize_grid(rng, vs, grid), "frcnn_bev_voxelize_grid")
    gx, gy = int(grid[0]), int(grid[1])
    bev = torch.empty((gy, gx, int(num_slices) + int(num_meta)), dtype=torch.float32, device=points.device)
    count = torch.zeros((1,), dtype=torch.int32, device=points.device)
    nbytes = lib.frcnn_bev_voxelize_ws_bytes(points.shape[0], rng, vs, int(max_voxels))
    ws = _workspace(nbytes, points.device)
    _hip.check(lib.frcnn_bev_voxelize(_ptr(points), points.shape[0], points.shape[1], rng, vs, float(z_shift),
                                      int(max_points), int(max_voxels), int(num_slices), int(num_meta),
                                      int(elongation_col), _ptr(bev), _ptr(count), _ptr(ws), nbytes, _stream()),
               "frcnn_bev_voxelize")
    return bev, count


# ----------------------------------------------------------------------------------------------
# training path
# ----------------------------------------------------------------------------------------------
def act_bwd(dy, y=None, scale=None, relu=False, want_res=False):
    """Backward of act(conv*scale + shift + res): returns (d_conv, d_res or None)."""
    lib = _hip.load()
    _dev_f32(dy, "dy")
    k = dy.shape[-1]
    rows = dy.numel() // k
    if relu:
        _dev_f32(y, "y")
    d_conv = torch.empty_like(dy)
    d_res = torch.empty_like(dy) if want_res else None
    _hip.check(lib.frcnn_act_bwd(_ptr(dy), _ptr(y) if relu else None, _ptr(scale), int(bool(relu)), rows, k, _ptr(d_conv),
                                 _ptr(d_res), _stream()), "frcnn_act_bwd")
    return d_conv, d_res


# True: the statistics / coefficient pass of frcnn_bn_train_fwd / _bwd runs inside the column-sum kernel (tile counters as for
# the filter gradients); False: as a launch of its own (A/B, and the arithmetic is the same code: bit-identical)
BN_FUSED_FINAL = os.environ.get('FRCNN_BN_FUSED_FINAL', '1') != '0'


def bn_train_fwd(y, gamma, beta, eps, momentum, running_mean=None, running_var=None, residual=None, relu=False):
    """BatchNorm with batch statistics over the rows of an NHWC tensor (+ residual, ReLU); the running statistics are
    updated in place.  Returns (out, save_mean, save_invstd)."""
    lib = _hip.load()
    _dev_f32(y, "y")
    c = y.shape[-1]
    rows = y.numel() // c
    for nm, t in (("gamma", gamma), ("beta", beta), ("running_mean", running_mean), ("running_var", running_var)):
        if t is not None:
            _dev_f32(t, nm)
            if t.numel() != c:
                raise _hip.HipError("bn_train_fwd: %s has %d elements, expected %d" % (nm, t.numel(), c))
    if residual is not None:
        _dev_f32(residual, "residual")
        if residual.shape != y.shape:
            raise _hip.HipError("bn_train_fwd: residual %s vs %s" % (tuple(residual.shape), tuple(y.shape)))
    out = torch.empty_like(y)
    mean = torch.empty((c,), dtype=torch.float32, device=y.device)
    invstd = torch.empty_like(mean)
    nbytes = lib.frcnn_bn_train_ws_bytes(c)
    ws = _workspace(nbytes, y.device)
    counters = _tile_counters(y.device, lib.frcnn_bn_train_counters(c)) if BN_FUSED_FINAL else None
    _hip.check(lib.frcnn_bn_train_fwd(_ptr(y), rows, c, _ptr(gamma), _ptr(beta), float(eps), float(momentum),
                                      _ptr(running_mean), _ptr(running_var), _ptr(residual), int(bool(relu)), _ptr(out),
                                      _ptr(mean), _ptr(invstd), _ptr(ws), nbytes, _ptr(counters), _stream()),
               "frcnn_bn_train_fwd")
    return out, mean, invstd


def bn_train_bwd(dout, out, y, gamma, save_mean, save_invstd, relu=False, want_res=False, grad_gamma=None, grad_beta=None):
    """Backward of bn_train_fwd: returns (dy, dres or None, dgamma, dbeta).  ``grad_gamma`` / ``grad_beta`` (both or neither):
    the parameters' own (c,) gradient buffers - the sums are ADDED into them and returned as dgamma / dbeta."""
    lib = _hip.load()
    _dev_f32(dout, "dout"); _dev_f32(y, "y"); _dev_f32(save_mean, "save_mean"); _dev_f32(save_invstd, "save_invstd")
    if relu:
        _dev_f32(out, "out")
    c = y.shape[-1]
    rows = y.numel() // c
    if dout.shape != y.shape or save_mean.numel() != c or save_invstd.numel() != c:
        raise _hip.HipError("bn_train_bwd: shape mismatch")
    dy = torch.empty_like(y)
    dres = torch.empty_like(y) if want_res else None
    accumulate = grad_gamma is not None
    if accumulate:
        if grad_beta is None or grad_gamma.numel() != c or grad_beta.numel() != c:
            raise _hip.HipError("bn_train_bwd: grad_gamma and grad_beta must both be (c,) gradient buffers")
        dgamma, dbeta = _dev_f32(grad_gamma, "grad_gamma"), _dev_f32(grad_beta, "grad_beta")
    else:
        dgamma = torch.empty((c,), dtype=torch.float32, device=y.device)
        dbeta = torch.empty_like(dgamma)
    nbytes = lib.frcnn_bn_train_ws_bytes(c)
    ws = _workspace(nbytes, y.device)
    counters = _tile_counters(y.device, lib.frcnn_bn_train_counters(c)) if BN_FUSED_FINAL else None
    _hip.check(lib.frcnn_bn_train_bwd(_ptr(dout), _ptr(out) if relu else None, _ptr(y), rows, c, _ptr(gamma),
                                      _ptr(save_mean), _ptr(save_invstd), int(bool(relu)), _ptr(dy), _ptr(dres),
                                      _ptr(dgamma), _ptr(dbeta), int(accumulate), _ptr(ws), nbytes, _ptr(counters), _stream()),
               "frcnn_bn_train_bwd")
    return dy, dres, dgamma, dbeta


def upsample_bilinear_add(x, lateral):
    """F.interpolate(x, size=lateral.shape[1:3], mode='bilinear', align_corners=False) + lateral, NHWC."""
    lib = _hip.load()
    _dev_f32(x, "x"); _dev_f32(lateral, "lateral")
    n, h, w, c = x.shape
    n2, oh, ow, c2 = lateral.shape
    if n2 != n or c2 != c:
        raise _hip.HipError("upsample_bilinear_add: %s vs %s" % (tuple(x.shape), tuple(lateral.shape)))
    out = torch.empty_like(lateral)
    _hip.check(lib.frcnn_upsample_bilinear_add_fwd(_ptr(x), _ptr(lateral), _ptr(out), n, h, w, oh, ow, c, _stream()),
               "frcnn_upsample_bilinear_add_fwd")
    return out


def upsample_bilinear_bwd(dout, in_hw):
    lib = _hip.load()
    _dev_f32(dout, "dout")
    n, oh, ow, c = dout.shape
    h, w = in_hw
    dx = torch.empty((n, h, w, c), dtype=torch.float32, device=dout.device)
    _hip.check(lib.frcnn_upsample_bilinear_bwd(_ptr(dout), _ptr(dx), n, h, w, oh, ow, c, _stream()),
               "frcnn_upsample_bilinear_bwd")
    return dx


ROI_ALIGN_BWD_PLANNED = True      # False: always the sample-by-sample backward (frcnn_roi_align_bwd)


def roi_align_bwd(dout, feat_shape, rois, spatial_scale, sampling_ratio=0, roi_count=None, level_of_roi=None, level=-1,
                  dfeat=None):
    """Scatter dout (R,P,P,C) into dfeat (1,H,W,C) (created zero-filled unless given, then accumulated into).  7x7 bins and
    C % 4 == 0 go through the forward's plan (frcnn_roi_align_bwd_planned), anything else sample by sample."""
    lib = _hip.load()
    _dev_f32(dout, "dout"); _dev_f32(rois, "rois")
    _, h, w, c = feat_shape
    r, p = dout.shape[0], dout.shape[1]
    if dfeat is None:
        dfeat = torch.zeros(tuple(feat_shape), dtype=torch.float32, device=dout.device)
    if ROI_ALIGN_BWD_PLANNED and feat_shape[0] == 1 and c % 4 == 0:
        ws_bytes = lib.frcnn_roi_align_fwd_ws_bytes(h, w, c, r, p)
        if ws_bytes:
            ws = _workspace(ws_bytes, dout.device)
            _hip.check(lib.frcnn_roi_align_bwd_planned(_ptr(dout), h, w, c, _ptr(rois), _ptr(roi_count), r, p,
                                                       float(spatial_scale), int(sampling_ratio), _ptr(level_of_roi), int(level),
                                                       _ptr(dfeat), _ptr(ws), ws_bytes, _stream()),
                       "frcnn_roi_align_bwd_planned")
            return dfeat
    _hip.check(lib.frcnn_roi_align_bwd(_ptr(dout), h, w, c, _ptr(rois), _ptr(roi_count), r, p, float(spatial_scale),
                                       int(sampling_ratio), _ptr(level_of_roi), int(level), _ptr(dfeat), _stream()),
               "frcnn_roi_align_bwd")
    return dfeat


def labelled_pixels(labels, hw, num_anchors, cap):
    """labels (hw * A,) in (H,W,A) order -> (idx int64 (cap,), count int32 (2,) = [min(total, cap), total]): the pixels that
    carry at least one anchor with a label != -1, ascending; -1 beyond the count (frcnn_labelled_pixels)."""
    lib = _hip.load()
    _dev_f32(labels, "labels")
    if labels.numel() != hw * num_anchors:
        raise _hip.HipError("labelled_pixels: labels has %d elements, expected %d x %d" % (labels.numel(), hw, num_anchors))
    idx = torch.empty((cap,), dtype=torch.int64, device=labels.device)
    count = torch.empty((2,), dtype=torch.int32, device=labels.device)
    ws_bytes = lib.frcnn_labelled_pixels_ws_bytes(hw)
    ws = _workspace(ws_bytes, labels.device)
    _hip.check(lib.frcnn_labelled_pixels(_ptr(labels), hw, num_anchors, cap, _ptr(idx), _ptr(count), _ptr(ws), ws_bytes,
                                         _stream()), "frcnn_labelled_pixels")
    return idx, count


def gather_patches(x, idx, count, r, s, pad):
    """x (1,H,W,C), idx (cap,) pixels -> (cap, r, s, C): the r x s windows around the listed pixels (frcnn_gather_patches)."""
    lib = _hip.load()
    _dev_f32(x, "x")
    n, h, w, c = x.shape
    if n != 1:
        raise _hip.HipError("gather_patches: one image per call")
    cap = idx.numel()
    out = torch.empty((cap, r, s, c), dtype=torch.float32, device=x.device)
    _hip.check(lib.frcnn_gather_patches(_ptr(x), h, w, c, _ptr(idx), _ptr(count), cap, r, s, pad, _ptr(out), _stream()),
               "frcnn_gather_patches")
    return out


def scatter_add_patches(d, idx, count, h, w, pad, out=None):
    """Adjoint of gather_patches: d (cap, r, s, C) -> dx (1,h,w,C) (zero-filled here unless ``out`` is given; float atomics)."""
    lib = _hip.load()
    _dev_f32(d, "d")
    cap, r, s, c = d.shape
    if out is None:
        out = torch.zeros((1, h, w, c), dtype=torch.float32, device=d.device)
    _hip.check(lib.frcnn_scatter_add_patches(_ptr(d), h, w, c, _ptr(idx), _ptr(count), cap, r, s, pad, _ptr(out), _stream()),
               "frcnn_scatter_add_patches")
    return out


def rpn_loss(rpn, num_anchors, labels, targets, inside, outside, grad_ce=1.0, grad_box=1.0, want_grad=True):
    """rpn (HW, ld) fused head output.  Returns (losses (3,) [ce, box, count], drpn or None)."""
    lib = _hip.load()
    for nm, t in (("rpn", rpn), ("labels", labels), ("targets", targets), ("inside", inside), ("outside", outside)):
        _dev_f32(t, nm)
    hw, ld = rpn.shape
    losses = torch.empty((3,), dtype=torch.float32, device=rpn.device)
    drpn = torch.empty_like(rpn) if want_grad else None
    ws_bytes = lib.frcnn_rpn_loss_ws_bytes()
    ws = _workspace(ws_bytes, rpn.device)
    _hip.check(lib.frcnn_rpn_loss(_ptr(rpn), ld, num_anchors, hw, _ptr(labels), _ptr(targets), _ptr(inside), _ptr(outside),
                                  float(grad_ce), float(grad_box), _ptr(losses), _ptr(drpn), _ptr(ws), ws_bytes, _stream()),
               "frcnn_rpn_loss")
    return losses, drpn


def det_loss(cls_score, labels, bbox_pred, targets, inside, outside, bbox_elem=4, grad_ce=1.0, grad_box=1.0,
             want_grad=True, lidar=None):
    """Returns (losses (2,) [ce, box], dcls, dbox).  ``lidar=(reg_loss_weight[7], ry_sin)`` selects the 7-element LiDAR
    form (sin() on the yaw difference, per-element weights)."""
    lib = _hip.load()
    for nm, t in (("cls_score", cls_score), ("labels", labels), ("bbox_pred", bbox_pred), ("targets", targets),
                  ("inside", inside), ("outside", outside)):
        _dev_f32(t, nm)
    r, k = cls_score.shape
    losses = torch.empty((2,), dtype=torch.float32, device=cls_score.device)
    dcls = torch.empty_like(cls_score) if want_grad else None
    dbox = torch.empty_like(bbox_pred) if want_grad else None
    if lidar is not None:
        weights, ry_sin = lidar
        if bbox_pred.shape[1] != 7 * k:
            raise _hip.HipError("det_loss: LiDAR bbox_pred must be (R, 7*K), got %s" % (tuple(bbox_pred.shape),))
        _hip.check(lib.frcnn_det_loss_lidar(_ptr(cls_score), _ptr(labels), r, k, _ptr(bbox_pred), _ptr(targets),
                                            _ptr(inside), _ptr(outside), _hip.float_array([float(v) for v in weights]),
                                            int(bool(ry_sin)), float(grad_ce), float(grad_box), _ptr(losses), _ptr(dcls),
                                            _ptr(dbox), _stream()), "frcnn_det_loss_lidar")
        return losses, dcls, dbox
    _hip.check(lib.frcnn_det_loss(_ptr(cls_score), _ptr(labels), r, k, _ptr(bbox_pred), _ptr(targets), _ptr(inside),
                                  _ptr(outside), int(bbox_elem), float(grad_ce), float(grad_box), _ptr(losses), _ptr(dcls),
                                  _ptr(dbox), _stream()), "frcnn_det_loss")
    return losses, dcls, dbox


def det_loss_aleatoric(cls_score, labels, bbox_pred, bbox_var, targets, inside, outside, bbox_elem=4, weights=None,
                       ry_sin=False, grad_ce=1.0, grad_box=1.0):
    """Detection losses with the aleatoric attenuation (loss_utils.py:82-85).  Returns (losses (2,), dcls, dbox, dvar)."""
    lib = _hip.load()
    for nm, t in (("cls_score", cls_score), ("labels", labels), ("bbox_pred", bbox_pred), ("bbox_var", bbox_var),
                  ("targets", targets), ("inside", inside), ("outside", outside)):
        _dev_f32(t, nm)
    r, k = cls_score.shape
    if bbox_pred.shape != (r, bbox_elem * k) or bbox_var.shape != bbox_pred.shape:
        raise _hip.HipError("det_loss_aleatoric: bbox_pred / bbox_var must be (R, %d*K)" % bbox_elem)
    losses = torch.empty((2,), dtype=torch.float32, device=cls_score.device)
    dcls, dbox, dvar = torch.empty_like(cls_score), torch.empty_like(bbox_pred), torch.empty_like(bbox_var)
    w = _hip.float_array([float(v) for v in weights]) if weights is not None else None
    _hip.check(lib.frcnn_det_loss_aleatoric(_ptr(cls_score), _ptr(labels), r, k, _ptr(bbox_pred), _ptr(bbox_var),
                                            _ptr(targets), _ptr(inside), _ptr(outside), int(bbox_elem), w,
                                            int(bool(ry_sin)), float(grad_ce), float(grad_box), _ptr(losses), _ptr(dcls),
                                            _ptr(dbox), _ptr(dvar), _stream()), "frcnn_det_loss_aleatoric")
    return losses, dcls, dbox, dvar


def mc_bbox_var(samples):
    """(T, ...) stack of stochastic passes -> unbiased variance over T, clamped at 0 (compute_bbox_var)."""
    lib = _hip.load()
    _dev_f32(samples, "samples")
    t = samples.shape[0]
    out = torch.empty(samples.shape[1:], dtype=torch.float32, device=samples.device)
    _hip.check(lib.frcnn_mc_bbox_var(_ptr(samples), t, out.numel(), _ptr(out), _stream()), "frcnn_mc_bbox_var")
    return out


def mc_cls_stats(cls_score_samples, want_var=False):
    """(T, N, K) logits -> (mean softmax (N,K), entropy (N,), mutual information (N,)[, variance of the softmax over T
    (N,K)]), log base 2."""
    lib = _hip.load()
    _dev_f32(cls_score_samples, "cls_score_samples")
    t, n, k = cls_score_samples.shape
    dev = cls_score_samples.device
    mean_prob = torch.empty((n, k), dtype=torch.float32, device=dev)
    entropy, mi = torch.empty((n,), dtype=torch.float32, device=dev), torch.empty((n,), dtype=torch.float32, device=dev)
    var = torch.empty((n, k), dtype=torch.float32, device=dev) if want_var else None
    _hip.check(lib.frcnn_mc_cls_stats(_ptr(cls_score_samples), t, n, k, _ptr(mean_prob), _ptr(entropy), _ptr(mi),
                                      _ptr(var), _stream()), "frcnn_mc_cls_stats")
    return (mean_prob, entropy, mi, var) if want_var else (mean_prob, entropy, mi)


def mc_mean(samples):
    """(T, ...) -> mean over T (summed in sample order)."""
    lib = _hip.load()
    _dev_f32(samples, "samples")
    out = torch.empty(samples.shape[1:], dtype=torch.float32, device=samples.device)
    _hip.check(lib.frcnn_mc_mean(_ptr(samples), samples.shape[0], out.numel(), _ptr(out), _stream()), "frcnn_mc_mean")
    return out


def _seed_dev(seed_dev):
    """Device word added to the scalar seed (a replayed hipGraph keeps its scalars): int32 / uint32 tensor of one element."""
    if seed_dev is None:
        return None
    if not seed_dev.is_cuda or seed_dev.numel() != 1 or seed_dev.element_size() != 4:
        raise _hip.HipError("seed_dev must be one 32-bit word on the device")
    return seed_dev.data_ptr()


def dropout(x, p, seed, stream_id, repeat=1, seed_dev=None):
    """nn.Dropout(p) in train() mode with counter-based masks; ``repeat`` stochastic copies: x (...) -> (repeat, ...)
    (the leading dimension is dropped for repeat == 1).  The draws use ``seed + *seed_dev``."""
    lib = _hip.load()
    _dev_f32(x, "x")
    shape = tuple(x.shape) if repeat == 1 else (repeat,) + tuple(x.shape)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    _hip.check(lib.frcnn_dropout_fwd(_ptr(x), x.numel(), int(repeat), float(p), int(seed) & 0xFFFFFFFF, _seed_dev(seed_dev),
                                     int(stream_id) & 0xFFFFFFFF, _ptr(y), _stream()), "frcnn_dropout_fwd")
    return y


def dropout_bwd(dy, p, seed, stream_id, repeat=1, seed_dev=None):
    lib = _hip.load()
    _dev_f32(dy, "dy")
    shape = tuple(dy.shape) if repeat == 1 else tuple(dy.shape[1:])
    dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
    _hip.check(lib.frcnn_dropout_bwd(_ptr(dy), dx.numel(), int(repeat), float(p), int(seed) & 0xFFFFFFFF, _seed_dev(seed_dev),
                                     int(stream_id) & 0xFFFFFFFF, _ptr(dx), _stream()), "frcnn_dropout_bwd")
    return dx


def logit_distort(score, var, num_samples, seed, stream_id, var_is_log=False, seed_dev=None):
    """logit_distort (loss_utils.py:143-147): (N,K) logits and (log-)variances -> ((S, N, K) distorted logits, (N,K)
    variances)."""
    lib = _hip.load()
    _dev_f32(score, "score"); _dev_f32(var, "var")
    if score.shape != var.shape:
        raise _hip.HipError("logit_distort: score %s vs var %s" % (tuple(score.shape), tuple(var.shape)))
    out = torch.empty((int(num_samples),) + tuple(score.shape), dtype=torch.float32, device=score.device)
    var_out = torch.empty_like(var)
    _hip.check(lib.frcnn_logit_distort(_ptr(score), _ptr(var), score.numel(), int(num_samples), int(seed) & 0xFFFFFFFF,
                                       _seed_dev(seed_dev), int(stream_id) & 0xFFFFFFFF, int(bool(var_is_log)), _ptr(out), _ptr(var_out),
                                       _stream()), "frcnn_logit_distort")
    return out, var_out


def exp(x):
    lib = _hip.load()
    _dev_f32(x, "x")
    y = torch.empty_like(x)
    _hip.check(lib.frcnn_exp(_ptr(x), x.numel(), _ptr(y), _stream()), "frcnn_exp")
    return y


def bayesian_cross_entropy(cls_score, cls_var, labels, num_samples, seed, stream_id, grad=1.0, want_grad=True,
                           var_is_log=False, seed_dev=None):
    """bayesian_cross_entropy (loss_utils.py:149-169).  Returns (loss (1,), dscore, dvar) (gradients scaled by grad)."""
    lib = _hip.load()
    _dev_f32(cls_score, "cls_score"); _dev_f32(cls_var, "cls_var"); _dev_f32(labels, "labels")
    n, k = cls_score.shape
    dev = cls_score.device
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    per_roi = torch.empty((n,), dtype=torch.float32, device=dev)
    dscore = torch.empty_like(cls_score) if want_grad else None
    dvar = torch.empty_like(cls_var) if want_grad else None
    _hip.check(lib.frcnn_bayesian_cross_entropy(_ptr(cls_score), _ptr(cls_var), _ptr(labels), n, k, int(num_samples),
                                                int(seed) & 0xFFFFFFFF, _seed_dev(seed_dev), int(stream_id) & 0xFFFFFFFF,
                                                int(bool(var_is_log)), float(grad), _ptr(loss), _ptr(per_roi), _ptr(dscore), _ptr(dvar), _stream()),
               "frcnn_bayesian_cross_entropy")
    return loss, dscore, dvar


def bbox_overlaps(boxes, query_boxes):
    """IoU (+1 convention) between boxes (N, >=4) and query_boxes (K, >=4) -> (N, K)."""
    lib = _hip.load()
    _dev_f32(boxes, "boxes"); _dev_f32(query_boxes, "query_boxes")
    n, k = boxes.shape[0], query_boxes.shape[0]
    out = torch.empty((n, k), dtype=torch.float32, device=boxes.device)
    if n and k:
        _hip.check(lib.frcnn_bbox_overlaps(_ptr(boxes), boxes.shape[1], n, _ptr(query_boxes), query_boxes.shape[1], k,
                                           _ptr(out), _stream()), "frcnn_bbox_overlaps")
    return out


def bbox_transform(ex_rois, gt_rois):
    """Row-wise regression targets (N,4) of gt_rois (N, >=4) against ex_rois (N, >=4) (bbox_transform.py:52-70)."""
    lib = _hip.load()
    _dev_f32(ex_rois, "ex_rois"); _dev_f32(gt_rois, "gt_rois")
    n = ex_rois.shape[0]
    if gt_rois.shape[0] != n:
        raise _hip.HipError("bbox_transform: %d ex_rois vs %d gt_rois" % (n, gt_rois.shape[0]))
    out = torch.empty((n, 4), dtype=torch.float32, device=ex_rois.device)
    if n:
        _hip.check(lib.frcnn_bbox_transform(_ptr(ex_rois), ex_rois.shape[1], _ptr(gt_rois), gt_rois.shape[1], n, _ptr(out),
                                            _stream()), "frcnn_bbox_transform")
    return out


def lidar_bbox_transform(ex_rois, ex_anchors_3d, gt_rois):
    """Row-wise 7-DoF regression targets (N,7) (bbox_transform.py:16-49)."""
    lib = _hip.load()
    _dev_f32(ex_rois, "ex_rois"); _dev_f32(ex_anchors_3d, "ex_anchors_3d"); _dev_f32(gt_rois, "gt_rois")
    n = ex_rois.shape[0]
    if gt_rois.shape[0] != n or ex_anchors_3d.shape[0] != n or ex_anchors_3d.shape[1] != 7:
        raise _hip.HipError("lidar_bbox_transform: row counts / anchor width do not match")
    out = torch.empty((n, 7), dtype=torch.float32, device=ex_rois.device)
    if n:
        _hip.check(lib.frcnn_lidar_bbox_transform(_ptr(ex_rois), ex_rois.shape[1], _ptr(ex_anchors_3d), _ptr(gt_rois),
                                                  gt_rois.shape[1], n, _ptr(out), _stream()), "frcnn_lidar_bbox_transform")
    return out


def _gt_count(gt_count):
    if gt_count is None:
        return None
    if not gt_count.is_cuda or gt_count.dtype != torch.int32 or gt_count.numel() != 1:
        raise _hip.HipError("gt_count must be a one-element int32 device tensor")
    return gt_count.data_ptr()


def anchor_target_layer(anchors, gt_boxes, info, rpn_batchsize, fg_fraction, neg_ov, pos_ov, seed, seed_dev=None,
                        gt_count=None):
    """Returns labels (N,), targets/inside/outside (N,4) in anchor order and counts (2,) int32 [fg, bg candidates].
    ``gt_count`` (int32 device tensor): live rows of a ``gt_boxes`` buffer padded to its row count."""
    lib = _hip.load()
    _dev_f32(anchors, "anchors"); _dev_f32(gt_boxes, "gt_boxes")
    n, g = anchors.shape[0], gt_boxes.shape[0]
    if gt_boxes.shape[1] != 5:
        raise _hip.HipError("anchor_target_layer: gt_boxes must be (G,5)")
    dev = anchors.device
    labels = torch.empty((n,), dtype=torch.float32, device=dev)
    targets, inside, outside = (torch.empty((n, 4), dtype=torch.float32, device=dev) for _ in range(3))
    counts = torch.zeros((2,), dtype=torch.int32, device=dev)
    ws_bytes = lib.frcnn_anchor_target_layer_ws_bytes(n, g, int(rpn_batchsize))
    ws = _workspace(ws_bytes, dev)
    _hip.check(lib.frcnn_anchor_target_layer(
        _ptr(anchors), n, _ptr(gt_boxes), g, _gt_count(gt_count), _hip.float_array([float(v) for v in list(info)[:4]]),
        int(rpn_batchsize),
        float(fg_fraction), float(neg_ov), float(pos_ov), int(seed) & 0xFFFFFFFF, _ptr(seed_dev), _ptr(labels), _ptr(targets),
        _ptr(inside), _ptr(outside), _ptr(counts), _ptr(ws), ws_bytes, _stream()), "frcnn_anchor_target_layer")
    return labels, targets, inside, outside, counts


def proposal_target_layer(rois, roi_scores, gt_boxes, num_classes, rois_per_frame, fg_fraction, fg_thresh, bg_hi, bg_lo,
                          means, stds, seed, roi_count=None, anchors_3d=None, true_gt_boxes=None, skip_mask=None, seed_dev=None,
                          gt_count=None):
    """Returns dict(labels (R,), rois (R,5), scores (R,), targets/inside/outside (R,4K), assign int32 (R,), counts int32 (4,)).
    With ``anchors_3d`` (num_rois,7) and ``true_gt_boxes`` (G,8) the LiDAR form: 7K-wide targets and ``anchors_3d`` (R,7)."""
    lib = _hip.load()
    _dev_f32(rois, "rois"); _dev_f32(gt_boxes, "gt_boxes")
    if roi_scores is not None:
        _dev_f32(roi_scores, "roi_scores")
    dev = rois.device
    r = int(rois_per_frame)
    if skip_mask is not None and (not skip_mask.is_cuda or skip_mask.dtype != torch.uint8 or
                                  skip_mask.numel() != rois.shape[0] or not skip_mask.is_contiguous()):
        raise _hip.HipError("proposal_target_layer: skip_mask must be a contiguous uint8 device tensor of num_rois bytes")
    if anchors_3d is not None:
        _dev_f32(anchors_3d, "anchors_3d"); _dev_f32(true_gt_boxes, "true_gt_boxes")
        if anchors_3d.shape != (rois.shape[0], 7) or true_gt_boxes.shape != (gt_boxes.shape[0], 8) or gt_boxes.shape[1] != 5:
            raise _hip.HipError("proposal_target_layer: LiDAR form needs anchors_3d (num_rois,7), gt_boxes (G,5), "
                                "true_gt_boxes (G,8)")
        e = 7
        out = {"labels": torch.empty((r,), dtype=torch.float32, device=dev),
               "rois": torch.empty((r, 5), dtype=torch.float32, device=dev),
               "scores": torch.empty((r,), dtype=torch.float32, device=dev),
               "anchors_3d": torch.empty((r, 7), dtype=torch.float32, device=dev),
               "targets": torch.empty((r, e * num_classes), dtype=torch.float32, device=dev),
               "inside": torch.empty((r, e * num_classes), dtype=torch.float32, device=dev),
               "outside": torch.empty((r, e * num_classes), dtype=torch.float32, device=dev),
               "assign": torch.empty((r,), dtype=torch.int32, device=dev),
               "counts": torch.zeros((4,), dtype=torch.int32, device=dev)}
        _hip.check(lib.frcnn_proposal_target_layer_lidar(
            _ptr(rois), _ptr(roi_scores), _ptr(roi_count), rois.shape[0], _ptr(anchors_3d), _ptr(gt_boxes),
            _ptr(true_gt_boxes), gt_boxes.shape[0], _gt_count(gt_count), int(num_classes), r, float(fg_fraction), float(fg_thresh), float(bg_hi),
            float(bg_lo), _hip.float_array(means), _hip.float_array(stds), int(seed) & 0xFFFFFFFF, _ptr(seed_dev), _ptr(out["labels"]),
            _ptr(out["rois"]), _ptr(out["scores"]), _ptr(out["anchors_3d"]), _ptr(out["targets"]), _ptr(out["inside"]),
            _ptr(out["outside"]), _ptr(out["assign"]), _ptr(out["counts"]), _ptr(skip_mask), _stream()),
            "frcnn_proposal_target_layer_lidar")
        return out
    out = {"labels": torch.empty((r,), dtype=torch.float32, device=dev),
           "rois": torch.empty((r, 5), dtype=torch.float32, device=dev),
           "scores": torch.empty((r,), dtype=torch.float32, device=dev),
           "targets": torch.empty((r, 4 * num_classes), dtype=torch.float32, device=dev),
           "inside": torch.empty((r, 4 * num_classes), dtype=torch.float32, device=dev),
           "outside": torch.empty((r, 4 * num_classes), dtype=torch.float32, device=dev),
           "assign": torch.empty((r,), dtype=torch.int32, device=dev),
           "counts": torch.zeros((4,), dtype=torch.int32, device=dev)}
    _hip.check(lib.frcnn_proposal_target_layer(
        _ptr(rois), _ptr(roi_scores), _ptr(roi_count), rois.shape[0], _ptr(gt_boxes), gt_boxes.shape[0], _gt_count(gt_count),
        int(num_classes), r, float(fg_fraction), float(fg_thresh), float(bg_hi), float(bg_lo), _hip.float_array(means),
        _hip.float_array(stds), int(seed) & 0xFFFFFFFF, _ptr(seed_dev), _ptr(out["labels"]), _ptr(out["rois"]), _ptr(out["scores"]),
        _ptr(out["targets"]), _ptr(out["inside"]), _ptr(out["outside"]), _ptr(out["assign"]), _ptr(out["counts"]),
        _ptr(skip_mask), _stream()), "frcnn_proposal_target_layer")
    return out


def fpn_level_map(rois, k_min, k_max, canonical_scale=224.0, canonical_level=4.0, eps=1e-6):
    """rois (R,5) -> int32 (R,) pyramid level relative to k_min (LevelMapper, torchpoolers.py:39-51)."""
    lib = _hip.load()
    _dev_f32(rois, "rois")
    levels = torch.empty((rois.shape[0],), dtype=torch.int32, device=rois.device)
    _hip.check(lib.frcnn_fpn_level_map(_ptr(rois), rois.shape[0], int(k_min), int(k_max), float(canonical_scale),
                                       float(canonical_level), float(eps), _ptr(levels), _stream()), "frcnn_fpn_level_map")
    return levels


def spatial_mean(x):
    """(R,P,P,C) NHWC -> (R,C): x.mean(3).mean(2) of the NCHW-shaped view."""
    lib = _hip.load()
    _dev_f32(x, "x")
    r, p, _, c = x.shape
    out = torch.empty((r, c), dtype=torch.float32, device=x.device)
    _hip.check(lib.frcnn_spatial_mean_fwd(_ptr(x), _ptr(out), r, p, c, _stream()), "frcnn_spatial_mean_fwd")
    return out


def spatial_mean_bwd(dout, pooled):
    lib = _hip.load()
    _dev_f32(dout, "dout")
    r, c = dout.shape
    dx = torch.empty((r, pooled, pooled, c), dtype=torch.float32, device=dout.device)
    _hip.check(lib.frcnn_spatial_mean_bwd(_ptr(dout), _ptr(dx), r, pooled, c, _stream()), "frcnn_spatial_mean_bwd")
    return dx
