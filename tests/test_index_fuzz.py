"""Property-based fuzzing (hypothesis) of the INDEX-producing kernels against the CPU oracle, at the sizes the configs use.

north_star: "bit-exact for anchor/proposal indices and NMS keep masks".  The seeded tests of tests/test_gpu_parity.py draw
uniform boxes; here the inputs are built to sit ON the decision boundaries: zero-area and inverted boxes (NaN / negative
IoU terms), exact duplicates, runs of more than 64 equal scores (the tie rule of the rank sort and of the max_dets cut),
NaN / +-Inf scores and coordinates, sets in which every box is suppressed and sets in which none is, 1 .. 12 000 boxes
(cfg.TEST / cfg.TRAIN RPN_PRE_NMS_TOP_N, lib/model/config.py) - NMS (torchvision.ops.nms at lib/layer_utils/
proposal_layer.py:46 and lib/utils/filter_predictions.py:75-130), the score sort (proposal_layer.py:39-42), the RPN decode
(lib/model/bbox_transform.py:75-105,235-257) and the per-class filter with its max_dets cut (lib/model/test.py:213-221).
Every example runs under both readings of IoU == threshold.  derandomize=True: the examples are the same in every run.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import frcnn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SETTINGS = dict(derandomize=True, deadline=None, suppress_health_check=list(HealthCheck), print_blob=True)


def _ops():
    from faster_rcnn_pytorch_multimodal_amd import ops
    return ops


def _fuzz_boxes(rng, n, mix):
    """n boxes [x1,y1,x2,y2]: clustered / random bases, then a fraction of each adversarial kind (mix: dict of fractions)."""
    centres = rng.uniform(0, 900, (max(1, n // 40 + 1), 2)).astype(np.float32)
    pick = rng.integers(0, len(centres), n)
    xy = centres[pick] + rng.normal(0, mix["spread"], (n, 2)).astype(np.float32)
    wh = rng.uniform(1, mix["size"], (n, 2)).astype(np.float32)
    b = np.concatenate((xy, xy + wh), 1).astype(np.float32)
    if mix["snap"]:                                   # integer coordinates: exact IoU ties, IoU == threshold cases
        b = np.round(b / mix["snap"]) * mix["snap"]

    def some(frac):
        return rng.random(n) < frac
    m = some(mix["zero_area"])
    b[m, 2:] = b[m, :2]                               # zero area: IoU 0/0 = NaN with itself
    m = some(mix["inverted"])
    b[m] = b[m][:, [2, 3, 0, 1]]                      # x2 < x1: negative "area"
    m = some(mix["dup"])
    if m.any() and n > 1:
        b[m] = b[rng.integers(0, n, int(m.sum()))]    # exact duplicates
    m = some(mix["huge"])
    b[m] = np.array([-1e6, -1e6, 1e6, 1e6], np.float32)
    m = some(mix["nan"])
    b[m, rng.integers(0, 4, int(m.sum()))] = np.nan
    m = some(mix["inf"])
    b[m, 2] = np.inf
    return torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32))


MIX = st.fixed_dictionaries({
    "spread": st.sampled_from([0.0, 2.0, 8.0, 60.0]), "size": st.sampled_from([2.0, 40.0, 300.0]),
    "snap": st.sampled_from([0, 0, 1, 8]), "zero_area": st.sampled_from([0.0, 0.0, 0.05, 0.5]),
    "inverted": st.sampled_from([0.0, 0.0, 0.05]), "dup": st.sampled_from([0.0, 0.1, 0.9]),
    "huge": st.sampled_from([0.0, 0.0, 0.02]), "nan": st.sampled_from([0.0, 0.0, 0.01]), "inf": st.sampled_from([0.0, 0.0, 0.01])})


def _check_nms(boxes, thresh):
    ops = _ops()
    n = boxes.shape[0]
    scores = torch.arange(n, 0, -1, dtype=torch.float32)          # already in score order
    for at_equal in (True, False):
        old_dev, old_cpu = ops.set_nms_suppress_at_equal(at_equal), O.NMS_SUPPRESS_AT_EQUAL
        O.NMS_SUPPRESS_AT_EQUAL = at_equal
        try:
            ref = O.nms(boxes, scores, thresh)
            keep_idx, count, mask = ops.nms_sorted(boxes.to(DEV), thresh, want_mask=True)
            c = int(count.item())
            assert c == len(ref), (n, thresh, at_equal, c, len(ref))
            assert torch.equal(keep_idx[:c].cpu(), ref), (n, thresh, at_equal)
            ref_mask = torch.zeros(n, dtype=torch.uint8)
            ref_mask[ref] = 1
            assert torch.equal(mask.cpu(), ref_mask)
            k2, c2, _ = ops.nms_sorted(boxes.to(DEV), thresh, max_keep=min(300, n))     # post_nms_topN cut
            assert torch.equal(k2[:int(c2.item())].cpu(), ref[:300])
        finally:
            ops.set_nms_suppress_at_equal(old_dev)
            O.NMS_SUPPRESS_AT_EQUAL = old_cpu


@settings(max_examples=60, **SETTINGS)
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.sampled_from([1, 2, 3, 63, 64, 65, 127, 128, 129, 300, 1000, 2000]),
       thresh=st.sampled_from([0.0, 0.3, 0.5, 0.7, 1.0]), mix=MIX)
def test_fuzz_nms_keep_masks_bit_exact(hip, seed, n, thresh, mix):
    _check_nms(_fuzz_boxes(np.random.default_rng(seed), n, mix), thresh)


@pytest.mark.parametrize("n", [6000, 12000])
@pytest.mark.parametrize("kind", ["adversarial", "all_suppressed", "none_suppressed", "chain"])
def test_nms_keep_masks_at_config_sizes(hip, n, kind):
    """6000 (cfg.TEST.RPN_PRE_NMS_TOP_N) and 12 000 (cfg.TRAIN) boxes: an adversarial mix, one box repeated n times (the first
    survives), a grid of disjoint boxes (all survive: the longest possible keep list), and a chain in which every box
    overlaps only its neighbours just above the threshold (alternating survival: the serial dependence NMS is made of)."""
    rng = np.random.default_rng(n + len(kind))
    if kind == "adversarial":
        boxes = _fuzz_boxes(rng, n, {"spread": 8.0, "size": 40.0, "snap": 1, "zero_area": 0.05, "inverted": 0.05, "dup": 0.1,
                                     "huge": 0.001, "nan": 0.001, "inf": 0.001})
    elif kind == "all_suppressed":
        boxes = torch.tensor([[10.0, 20.0, 110.0, 220.0]]).repeat(n, 1)
    elif kind == "none_suppressed":
        i = torch.arange(n)
        x, y = (i % 120).float() * 8, (i // 120).float() * 6
        boxes = torch.stack((x, y, x + 5, y + 4), 1)
    else:
        x = torch.arange(n).float() * 2.0                              # width 10, shift 2: IoU with the neighbour 8/12 = 0.667
        boxes = torch.stack((x, torch.zeros(n), x + 10, torch.full((n,), 10.0)), 1)
    _check_nms(boxes.contiguous(), 0.7 if kind != "chain" else 0.6)


def _fuzz_scores(rng, n, kind):
    s = rng.standard_normal(n).astype(np.float32)
    if kind == "ties":
        s = np.round(s * 2) / 2                                         # a handful of distinct values: runs >> 64 long
    elif kind == "constant":
        s[:] = 0.25
    elif kind == "special":
        for v in (np.nan, np.inf, -np.inf, 0.0, -0.0, np.float32(1e-45), -np.float32(1e-45)):
            s[rng.random(n) < 0.03] = v
        s[rng.random(n) < 0.01] = np.float32(np.nan) * -1               # NaN with the sign bit set
    elif kind == "saturated":
        s = (rng.random(n) > 0.3).astype(np.float32)                    # > top_n scores exactly 1.0 (an untrained RPN)
    return torch.from_numpy(s)


@settings(max_examples=50, **SETTINGS)
@given(seed=st.integers(0, 2 ** 31 - 1), n=st.sampled_from([1, 37, 64, 65, 1100, 5000, 16385, 59850]),
       top=st.sampled_from([1, 16, 300, 6000, 12000]), kind=st.sampled_from(["random", "ties", "constant", "special", "saturated"]))
def test_fuzz_sort_topk_order_bit_exact(hip, seed, n, top, kind):
    """(score desc, index asc) - torch.sort(descending=True, stable=True): NaN ranks first (whatever its sign bit), then
    +Inf ... -Inf, -0.0 == +0.0; the returned scores are the input's own bits."""
    ops = _ops()
    scores = _fuzz_scores(np.random.default_rng(seed), n, kind)
    order, sout, count = ops.sort_topk_desc(scores.to(DEV), top)
    m = min(n, top)
    ref = O.stable_desc_order(scores)[:m]
    assert int(count.item()) == m
    assert torch.equal(order.cpu(), ref), (n, top, kind)
    assert torch.equal(sout.cpu().view(torch.int32), scores[ref].view(torch.int32))


def test_sort_topk_at_fpn_size_with_special_values(hip):
    """937 500 anchors (res101+FPN at 1000x600), top 12 000, with NaN / Inf / signed zeros / long tie runs mixed in."""
    ops = _ops()
    rng = np.random.default_rng(5)
    scores = _fuzz_scores(rng, 937500, "special")
    scores[rng.integers(0, 937500, 30000)] = 0.75
    order, sout, count = ops.sort_topk_desc(scores.to(DEV), 12000)
    ref = O.stable_desc_order(scores)[:12000]
    assert torch.equal(order.cpu(), ref)


@settings(max_examples=30, **SETTINGS)
@given(seed=st.integers(0, 2 ** 31 - 1), hw=st.sampled_from([(1, 1), (5, 7), (12, 17), (38, 63)]),
       scale=st.sampled_from([0.3, 3.0, 30.0, 200.0]), special=st.booleans())
def test_fuzz_rpn_decode_clip(hip, seed, hw, scale, special):
    """Decode + clip with deltas far outside the trained range: exp() overflow to Inf (Inf - Inf = NaN through the clip), NaN
    deltas.  Finite results within 1e-4 (relative for huge boxes), non-finite results in the same places with the same
    class (NaN / +Inf / -Inf; torch.clamp hands NaN through, so must the kernel); clipped coordinates identical."""
    ops = _ops()
    h, w = hw
    a = 25
    g = torch.Generator().manual_seed(seed)
    anchors = torch.from_numpy(O.generate_anchors_pre(h, w, 16, (2, 4, 8, 16, 32), (0.5, 0.75, 1, 1.25, 2))[0])
    deltas = torch.randn(h * w * a, 4, generator=g) * scale
    probs = torch.rand(h * w * a, generator=g)
    if special:
        m = torch.rand(h * w * a, generator=g) < 0.02
        deltas[m, 2] = float("nan")
        deltas[torch.rand(h * w * a, generator=g) < 0.02, 3] = float("inf")
        deltas[torch.rand(h * w * a, generator=g) < 0.02, 0] = -float("inf")
    info = np.array([0, 16.0 * w, 0, 16.0 * h, 0, 0, 1.0], np.float32)
    s, p = ops.rpn_decode_clip(anchors.to(DEV), info, a, probs=probs.to(DEV), deltas=deltas.contiguous().to(DEV))
    with np.errstate(all="ignore"):
        ref = O.clip_boxes(O.bbox_transform_inv(anchors, deltas), info)
    got = p.cpu()
    assert torch.equal(s.cpu(), probs)
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isnan(got), torch.isnan(ref)), "NaN pattern differs"
    assert torch.equal(torch.isfinite(got), fin)
    assert torch.equal(got[~fin & ~torch.isnan(ref)], ref[~fin & ~torch.isnan(ref)])        # +-Inf with the same sign
    # a coordinate is centre -+ half a size: with deltas this large the two terms reach 1e5 .. 1e30 before the clip and the
    # last-ulp freedom of expf is amplified by the cancellation - the tolerance follows the terms, not the result
    aw, ah = anchors[:, 2] - anchors[:, 0] + 1.0, anchors[:, 3] - anchors[:, 1] + 1.0
    with np.errstate(all="ignore"):
        term_x = (deltas[:, 0] * aw).abs() + torch.exp(deltas[:, 2]) * aw
        term_y = (deltas[:, 1] * ah).abs() + torch.exp(deltas[:, 3]) * ah
    tol = 1e-4 + 1e-6 * torch.stack((term_x, term_y, term_x, term_y), 1)
    tol = torch.nan_to_num(tol, nan=float("inf"), posinf=float("inf"))
    assert bool(((got - ref).abs()[fin] <= tol[fin]).all()), float(((got - ref).abs()[fin] - tol[fin]).max())


@settings(max_examples=40, **SETTINGS)
@given(seed=st.integers(0, 2 ** 31 - 1), r=st.sampled_from([1, 37, 64, 65, 300, 1024, 1500]),
       thresh=st.sampled_from([0.0, 0.05, 0.5, 0.999]), max_dets=st.sampled_from([1, 20, 100]),
       score_kind=st.sampled_from(["random", "ties", "all_equal", "none", "special"]), mix=MIX)
def test_fuzz_filter_per_class(hip, seed, r, thresh, max_dets, score_kind, mix):
    """Per-class threshold + clamp + NMS 0.6 + max_dets cut (ties at the cut stay, lib/model/test.py:213-221) on adversarial
    boxes and scores: runs of > 64 equal scores, every score equal, no score above the threshold, NaN scores."""
    from faster_rcnn_pytorch_multimodal_amd.utils.filter_predictions import filter_device
    ops = _ops()
    rng = np.random.default_rng(seed)
    k = 3
    info = np.array([0, 1000, 0, 600, 0, 0, 1.0], np.float32)
    if score_kind == "random":
        prob = F.softmax(torch.from_numpy(rng.standard_normal((r, k)).astype(np.float32)) * 2, 1)
    elif score_kind == "ties":
        prob = torch.from_numpy((np.round(rng.random((r, k)) * 4) / 4).astype(np.float32))
    elif score_kind == "all_equal":
        prob = torch.full((r, k), 0.625)
    elif score_kind == "none":
        prob = torch.from_numpy((rng.random((r, k)) * min(thresh, 0.04)).astype(np.float32))
    else:
        prob = torch.from_numpy(rng.random((r, k)).astype(np.float32))
        prob[torch.from_numpy(rng.random((r, k)) < 0.05)] = float("nan")
        prob[torch.from_numpy(rng.random((r, k)) < 0.02)] = float("inf")
    boxes = torch.cat([_fuzz_boxes(rng, r, dict(mix, nan=0.0, inf=0.0)) for _ in range(k)], 1)   # box NaNs: decode test above
    rois = torch.cat((torch.zeros(r, 1), boxes[:, :4]), 1)
    for at_equal in (True, False):
        old_dev, old_cpu = ops.set_nms_suppress_at_equal(at_equal), O.NMS_SUPPRESS_AT_EQUAL
        O.NMS_SUPPRESS_AT_EQUAL = at_equal
        try:
            _, ref_boxes, ref_clamped = O.filter_and_draw_prep(rois, prob, boxes, info, k, thresh)
            ref = [O.max_dets_cut(b, max_dets) for b in ref_boxes]
            boxes_dev = boxes.contiguous().to(DEV)
            dets, counts = filter_device(None, prob.contiguous().to(DEV), boxes_dev, info, thresh, max_dets, r)
            assert torch.equal(boxes_dev.cpu(), ref_clamped)
            dets, counts = dets.cpu().numpy(), counts.cpu().numpy()
            for j in range(1, k):
                assert counts[j] == len(ref[j]), (j, int(counts[j]), len(ref[j]), at_equal)
                want = ref[j]
                if len(want):
                    want = want[np.lexsort((np.arange(len(want)), -want[:, 4]))]       # ours is score order, ties by RoI index
                    np.testing.assert_array_equal(dets[j, :counts[j]], want)
        finally:
            ops.set_nms_suppress_at_equal(old_dev)
            O.NMS_SUPPRESS_AT_EQUAL = old_cpu


# ------------------------------------------------------------------------------------------------
# filter gradient: the addressing of the LDS-DMA kernel (range-checked buffer loads, incremental (img, ho, wo), tile tails)
# ------------------------------------------------------------------------------------------------
WGRAD_SHAPE = st.tuples(st.integers(1, 3), st.integers(1, 41), st.integers(1, 41), st.sampled_from([4, 8, 36, 64, 68, 132]),
                        st.sampled_from([4, 12, 64, 72, 136]), st.sampled_from([1, 3, 5, 7]), st.integers(1, 3),
                        st.integers(0, 3), st.sampled_from([1, 2, 7]))


@settings(max_examples=40, **SETTINGS)
@given(shape=WGRAD_SHAPE, seed=st.integers(0, 2 ** 16))
def test_fuzz_wgrad_dma_addressing_against_register_staged_kernel(shape, seed):
    """conv_wgrad_dma_f32 against conv_wgrad_f32 on random shapes (images smaller than a 32-pixel step, strides 1..3, every
    padding, channel / filter counts off the tile sizes): the 128-tile DMA kernel keeps the register-staged kernel's summation
    order, so ANY addressing slip - a wrong carry in (img, ho, wo), a row past M that is not zero, a tail chunk - shows as a
    bit difference; the 64-tile kernel (other order) is held to 2e-5 of the result's scale, and both to float64 autograd."""
    ops = _ops()
    n, h, w, c, k, r, stride, pad, splits = shape
    pad = min(pad, r - 1)
    if h + 2 * pad < r or w + 2 * pad < r:
        return
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h, w, c, generator=g)
    ho, wo = (h + 2 * pad - r) // stride + 1, (w + 2 * pad - r) // stride + 1
    dy = torch.randn(n, ho, wo, k, generator=g)
    wd = torch.zeros(k, c, r, r, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.permute(0, 3, 1, 2).double(), wd, None, stride=stride, padding=pad).backward(dy.permute(0, 3, 1, 2).double())
    ref = wd.grad.permute(0, 2, 3, 1).float()
    xd, dyd = x.to(DEV), dy.to(DEV)
    got = {}
    try:
        for kernel in (1, 2, 3, 4):
            ops.set_wgrad_plan(kernel, splits)
            got[kernel] = ops.conv2d_bwd_weight(xd, dyd, r, r, stride=stride, pad=pad)[0].cpu()
    finally:
        ops.set_wgrad_plan(0)
    assert torch.equal(got[2], got[4]), shape
    scale = max(float(ref.abs().max()), 1e-6)
    for kernel in (1, 3, 4):
        assert float((got[kernel] - ref).abs().max()) <= 2e-5 * scale + 1e-6, (shape, kernel)


def test_wgrad_split_slabs_are_fresh_in_every_launch():
    """The pixel-split slabs of conv_wgrad_dma_f32 cross XCDs at agent scope (sc1 stores / loads, no device-scope fence): a
    launch must never read a slab line that an earlier launch left in some L2.  Six launches of one shape with DIFFERENT operands,
    back to back on one stream - the caching allocator hands every launch the same workspace block, a replayed graph would too -
    each against float64 autograd; the counters end at zero."""
    ops = _ops()
    n, h, w, c, k, r = 1, 38, 63, 256, 256, 1
    g = torch.Generator().manual_seed(7)
    try:
        for kernel, splits in ((3, 16), (4, 16), (3, 64)):
            ops.set_wgrad_plan(kernel, splits)
            outs, refs, ptrs = [], [], set()
            for it in range(6):
                x = torch.randn(n, h, w, c, generator=g) * (1.0 + it)
                dy = torch.randn(n, h, w, k, generator=g)
                refs.append((dy.reshape(-1, k).double().t() @ x.reshape(-1, c).double()).float())       # (K, C): 1x1 filter gradient
                outs.append(ops.conv2d_bwd_weight(x.to(DEV), dy.to(DEV), r, r)[0])
            torch.cuda.synchronize()
            for it, (o, ref) in enumerate(zip(outs, refs)):
                scale = float(ref.abs().max())
                assert float((o.cpu().reshape(k, c) - ref).abs().max()) <= 2e-5 * scale, (kernel, splits, it)
    finally:
        ops.set_wgrad_plan(0)
    for ring in ops._COUNTER_RINGS.values():
        assert int(ring.ints.abs().max()) == 0
