"""The measured arrangement behind the drop-in API.

``model.test.test_net`` (lib/model/test.py:138-257), ``Network.test_frame`` (:75) and ``frame_detect`` (:68-93) replay each
frame as a captured hipGraph, ``cfg.TEST.FRAMES_IN_FLIGHT`` frames in flight on their own HIP streams
(model/frame_graph.FramePool) - the arrangement bench.py times.  Everything here compares that path with the eager
single-stream ``detect_frame_device`` on the same blob, bit for bit.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
COMPAT = os.path.join(ROOT, "faster_rcnn_pytorch_multimodal_amd", "compat")
DEV = "cuda:0"
UC_FLAGS = ("EN_BBOX_ALEATORIC", "EN_CLS_ALEATORIC", "EN_BBOX_EPISTEMIC", "EN_CLS_EPISTEMIC")


def _image_net(seed=7, uc=False):
    import torch
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    if uc:
        for k in UC_FLAGS:
            C.cfg.UC[k] = True
    net = imagenet(num_layers=101)
    net.create_architecture(2, tag="default", anchor_scales=C.cfg.ANCHOR_SCALES, anchor_ratios=C.cfg.ANCHOR_RATIOS)
    sd = seeded_state_dict(net, seed, bn_mode="tame")
    if uc:
        g = torch.Generator().manual_seed(seed)
        for name in sd:       # the new heads' reference init is N(0, 0.01): give them weights that spread the statistics
            if any(t in name for t in ("_fc1", "_fc2", "al_var_net", "cls_score_net", "bbox_pred_net")) and name.endswith("weight"):
                sd[name] = torch.randn(sd[name].shape, generator=g) * (0.02 if "fc" in name else 0.004)
    net.load_state_dict(sd, strict=True)
    net.eval()
    net._device = DEV
    net.to(DEV)
    return net, sd, C


class _Frames:
    """Frame source of model.test.test_net over in-memory blobs (the ``blobs_at`` protocol)."""
    num_classes = 2
    name = "synthetic_frames"

    def __init__(self, blobs, info):
        self.blobs, self.info = blobs, info

    def num_frames(self, mode):
        return len(self.blobs)

    def blobs_at(self, i, mode):
        return {"data": self.blobs[i], "info": self.info if not isinstance(self.info, list) else self.info[i]}


def _eager_records(net, blobs, infos, thresh, max_dets, max_out, seed=None):
    import torch
    from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device
    if seed is not None:
        net.set_uc_seed(seed)
    out = []
    for b, info in zip(blobs, infos):
        dets, counts = detect_frame_device(net, b, info, thresh, max_dets, max_out)
        torch.cuda.synchronize()
        out.append((dets.cpu().numpy().copy(), counts.cpu().numpy().copy()))
    return out


def _assert_boxes_equal(all_boxes, eager, k=2):
    total = 0
    for i, (dets, counts) in enumerate(eager):
        for j in range(1, k):
            n = int(counts[j])
            total += n
            got = all_boxes[j][i]
            assert got.reshape(-1, dets.shape[2]).shape[0] == n, (i, j, got.shape, n)
            np.testing.assert_array_equal(got.reshape(-1, dets.shape[2]), dets[j, :n])
    return total


@pytest.fixture
def reference_names_on_path():
    from faster_rcnn_pytorch_multimodal_amd import reference_names
    sys.path.insert(0, COMPAT)
    try:
        yield
    finally:
        sys.path.remove(COMPAT)
        reference_names.uninstall()


class _RefDb:
    """lib/datasets/db.py:39-40,46-51,139-148 - the members lib/model/test.py reads."""

    def __init__(self, directory, files):
        self._dir, self._val_index, self._test_index = directory, list(files), []
        self.evaluated = None

    name = "synthetic_db"
    num_classes = 2

    def path_at(self, i, mode="train"):
        return os.path.join(self._dir, mode, self._val_index[i]) if mode == "val" else None

    def evaluate_detections(self, all_boxes, output_dir, mode):
        self.evaluated = (len(all_boxes), output_dir, mode)


@pytest.mark.gpu
def test_test_net_cli_sequence_replays_graphs_on_18_full_size_frames(hip, tmp_path, reference_names_on_path):
    """tools/test_net.py:247-290 through the reference's import names on 18 DISTINCT 1000x600 frames (uint8 BGR files
    behind the reference's db protocol): every frame is a hipGraph replay (4 captures, 18 replays, no eager frame), 4
    frames in flight, and every all_boxes[cls][frame] is bit-equal to the eager single-stream detect_frame_device."""
    import torch
    from model.test import test_net                                           # noqa: E402  (the reference's lines)
    from model.config import cfg, cfg_from_list                              # noqa: E402
    from nets.imagenet import imagenet                                        # noqa: E402
    import faster_rcnn_pytorch_multimodal_amd.model.test as real_test
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    C.reset_cfg()
    cfg_from_list(["NET_TYPE", "image"])
    cfg.ROOT_DIR = str(tmp_path)
    os.makedirs(tmp_path / "val")
    rng = np.random.default_rng(23)
    files = []
    for i in range(18):
        name = "frame_%02d.npy" % i
        np.save(tmp_path / "val" / name, rng.integers(0, 256, (600, 1000, 3), dtype=np.uint8))
        files.append(name)
    db = _RefDb(str(tmp_path), files)
    net = imagenet(num_layers=101)                                            # test_net.py:256
    net.create_architecture(db.num_classes, tag='default', anchor_scales=cfg.ANCHOR_SCALES,
                            anchor_ratios=cfg.ANCHOR_RATIOS)                  # :272-276
    net.eval()                                                                # :278
    net.load_state_dict(seeded_state_dict(net, 7, bn_mode="tame"))            # :283
    net.to(net._device)                                                       # :287
    assert cfg.TEST.FRAME_GRAPHS and cfg.TEST.FRAMES_IN_FLIGHT == 4           # the defaults ARE the measured arrangement
    timers = {}
    all_boxes = test_net(net, db, None, max_dets=100, mode='val', thresh=0.05, draw_det=False, eval_det=True,
                         timers=timers)                                       # :290
    assert db.evaluated is not None and len(all_boxes[1]) == 18
    assert timers["pool"] == {"replays": 18, "eager": 0, "captures": 4, "refreshes": 0, "invalidations": 0}, timers
    blobs = [real_test._get_blobs([db.path_at(i, "val")]) for i in range(18)]
    assert tuple(blobs[0]["data"].shape) == (1, 600, 1000, 3)
    eager = _eager_records(net, [b["data"] for b in blobs], [b["info"] for b in blobs], 0.05, 100, 300)
    total = _assert_boxes_equal(all_boxes, eager)
    assert total > 0 and len({e[0].tobytes() for e in eager}) == 18           # distinct frames, distinct records
    # a second pass over the same db: replays only
    all_boxes2 = test_net(net, db, None, max_dets=100, mode='val', thresh=0.05, timers=timers)
    assert timers["pool"]["captures"] == 4 and timers["pool"]["replays"] == 36
    _assert_boxes_equal(all_boxes2, eager)
    C.reset_cfg()


@pytest.mark.gpu
def test_test_net_with_uncertainty_heads_replays_the_eager_draws(hip, tmp_path):
    """cfg.UC.* with E_NUM_SAMPLE = 10 Monte-Carlo passes (lib/model/test.py:73-77, lib/model/config.py:46): the captured
    frames read the seed of their counter-based draws from a device word, so frame i of the replayed loop draws the
    masks / logit noise the eager call number i draws - records incl. the uncertainty columns are bit-equal."""
    import torch
    from faster_rcnn_pytorch_multimodal_amd.model.test import test_net
    from faster_rcnn_pytorch_multimodal_amd.nets.uncertainty import num_uncertainty_pos
    net, sd, C = _image_net(seed=31, uc=True)
    try:
        assert C.cfg.UC.E_NUM_SAMPLE == 10
        rng = np.random.default_rng(5)
        info = np.array([0, 320, 0, 224, 0, 0, 1.0], np.float32)
        blobs = [torch.from_numpy((rng.standard_normal((1, 224, 320, 3)) * 50).astype(np.float32)).to(DEV) for _ in range(9)]
        net.set_uc_seed(77)
        timers = {}
        all_boxes = test_net(net, _Frames(blobs, info), str(tmp_path / "eval"), max_dets=100, thresh=0.05, timers=timers)
        assert timers["pool"]["replays"] == 9 and timers["pool"]["eager"] == 0
        eager = _eager_records(net, blobs, [info] * 9, 0.05, 100, 300, seed=77)
        width = 5 + num_uncertainty_pos(2, 4)
        assert eager[0][0].shape[2] == width == 21
        total = _assert_boxes_equal(all_boxes, eager)
        assert total > 0
        # the draws differ from frame to frame (the seed advanced): the same blob at positions 0 and 1 gives other columns
        net.set_uc_seed(77)
        twice = test_net(net, _Frames([blobs[0], blobs[0]], info), str(tmp_path / "eval2"), max_dets=100, thresh=0.05)
        a, b = twice[1][0].reshape(-1, width), twice[1][1].reshape(-1, width)
        assert not (a.shape == b.shape and np.array_equal(a, b))
    finally:
        C.reset_cfg()


@pytest.mark.gpu
def test_frame_detect_through_forward_graphs(hip):
    """lib/model/test.py:68-93 (frame_detect -> net.test_frame -> filter_and_draw_prep) with
    ``net.enable_frame_graphs()``: the forward pass is a replay, the tensors handed out are fresh copies, results equal
    the eager call's; a second frame does not disturb what the first call returned."""
    import torch
    from faster_rcnn_pytorch_multimodal_amd.model.test import frame_detect
    net, sd, C = _image_net()
    try:
        rng = np.random.default_rng(9)
        info = np.array([0, 256, 0, 192, 0, 0, 1.0], np.float32)
        frames = [(rng.standard_normal((1, 192, 256, 3)) * 50).astype(np.float32) for _ in range(3)]
        net.enable_frame_graphs()
        held = []
        for f in frames:
            out = net.test_frame(f, info)
            held.append([out[:4], frame_detect(net, {"data": f, "info": info}, 2, 0.05)])
        snapshot = [[t.clone() for t in h[0]] for h in held]
        # the eager path AFTER the pool's warm-up frames (same tuned convolution plans)
        net.enable_frame_graphs(False)
        for f, h, snap in zip(frames, held, snapshot):
            out = net.test_frame(f, info)
            for a, b, c in zip(out[:4], h[0], snap):
                assert torch.equal(a, b) and torch.equal(a, c)        # equal to eager, and intact after the later replays
            rois_np, boxes, _ = frame_detect(net, {"data": f, "info": info}, 2, 0.05)
            np.testing.assert_array_equal(rois_np, h[1][0])
            np.testing.assert_array_equal(np.asarray(boxes[1]), np.asarray(h[1][1][1]))
            assert len(boxes[1]) > 0
        st = net.frame_pool().stats
        assert st["captures"] == 1 and st["eager"] == 0 and st["replays"] == 6
    finally:
        C.reset_cfg()


@pytest.mark.gpu
def test_pool_shapes_weights_and_cached_filters(hip, tmp_path):
    """Two frame sizes through one pool, then new weights.
      * the first size is captured at once, a second size runs eagerly on first sight and is captured on the second;
      * the sizes get DIFFERENT plans for a 3x3 layer (size A forced to Winograd, size B to implicit GEMM): B's frames
        must not drop the Winograd filter A's graph reads by address;
      * ``load_state_dict`` with other values: the pool re-derives the cached filters in place and the SAME graphs give
        the new weights' records (equal to the eager path, different from the old records)."""
    import torch
    from faster_rcnn_pytorch_multimodal_amd import ops
    from faster_rcnn_pytorch_multimodal_amd.model.test import test_net
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    net, sd, C = _image_net()
    try:
        rng = np.random.default_rng(13)
        mk = lambda h, w: torch.from_numpy((rng.standard_normal((1, h, w, 3)) * 50).astype(np.float32)).to(DEV)
        info_a, info_b = np.array([0, 256, 0, 192, 0, 0, 1.0], np.float32), np.array([0, 320, 0, 160, 0, 0, 1.0], np.float32)
        pool = net.frame_pool(streams=2, capture_after=2, autotune=False)
        ops.set_conv_algo(2)                                        # size A: Winograd wherever it applies
        a_frames = [mk(192, 256) for _ in range(4)]
        boxes_a = test_net(net, _Frames(a_frames, info_a), str(tmp_path / "a"), thresh=0.05)
        eager_a = _eager_records(net, a_frames, [info_a] * 4, 0.05, 100, 300)
        _assert_boxes_equal(boxes_a, eager_a)
        conv = net.resnet.layer3[1].conv2
        u = conv.__dict__["_frcnn_winograd"][1][0]
        ptr = u.data_ptr()
        ops.set_conv_algo(1)                                        # size B: implicit GEMM only -> no plan reads U
        b_frames = [mk(160, 320) for _ in range(5)]
        boxes_b = test_net(net, _Frames(b_frames, info_b), str(tmp_path / "b"), thresh=0.05)
        _assert_boxes_equal(boxes_b, _eager_records(net, b_frames, [info_b] * 5, 0.05, 100, 300))
        assert pool.stats["eager"] == 1 and pool.stats["captures"] == 4, pool.stats
        assert conv.__dict__["_frcnn_winograd"][1][0].data_ptr() == ptr and "_frcnn_winograd_refresh" in conv.__dict__
        # new weights, in place
        sd2 = seeded_state_dict(net, 99, bn_mode="tame")
        net.load_state_dict(sd2, strict=True)
        ops.set_conv_algo(2)
        boxes_a2 = test_net(net, _Frames(a_frames, info_a), str(tmp_path / "a2"), thresh=0.05)
        assert pool.stats["refreshes"] == 1 and pool.stats["invalidations"] == 0 and pool.stats["captures"] == 4, pool.stats
        assert conv.__dict__["_frcnn_winograd"][1][0].data_ptr() == ptr
        fresh = ops.winograd_filter(conv.__dict__["_frcnn_prepared"][1][0])
        assert torch.equal(conv.__dict__["_frcnn_winograd"][1][0], fresh)
        eager_a2 = _eager_records(net, a_frames, [info_a] * 4, 0.05, 100, 300)
        _assert_boxes_equal(boxes_a2, eager_a2)
        assert any(not np.array_equal(x[0], y[0]) for x, y in zip(eager_a, eager_a2))
    finally:
        ops.set_conv_algo(0)
        C.reset_cfg()


def test_projected_head_respects_the_hook_protocol():
    """A subclass that overrides _head_to_tail / _crop_pool_layer (lib/nets/vgg16.py:49-59 does) must have its methods
    called in TEST mode too: the convolve-before-pooling shortcut of Network._predict only applies to Network's own
    methods.  Host logic only (no device work)."""
    import torch
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    from faster_rcnn_pytorch_multimodal_amd.nets.network import Network
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"

    class OwnTail(imagenet):
        def _head_to_tail(self, pool5):
            self.called = True
            return Network._head_to_tail(self, pool5)

    class OwnPool(imagenet):
        def _crop_pool_layer(self, bottom, rois):
            return Network._crop_pool_layer(self, bottom, rois)

    try:
        nets = []
        for cls in (imagenet, OwnTail, OwnPool):
            net = cls(num_layers=101)
            net.create_architecture(2, tag="default", anchor_scales=C.cfg.ANCHOR_SCALES, anchor_ratios=C.cfg.ANCHOR_RATIOS)
            net.eval()
            net._mode, net._pyramid = "TEST", None
            nets.append(net)
        with torch.no_grad():
            assert nets[0]._projected_head_ok() is True
            assert nets[1]._projected_head_ok() is False
            assert nets[2]._projected_head_ok() is False
            nets[0]._head_to_tail = lambda pool5: None            # an instance-level override counts too
            assert nets[0]._projected_head_ok() is False
    finally:
        C.reset_cfg()


@pytest.mark.gpu
def test_subclass_tail_is_called_in_test_mode(hip):
    """Device counterpart of the host test: the overriding tail runs and the detections equal the base class's (same
    function through the reference's order of operations, <= 2e-5 on probabilities / deltas)."""
    import torch
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    from faster_rcnn_pytorch_multimodal_amd.nets.network import Network
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    calls = []

    class OwnTail(imagenet):
        def _head_to_tail(self, pool5):
            calls.append(tuple(pool5.shape))
            return Network._head_to_tail(self, pool5)

    try:
        outs = []
        for cls in (imagenet, OwnTail):
            net = cls(num_layers=101)
            net.create_architecture(2, tag="default", anchor_scales=C.cfg.ANCHOR_SCALES, anchor_ratios=C.cfg.ANCHOR_RATIOS)
            net.load_state_dict(seeded_state_dict(net, 7, bn_mode="tame"))
            net.eval()
            net._device = DEV
            net.to(DEV)
            rng = np.random.default_rng(3)
            data = (rng.standard_normal((1, 160, 224, 3)) * 50).astype(np.float32)
            outs.append(net.test_frame(data, np.array([0, 224, 0, 160, 0, 0, 1.0], np.float32)))
        assert calls == [(300, 1024, 7, 7)] or (len(calls) == 1 and calls[0][1:] == (1024, 7, 7))
        assert torch.equal(outs[0][3], outs[1][3])                                   # same RoIs
        assert (outs[0][1] - outs[1][1]).abs().max().item() <= 2e-5                  # class probabilities
    finally:
        C.reset_cfg()


@pytest.mark.gpu
def test_roi_align_split_equals_two_calls_bit_for_bit(hip):
    """frcnn_roi_align_fwd_split (one plan, channel ranges of one map, per-range epilogue) against two complete
    frcnn_roi_align_fwd_affine calls on the two halves: identical bits (same kernel arithmetic per channel), for the bench's
    frame-wide RoIs, typical boxes, a live-row count below the row count, and a map whose first range is 256 channels."""
    import torch
    from faster_rcnn_pytorch_multimodal_amd import ops
    g = torch.Generator().manual_seed(4)
    for (h, w, c, split) in ((38, 63, 2560, 512), (25, 22, 768, 256)):
        feat = torch.randn((1, h, w, c), generator=g).to(DEV)
        sc, sh = (torch.rand(c, generator=g) + 0.5).to(DEV), torch.randn(c, generator=g).to(DEV)
        n = 300
        wh = torch.rand(n, 2, generator=g) * torch.tensor([w * 16.0, h * 16.0]) * 0.9 + 8
        xy = torch.rand(n, 2, generator=g) * (torch.tensor([w * 16.0, h * 16.0]) - wh).clamp(min=0)
        rois = torch.cat((torch.zeros(n, 1), xy, xy + wh), 1)
        rois[::7, 1] = 0.0
        rois[::7, 3] = w * 16.0 - 1            # frame-wide strips like an untrained RPN's
        rois = rois.to(DEV)
        for count in (None, torch.tensor([217], dtype=torch.int32, device=DEV)):
            o1, o2 = ops.roi_align_split(feat, rois, 7, 1.0 / 16.0, split, 0, roi_count=count, scale=sc, shift=sh, relu1=True)
            r1 = ops.roi_align_nhwc(feat[..., :split].contiguous(), rois, 7, 1.0 / 16.0, 0, roi_count=count, scale=sc[:split].contiguous(),
                                    shift=sh[:split].contiguous(), relu=True)
            r2 = ops.roi_align_nhwc(feat[..., split:].contiguous(), rois, 7, 1.0 / 16.0, 0, roi_count=count, scale=sc[split:].contiguous(),
                                    shift=sh[split:].contiguous(), relu=False)
            torch.cuda.synchronize()
            assert torch.equal(o1, r1) and torch.equal(o2, r2), (h, w, c, split, count)
            assert float(o1.min()) >= 0.0 and float(o2.min()) < 0.0


@pytest.mark.gpu
def test_lidar_test_net_replays_graphs(hip, tmp_path):
    """The LiDAR-BEV detector (BASELINE configs[2] geometry at a small grid) through ``test_net``: captured frames, 4 in flight,
    records equal to the eager path row for row after the voxel-grid -> metres conversion of lib/model/test.py:223-224, result
    text file in the LiDAR format of lib/datasets/db.py:336-367."""
    import torch
    import test_gpu_parity as T
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.model.test import bbox_voxel_grid_to_pc, lidar_extents, test_net
    net, _ = T._build_lidar_pair(seed=9)
    try:
        info = np.array([0, 176, 0, 208, 0, 12, 0.5], np.float32)
        blobs = [torch.from_numpy(T._bev_blob(208, 176, 40 + i)).to(DEV) for i in range(6)]
        timers = {}
        all_boxes = test_net(net, _Frames(blobs, info), str(tmp_path / "eval"), max_dets=100, thresh=0.05, timers=timers)
        assert timers["pool"]["replays"] == 6 and timers["pool"]["eager"] == 0 and timers["pool"]["captures"] == 4
        eager = _eager_records(net, blobs, [info] * 6, 0.05, 100, 300)
        total = 0
        for i, (dets, counts) in enumerate(eager):
            n = int(counts[1])
            total += n
            want = bbox_voxel_grid_to_pc(dets[1, :n].copy(), lidar_extents(), info) if n else dets[1, :0]
            np.testing.assert_array_equal(all_boxes[1][i].reshape(-1, 8), want)
        assert total > 0
        lines = open(tmp_path / "eval" / "det_test_cls1.txt").read().splitlines()
        assert len(lines) == total and len(lines[0].split(" ")) == 10
    finally:
        C.reset_cfg()


@pytest.mark.gpu
def test_fpn_detector_test_net_replays_graphs(hip, tmp_path):
    """The res101+FPN detector (cfg.USE_FPN, 'multiscale' pooling over p2..p5, custom tail: tools/trainval_net.py:326-330)
    through ``test_net``: the pyramid, the FPN level map and the multi-level RoIAlign are captured like the plain detector's
    frame; records equal the eager path."""
    import torch
    import test_gpu_parity as T
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.model.test import test_net
    net, _ = T._build_fpn_pair(seed=23)
    try:
        net.eval()
        rng = np.random.default_rng(17)
        info = np.array([0, 320, 0, 256, 0, 0, 1.0], np.float32)
        blobs = [torch.from_numpy((rng.standard_normal((1, 256, 320, 3)) * 50).astype(np.float32)).to(DEV) for _ in range(6)]
        timers = {}
        all_boxes = test_net(net, _Frames(blobs, info), str(tmp_path / "eval"), max_dets=100, thresh=0.05, timers=timers)
        assert timers["pool"]["replays"] == 6 and timers["pool"]["eager"] == 0, timers
        eager = _eager_records(net, blobs, [info] * 6, 0.05, 100, 300)
        assert _assert_boxes_equal(all_boxes, eager) > 0
    finally:
        C.reset_cfg()
