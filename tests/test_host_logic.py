"""CPU: host logic of the product package — config, anchor base table, C-ABI surface, loud failure."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from faster_rcnn_pytorch_multimodal_amd import _hip, ops
from faster_rcnn_pytorch_multimodal_amd.layer_utils.generate_anchors import generate_anchors
from faster_rcnn_pytorch_multimodal_amd.model import config as C

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_base_anchor_table_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "anchors.npz"))
    np.testing.assert_array_equal(generate_anchors(), g["default9"])
    np.testing.assert_array_equal(generate_anchors(ratios=np.array([0.5, 0.75, 1, 1.25, 2]),
                                                   scales=np.array([2, 4, 8, 16, 32])), g["waymo25"])
    # the dense grid = base + shift in float64, rounded once: check the first cell rows against the golden grid
    full = g["pre_38x63_s16"]
    np.testing.assert_array_equal(g["waymo25"].astype(np.float32), full[:25])
    # non-dyadic frame scale (scales * 0.3): float32(base64 + shift) must equal the reference grid
    base = generate_anchors(ratios=np.array([0.5, 0.75, 1, 1.25, 2]), scales=np.array([2, 4, 8, 16, 32]) * 0.3)
    grid = g["pre_5x7_s16_fs0.3"].reshape(5, 7, 25, 4)
    for (y, x) in ((0, 0), (2, 3), (4, 6)):
        shift = np.array([x * 16, y * 16, x * 16, y * 16], dtype=np.float64)
        np.testing.assert_array_equal((base + shift).astype(np.float32), grid[y, x])


def test_cfg_defaults_and_overrides():
    C.reset_cfg()
    cfg = C.cfg
    assert cfg.TEST.RPN_PRE_NMS_TOP_N == 6000 and cfg.TEST.RPN_POST_NMS_TOP_N == 300
    assert cfg.TEST.RPN_NMS_THRESH == 0.7 and cfg.TEST.NMS_THRESH == 0.6
    assert cfg.ANCHOR_SCALES == [2, 4, 8, 16, 32] and cfg.POOLING_SIZE == 7 and cfg.RPN_CHANNELS == 512
    assert cfg.TRAIN.IMAGE.BBOX_NORMALIZE_STDS == (0.1, 0.1, 0.2, 0.2)
    C.cfg_from_list(["TEST.RPN_POST_NMS_TOP_N", "100", "POOLING_MODE", "multiscale"])
    assert cfg.TEST.RPN_POST_NMS_TOP_N == 100 and cfg.POOLING_MODE == "multiscale"
    with pytest.raises(AssertionError):
        C.cfg_from_list(["TEST.NO_SUCH_KEY", "1"])
    with pytest.raises(AssertionError):
        C.cfg_from_list(["TEST.RPN_POST_NMS_TOP_N", "'three'"])
    C.reset_cfg()
    assert cfg.TEST.RPN_POST_NMS_TOP_N == 300


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    declared = set(re.findall(r"\b(frcnn_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    assert os.path.exists(_hip.library_path()), "libfrcnn_hip.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(_hip.library_path())
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, "declared in include/frcnn_hip.h but not exported: %s" % missing
    assert declared == set(_hip.PROTOTYPES), "ctypes prototypes out of sync with the header"
    lib.frcnn_version.restype = ctypes.c_int
    assert lib.frcnn_version() >= 100


def test_argument_errors_are_reported_without_a_gpu():
    lib = _hip.load()
    # null pointers / bad shapes are rejected on the host before any launch
    rc = lib.frcnn_conv2d_fwd(None, None, None, None, None, None, 1, 8, 8, 4, 8, 3, 3, 1, 1, 0, 0, None, 0, None)
    assert rc == -1 and b"null" in lib.frcnn_last_error()
    assert lib.frcnn_conv2d_fwd_ws_bytes(1, 8, 8, 3, 8, 3, 3, 1, 1, 0) == 0      # c % 4 != 0 -> invalid
    assert lib.frcnn_nms_ws_bytes(6000) == 6000 * 94 * 8 + 6000 * 8      # bit-matrix + one predecessor word per box
    assert lib.frcnn_conv2d_fwd_ws_bytes(1, 38, 63, 256, 256, 3, 3, 1, 1, 3) == 3 * 38 * 63 * 256 * 4


def test_product_has_no_cpu_path():
    with pytest.raises(_hip.HipError, match="no CPU path"):
        ops.conv2d_nhwc(torch.zeros(1, 8, 8, 4), torch.zeros(8, 3, 3, 4))
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    net = imagenet(num_layers=101)
    net.create_architecture(2, tag="default", anchor_scales=C.cfg.ANCHOR_SCALES, anchor_ratios=C.cfg.ANCHOR_RATIOS)
    net._device = "cpu"
    with pytest.raises(RuntimeError, match="MI355X only"):
        net.test_frame(np.zeros((1, 32, 32, 3), np.float32), np.array([0, 32, 0, 32, 0, 0, 1], np.float32))


def test_state_dict_keys_follow_the_reference_names():
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    from oracle.frcnn_oracle import ImageNetOracle
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    net = imagenet(num_layers=101)
    net.create_architecture(2, tag="default", anchor_scales=C.cfg.ANCHOR_SCALES, anchor_ratios=C.cfg.ANCHOR_RATIOS)
    keys = set(net.state_dict().keys())
    for k in ("resnet.conv1.weight", "resnet.bn1.running_var", "resnet.layer1.0.downsample.0.weight",
              "resnet.layer3.22.bn3.num_batches_tracked", "resnet.layer4.2.conv3.weight", "rpn_net.weight",
              "rpn_cls_score_net.bias", "rpn_bbox_pred_net.weight", "cls_score_net.weight", "bbox_pred_net.bias"):
        assert k in keys, k
    assert keys == set(ImageNetOracle().state_dict().keys())
    assert net.rpn_cls_score_net.weight.shape == (50, 512, 1, 1) and net.bbox_pred_net.weight.shape == (8, 2048)
    # frozen-parameter policy of lib/nets/imagenet.py:96-116 with FIXED_BLOCKS = 1
    assert not net.resnet.conv1.weight.requires_grad and not net.resnet.layer1[0].conv1.weight.requires_grad
    assert net.resnet.layer2[0].conv1.weight.requires_grad and not net.resnet.layer2[0].bn1.weight.requires_grad
    # caffe stride placement (lib/nets/resnet.py:232-238)
    assert net.resnet.layer2[0].conv1.stride == (2, 2) and net.resnet.layer2[0].conv2.stride == (1, 1)
    assert net.resnet.layer4[0].conv2.stride == (1, 1) and net.resnet.layer4[0].downsample[0].stride == (1, 1)


def test_lidarnet_module_protocol():
    """lib/nets/lidarnet.py:29-151: 15-plane stem, 2 anchors per cell, 7-DoF box head, trainable BN above the
    frozen blocks, state-dict keys equal to the oracle's (= the reference's names)."""
    from faster_rcnn_pytorch_multimodal_amd.nets.lidarnet import lidarnet
    from oracle.frcnn_oracle import LidarNetOracle
    C.reset_cfg()
    C.cfg.NET_TYPE = "lidar"
    net = lidarnet(num_layers=101)
    net.create_architecture(2, tag="default", anchor_scales=C.cfg.LIDAR.ANCHOR_SCALES[0],
                            anchor_ratios=C.cfg.LIDAR.ANCHOR_ANGLES)
    assert set(net.state_dict().keys()) == set(LidarNetOracle().state_dict().keys())
    assert net.resnet.conv1.weight.shape == (64, 15, 7, 7) and net._num_anchors == 2
    assert net.rpn_cls_score_net.weight.shape == (4, 512, 1, 1) and net.bbox_pred_net.weight.shape == (14, 2048)
    assert net._batchnorm_en is False and net.resnet.layer4[0].batchnorm_en is False
    # FIXED_BLOCKS = 1: stem + layer1 frozen, BN of layer2.. trainable (set_bn_var)
    assert not net.resnet.conv1.weight.requires_grad and not net.resnet.layer1[0].bn1.weight.requires_grad
    assert net.resnet.layer2[0].conv1.weight.requires_grad and net.resnet.layer2[0].bn1.weight.requires_grad
    C.reset_cfg()


def test_anchor_type_table_matches_oracle_bev_boxes():
    from faster_rcnn_pytorch_multimodal_amd.layer_utils.generate_3d_anchors import anchor_type_table
    from oracle import frcnn_oracle as O
    C.reset_cfg()
    for fs in (0.5, 1.0, 0.3):
        t = anchor_type_table((1.0,), C.cfg.LIDAR.ANCHOR_ANGLES, fs)
        _, a3 = O.generate_anchors_3d(2, 3, 16, frame_scale=fs)
        a2 = O.bbaa_graphics_gems(a3)
        for k in range(2):          # the two anchor types at cell (0, 0)
            np.testing.assert_array_equal(t[k, 4:], a3[k, 2:])
            np.testing.assert_array_equal(t[k, :4], a2[k])


def test_bbaa_graphics_gems_matches_reference_vectors():
    """utils/bbox.bbaa_graphics_gems (host helper of the LiDAR training path) against the reference's own outputs
    (tests/golden/lidar_train.npz: BEV rectangles of the 3-D anchor grids)."""
    import os
    import numpy as np
    from faster_rcnn_pytorch_multimodal_amd.utils.bbox import bbaa_graphics_gems
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lidar_train.npz"))
    for tag in ("25x22_fs0.5", "50x44_fs1", "7x5_fs0.3"):
        np.testing.assert_array_equal(bbaa_graphics_gems(z["a3d_" + tag]), z["a2d_" + tag])
    assert bbaa_graphics_gems(np.zeros((0, 7), np.float32)).shape == (0, 4)
    clipped = bbaa_graphics_gems(np.array([[5.0, 5.0, 0, 20.0, 8.0, 2.0, 0.3]], np.float32), 16, 12, clip=True)
    assert clipped.min() >= 0 and clipped[0, 2] <= 15 and clipped[0, 3] <= 11


def test_oracle_voxel_generator_known_answers():
    """oracle.points_to_voxel / get_lidar_blob on a hand-built cloud (spconv is absent: the restatement is checked
    against answers worked out by hand from lib/roi_data_layer/minibatch.py:434-512)."""
    import numpy as np
    from oracle import frcnn_oracle as O
    pts = np.array([
        [0.05, -39.95, -2.9, 1.0, 0.5],    # cell (x0, y0, z0)
        [0.06, -39.96, -2.7, 3.0, 0.5],    # same cell, higher
        [0.05, -39.95, 0.30, 2.0, 1.5],    # same column, slice 6 -> created later: owns the meta channels
        [69.99, 39.99, 2.99, 0.0, 0.0],    # last cell of the grid
        [70.00, 0.0, 0.0, 9.0, 9.0],       # x == X_RANGE[1]: filtered
        [5.0, 5.0, -3.01, 9.0, 9.0],       # below Z_RANGE[0]: filtered
    ], np.float32)
    info, blob = O.get_lidar_blob(pts, 0.5, elongation=True)
    assert info.tolist() == [0, 350, 0, 400, 0, 12, 0.5] and blob.shape == (1, 400, 350, 15)
    px = blob[0, 0, 0]
    assert abs(px[0] - 0.3) < 1e-6                      # max(z+3) - 0*0.5 = 0.3 (second point)
    assert abs(px[6] - 0.3) < 1e-6                      # 3.3 - 6*0.5
    assert px[12] == 1 / 32                             # density of the voxel created last in the column
    assert abs(px[13] - np.tanh(2.0)) < 1e-6 and abs(px[14] - np.tanh(1.5)) < 1e-6
    last = blob[0, 399, 349]
    assert abs(last[11] - (5.99 - 5.5)) < 1e-5 and last[12] == 1 / 32 and last[13] == 0.0
    assert int((blob != 0).sum()) == 2 + 3 + 2          # two slices + 3 meta | one slice + density (intensity 0)
    # voxel cap: with max_voxels = 1 only the first cell survives, and only its first max_points = 1 point
    _, capped = O.get_lidar_blob(pts, 0.5, elongation=True, max_points=1, max_voxels=1)
    assert int((capped != 0).sum()) == 4 and abs(capped[0, 0, 0, 0] - 0.1) < 1e-6 and capped[0, 0, 0, 12] == 1.0


def test_voc_eval_matches_reference_vectors(tmp_path):
    """datasets/voc_eval.voc_eval_arrays + voc_ap against the reference's voc_eval on a synthetic 6-frame set
    (tests/golden/eval.npz, generated by tests/golden/make_golden_eval.py), then the detections file format."""
    import os
    import numpy as np
    from faster_rcnn_pytorch_multimodal_amd.datasets import voc_eval as V
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eval.npz"))
    frames = [str(f) for f in z["frames"]]
    recs = {f: {"bbox": z["gt_box_" + f], "difficult": z["gt_diff_" + f]} for f in frames}
    ids = [str(f) for f in z["det_frame"]]
    for tag, (thr, m07) in {"iou50": (0.5, False), "iou70": (0.7, False), "iou50_07": (0.5, True)}.items():
        rec, prec, ap = V.voc_eval_arrays(ids, z["det_score"], z["det_box"], recs, ovthresh=thr, use_07_metric=m07)
        np.testing.assert_array_equal(rec, z["rec_" + tag])
        np.testing.assert_array_equal(prec, z["prec_" + tag])
        assert ap == z["ap_" + tag][0]
    assert 0.3 < z["ap_iou70"][0] < z["ap_iou50"][0] < 1.0                      # the case is not degenerate
    # results files (lib/datasets/db.py:305-367): the exact text of one row, and a write -> read round trip
    dets = [np.array([[10.04, 20.06, 110.96, 220.0, 0.98765, 0.5]]), np.zeros((0, 6)),
            np.array([[1, 2, 3, 4, 0.5, 0.25], [5, 6, 7, 8, 0.125, 0.75]], dtype=np.float64)]
    path = str(tmp_path / "det_car.txt")
    V.write_image_results_file(dets, ["a", "b", "c"], path)
    lines = open(path).read().splitlines()
    assert lines[0] == "0 a 0.988 10.0 20.1 111.0 220.0 0.5000000000" and len(lines) == 3 and lines[1].startswith("2 c 0.500 ")
    idx, tok, score, box, extra = V.read_results_file(path)
    assert idx.tolist() == [0, 2, 2] and tok == ["a", "c", "c"] and box.shape == (3, 4) and extra.shape == (3, 1)
    lid = [np.array([[1.5, -2.25, 0.5, 4.7, 2.1, 1.8, 0.31, 0.9]])]
    lpath = str(tmp_path / "det_lidar.txt")
    V.write_lidar_results_file(lid, ["t0"], lpath)
    assert open(lpath).read() == "0 t0 0.900 1.500 -2.250 0.500 4.700 2.100 1.800 0.31000\n"
    _, _, s, b, _ = V.read_results_file(lpath, num_box_values=7)
    assert s.tolist() == [0.9] and b.shape == (1, 7)


def test_c_abi_from_plain_c(tmp_path):
    """include/frcnn_hip.h is valid C99 and libfrcnn_hip.so is usable without Python, C++ or torch: a C program built
    with gcc dlopens the library, resolves every declared symbol and calls the GPU-free entry points."""
    import os
    import shutil
    import subprocess
    from faster_rcnn_pytorch_multimodal_amd import _hip
    from faster_rcnn_pytorch_multimodal_amd.build import LIB_PATH
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_check")
    subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_abi", "abi_check.c"), "-o", exe, "-ldl"], check=True)
    out = subprocess.run([exe, LIB_PATH] + sorted(_hip.PROTOTYPES), check=True, capture_output=True, text=True)
    assert "abi ok" in out.stdout and ("%d symbols" % len(_hip.PROTOTYPES)) in out.stdout


def test_storage_stable_weight_caches_keep_addresses_and_refresh():
    """nets.hip_modules.stable_store / refresh_derived_weights (what a captured training graph relies on): a derived tensor
    is re-derived INTO the storage of the previous entry, dependents are refreshed after what they read, and a missing or
    reshaped entry falls back to new storage."""
    import torch
    from faster_rcnn_pytorch_multimodal_amd.nets import hip_modules as HM

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.arange(6, dtype=torch.float32).view(2, 3))

    net = torch.nn.Sequential(Holder(), Holder())
    calls = []

    def doubled(m):                               # "KRSC filter": derived from the parameter
        key = (m.weight._version, m.weight.data_ptr())
        c = m.__dict__.get('_frcnn_prepared')
        if c is not None and c[0] == key:
            return c[1][0]
        calls.append('prepared')
        return HM.stable_store(m, '_frcnn_prepared', key, (m.weight.detach() * 2,), refresh=lambda: doubled(m))[0]

    def transposed(m):                            # "transposed filter": derived from the derived tensor
        src = doubled(m)
        key = (src.data_ptr(), src._version)
        c = m.__dict__.get('_frcnn_wt')
        if c is not None and c[0] == key:
            return c[1][0]
        calls.append('wt')
        return HM.stable_store(m, '_frcnn_wt', key, (src.t().contiguous(),), refresh=lambda: transposed(m))[0]

    first = [(doubled(m), transposed(m)) for m in net]
    ptrs = [(a.data_ptr(), b.data_ptr()) for a, b in first]
    with torch.no_grad():
        for m in net:
            m.weight.add_(1.0)                    # the optimizer step: in place, version bump
    calls.clear()
    assert HM.refresh_derived_weights(net) == 4
    assert calls == ['prepared', 'prepared', 'wt', 'wt']            # sources before their dependents
    for m, (a, b), (pa, pb) in zip(net, first, ptrs):
        assert (a.data_ptr(), b.data_ptr()) == (pa, pb)             # same storage ...
        assert torch.equal(a, m.weight.detach() * 2) and torch.equal(b, (m.weight.detach() * 2).t())   # ... new values
    assert HM.refresh_derived_weights(net) == 4 and calls == ['prepared', 'prepared', 'wt', 'wt']      # nothing changed: hits
    # a reshaped entry cannot reuse the storage
    h = Holder()
    old = HM.stable_store(h, 'x', 1, (torch.zeros(3),))[0]
    new = HM.stable_store(h, 'x', 2, (torch.ones(4),))[0]
    assert new.data_ptr() != old.data_ptr() and new.shape == (4,)


def test_train_graph_refuses_what_it_cannot_capture():
    """graphable(): what still runs eagerly (with a warning from Network.train_step) - more don't-care boxes than the captured
    step's buffer holds, frames without gt, BatchNorm with a host-computed momentum.  The LiDAR detector, FIXED_BLOCKS = -1 and
    the uncertainty heads ARE capturable since round 4 (device-side gt count / seeds, in-kernel running statistics), frames
    with don't-care boxes since round 5 (padded buffer)."""
    import torch.nn as nn
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.model import train_graph
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    net = nn.Sequential(nn.Conv2d(4, 4, 1), nn.BatchNorm2d(4))
    blobs = {"gt_boxes": np.zeros((2, 5), np.float32), "gt_boxes_dc": np.zeros((0, 5), np.float32)}
    assert train_graph.graphable(net, blobs) is None
    assert "ground-truth" in train_graph.graphable(net, {"gt_boxes": np.zeros((0, 5), np.float32)})
    for key, val in (("RESNET.FIXED_BLOCKS", -1), ("NET_TYPE", "lidar")):
        C.reset_cfg()
        C.cfg_from_list([key, repr(val)])
        assert train_graph.graphable(net, blobs) is None
    C.reset_cfg()
    C.cfg.UC.EN_BBOX_ALEATORIC = True
    assert train_graph.graphable(net, blobs) is None
    C.reset_cfg()
    C.cfg.TRAIN.IGNORE_DC = True
    assert train_graph.graphable(net, dict(blobs, gt_boxes_dc=np.zeros((3, 5), np.float32))) is None
    assert "don't-care" in train_graph.graphable(net, dict(blobs, gt_boxes_dc=np.zeros((train_graph.DC_CAPACITY + 1, 5), np.float32)))
    C.reset_cfg()
    net[1].momentum = None
    assert "momentum" in train_graph.graphable(net, blobs)
    assert [train_graph.gt_capacity(n) for n in (1, 32, 33, 64, 65)] == [32, 32, 64, 64, 128]
    C.reset_cfg()


def test_committed_profiles_describe_the_committed_plan_table():
    """bench.py quotes counter values from profiles/ only for the plan table they were collected with: the committed PMC
    summaries must carry the hash of the committed table (bench.plans_sha) and the head they were collected at, and
    `from_profiles` must null every plan-dependent value for a run on another table."""
    import json
    import bench
    rows = json.load(open(os.path.join(ROOT, "profiles", bench.PLANS_FILE)))
    sha = bench.plans_sha(rows)
    assert sha == bench.plans_sha(list(reversed(rows)))                       # independent of row order
    for name in (bench.PMC_FILE, bench.PMC_TIMED_FILE):
        prof = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert prof["plans_sha"] == sha, name
        assert prof["collected_at_head"] and prof["collected_at_head"] != "unknown", name
    mine = bench.from_profiles(sha)
    assert mine["plans_match"] and mine["conv_traffic_bytes_per_call"] > 3e7 and mine["mfma_busy_ms_per_frame_timed_mode"] > 1.0
    other = bench.from_profiles("0" * 16)
    assert not other["plans_match"] and other["conv_traffic_bytes_per_call"] is None
    assert other["mfma_busy_ms_per_frame_timed_mode"] is None and other["conv_mfma_util_percent_isolated"] is None
    assert other["roi_align_traffic_bytes_per_call"] == mine["roi_align_traffic_bytes_per_call"]   # not plan dependent
    # every plan of the table is one the library accepts: tile index < 13, algorithm code <= 2
    for r in rows:
        assert len(r) == 13 and (r[10] & 15) < 14 and (r[10] >> 4) <= 2 and r[11] >= 1


def test_stream_selection_from_a_measured_collision_matrix():
    """model/streams.choose_overlapping (the host half of concurrent_streams): given which candidate pairs were measured to run
    one after the other - here the pattern of a fresh process on four hardware queues, tools/stream_concurrency_probe.py:
    candidates {2,3,7,11}, {0,5,9}, {1,4,8}, {6,10} share a queue - it returns a mutually overlapping set of the size asked
    for when there is one, and otherwise the largest one padded evenly over its queues."""
    import itertools
    from faster_rcnn_pytorch_multimodal_amd.model.streams import choose_overlapping
    queue = {2: 0, 3: 0, 7: 0, 11: 0, 0: 1, 5: 1, 9: 1, 1: 2, 4: 2, 8: 2, 6: 3, 10: 3}
    ser = {(i, j): queue[i] == queue[j] for i, j in itertools.combinations(range(12), 2)}
    for n in (2, 3, 4):
        chosen, distinct = choose_overlapping(ser, 12, n)
        assert distinct == n == len(chosen) == len({queue[i] for i in chosen}), (n, chosen)
    chosen, distinct = choose_overlapping(ser, 12, 4)
    assert chosen == [0, 1, 2, 6]                                    # smallest indices among equal choices
    chosen, distinct = choose_overlapping(ser, 12, 7)                # more streams than queues: 4 distinct, 3 more, one per queue
    assert distinct == 4 and len(set(chosen)) == 7
    load = [sum(queue[i] == q for i in chosen) for q in range(4)]
    assert sorted(load) == [1, 2, 2, 2], load
    everything_collides = {(i, j): True for i, j in itertools.combinations(range(5), 2)}       # GPU_MAX_HW_QUEUES=1
    chosen, distinct = choose_overlapping(everything_collides, 5, 3)
    assert distinct == 1 and len(set(chosen)) == 3


def test_tile_counter_arenas():
    """ops.CounterArena / wgrad_counter_arena / _tile_counters (host logic of the `counters` argument of frcnn_conv2d_bwd_weight
    and frcnn_bn_train_*): a runner's arena hands every launch of a pass its own range, the same ranges again after rewind(),
    and refuses to overrun; without an arena eager launches walk a per-device ring that wraps instead of growing."""
    from faster_rcnn_pytorch_multimodal_amd import _hip, ops
    cpu = torch.device("cpu")
    arena = ops.CounterArena(cpu, 16)
    assert int(arena.ints.abs().max()) == 0 and arena.ints.dtype == torch.int32
    with ops.wgrad_counter_arena(arena) as inside:
        assert inside is arena and ops.WGRAD_ARENA is arena
        a = ops._tile_counters(cpu, 6)
        b = ops._tile_counters(cpu, 6)
        assert a.numel() == b.numel() == 6 and b.data_ptr() == a.data_ptr() + 6 * 4
        with pytest.raises(_hip.HipError, match="exhausted"):
            ops._tile_counters(cpu, 6)
        arena.rewind()                                             # the next pass over the same launch sequence
        assert ops._tile_counters(cpu, 6).data_ptr() == a.data_ptr()
        with ops.wgrad_counter_arena(ops.CounterArena(cpu, 4)) as inner:      # nests, and restores
            assert ops.WGRAD_ARENA is inner
        assert ops.WGRAD_ARENA is arena
    assert ops.WGRAD_ARENA is None
    saved, ops.WGRAD_COUNTER_RING = ops.WGRAD_COUNTER_RING, 8
    ops._COUNTER_RINGS.pop("cpu", None)
    try:
        first = ops._tile_counters(cpu, 5)
        second = ops._tile_counters(cpu, 5)                        # 5 + 5 > 8: the ring wraps to its start
        assert second.data_ptr() == first.data_ptr() and ops._COUNTER_RINGS["cpu"].ints.numel() == 8
    finally:
        ops.WGRAD_COUNTER_RING = saved
        ops._COUNTER_RINGS.pop("cpu", None)
