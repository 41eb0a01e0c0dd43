"""CPU: the oracle (oracle/frcnn_oracle.py) against vectors produced by the reference itself
(tests/golden/make_golden.py) — this is what pins the oracle."""
import hashlib
import os

import numpy as np
import torch

from oracle import frcnn_oracle as O

SCALES, RATIOS = (2, 4, 8, 16, 32), (0.5, 0.75, 1, 1.25, 2)


def _npz(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_anchor_table_known_answer():
    # lib/layer_utils/generate_anchors.py:30-38: the MATLAB table in the comment is 1-based; python = table - 1
    matlab = np.array([[-83, -39, 100, 56], [-175, -87, 192, 104], [-359, -183, 376, 200], [-55, -55, 72, 72],
                       [-119, -119, 136, 136], [-247, -247, 264, 264], [-35, -79, 52, 96], [-79, -167, 96, 184],
                       [-167, -343, 184, 360]], dtype=np.float64)
    np.testing.assert_array_equal(O.generate_anchors(), matlab - 1)


def test_anchors_match_reference(golden_dir):
    g = _npz(golden_dir, "anchors.npz")
    np.testing.assert_array_equal(O.generate_anchors(), g["default9"])
    np.testing.assert_array_equal(O.generate_anchors(ratios=np.array(RATIOS), scales=np.array(SCALES)), g["waymo25"])
    a, n = O.generate_anchors_pre(38, 63, 16, SCALES, RATIOS)
    assert n == 59850 and a.dtype == np.float32
    np.testing.assert_array_equal(a, g["pre_38x63_s16"])
    np.testing.assert_array_equal(a[0], [-15, -4, 30, 19])
    np.testing.assert_array_equal(O.generate_anchors_pre(19, 32, 16, SCALES, RATIOS, 0.5)[0], g["pre_19x32_s16_fs0.5"])
    np.testing.assert_array_equal(O.generate_anchors_pre(5, 7, 16, SCALES, RATIOS, 0.3)[0], g["pre_5x7_s16_fs0.3"])
    fpn = O.generate_anchors_pre(150, 250, 4, SCALES, RATIOS)[0]
    assert hashlib.sha256(fpn.tobytes()).hexdigest() == bytes(g["pre_150x250_s4_sha256"]).decode()
    np.testing.assert_array_equal(fpn[::9973], g["pre_150x250_s4_probe"])


def test_box_codec_matches_reference(golden_dir):
    g = _npz(golden_dir, "box_codec.npz")
    boxes, d1, d2, gt = (torch.from_numpy(g[k]) for k in ("boxes", "deltas1", "deltas2", "gt"))
    # exp/log come from the CPU's vector math library: allow 2 ulp across hosts, exact otherwise
    tol = dict(rtol=3e-7, atol=1e-5)
    np.testing.assert_allclose(O.bbox_transform_inv(boxes, d1).numpy(), g["inv1"], **tol)
    np.testing.assert_allclose(O.bbox_transform_inv(boxes, d2).numpy(), g["inv2"], **tol)
    np.testing.assert_allclose(O.bbox_transform_inv(boxes, d2, 0.5).numpy(), g["inv2_scale0.5"], **tol)
    np.testing.assert_allclose(O.clip_boxes(O.bbox_transform_inv(boxes, d1), g["info"]).numpy(), g["clip1"], **tol)
    np.testing.assert_allclose(O.clip_boxes(O.bbox_transform_inv(boxes, d2), g["info2"]).numpy(), g["clip2_info2"], **tol)
    np.testing.assert_allclose(O.bbox_transform(boxes, gt).numpy(), g["fwd"], rtol=3e-7, atol=1e-6)
    np.testing.assert_allclose(O.bbox_overlaps(boxes[:64], gt[:48]).numpy(), g["overlaps"], rtol=3e-7, atol=1e-7)
    # known answers quoted in SURVEY.md §8c
    z = O.bbox_transform_inv(torch.tensor([[10., 20, 49, 39]]), torch.zeros(1, 4))
    np.testing.assert_array_equal(z.numpy(), [[10, 20, 50, 40]])
    k = O.bbox_transform_inv(torch.tensor([[0., 0, 15, 15]]), torch.tensor([[0.1, -0.2, 0.3, 0.0]]))
    np.testing.assert_allclose(k.numpy(), [[-0.5361, -4.5255, 21.0616, 11.4745]], atol=1e-4)


def test_resnet101_stages_match_reference(golden_dir):
    g = _npz(golden_dir, "resnet101_stages.npz")
    net = O.ResNet101()
    net.eval()
    for mode, seed in (("random", 11), ("tame", 12)):
        net.load_state_dict(O.seeded_state_dict(net, seed, bn_mode=mode, all_backbone=True), strict=True)
        x = torch.from_numpy(g[mode + "_x"])
        with torch.no_grad():
            stem = net.stem(x)
            l1 = net.layer1(stem)
            l2 = net.layer2(l1)
            l3 = net.layer3(l2)
            l4 = net.layer4(torch.from_numpy(g[mode + "_pooled"]))
        for name, got in (("stem", stem), ("layer1", l1), ("layer2", l2), ("layer3", l3)):
            ref = g[mode + "_" + name]
            np.testing.assert_allclose(got.numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max(), err_msg=name)
        ref = g[mode + "_layer4_probe"]
        np.testing.assert_allclose(l4.numpy()[:, ::16], ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
        ref = g[mode + "_layer4_mean"]
        np.testing.assert_allclose(l4.mean(3).mean(2).numpy(), ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())


def test_nms_known_answers():
    # hand-computed: IoU([0,0,10,10],[0,0,10,5]) = 0.5 exactly; [0,0,10,7] vs first = 0.7
    boxes = torch.tensor([[0., 0, 10, 10], [0, 0, 10, 5], [0, 0, 10, 7], [20, 20, 30, 30], [0, 0, 10, 7.0001]])
    scores = torch.tensor([0.9, 0.8, 0.7, 0.6, 0.5])
    # IoU == threshold: suppressed under the default (torchvision 0.4.0's CPU kernel, `>=`), kept under its CUDA kernel's `>`
    assert O.NMS_SUPPRESS_AT_EQUAL is True
    assert O.nms(boxes, scores, 0.5).tolist() == [0, 3]
    O.NMS_SUPPRESS_AT_EQUAL = False
    try:
        assert O.nms(boxes, scores, 0.5).tolist() == [0, 1, 3]      # 0.5 is NOT > 0.5 -> box 1 stays
    finally:
        O.NMS_SUPPRESS_AT_EQUAL = True
    assert O.nms(boxes, scores, 0.49).tolist() == [0, 3]
    # equal scores: ties resolved by ascending index
    assert O.nms(boxes[[1, 0]], torch.tensor([0.5, 0.5]), 0.4).tolist() == [0]
    # zero-area boxes give NaN IoU -> never suppressed
    deg = torch.tensor([[5., 5, 5, 5], [5, 5, 5, 5]])
    assert O.nms(deg, torch.tensor([0.9, 0.8]), 0.5).tolist() == [0, 1]
    assert O.nms(torch.zeros(0, 4), torch.zeros(0), 0.5).numel() == 0


def test_nms_matches_bruteforce():
    g = torch.Generator().manual_seed(5)
    xy = torch.rand(400, 2, generator=g) * 100
    wh = torch.rand(400, 2, generator=g) * 60 + 1
    boxes = torch.cat((xy, xy + wh), 1)
    scores = torch.rand(400, generator=g)
    scores[10:20] = scores[10]  # ties
    order = sorted(range(400), key=lambda i: (-scores[i].item(), i))
    b = boxes.numpy()
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep = []
    for i in order:
        ok = True
        for j in keep:
            w = max(np.float32(0), min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0]))
            h = max(np.float32(0), min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1]))
            inter = np.float32(w * h)
            if inter / (area[j] + area[i] - inter) > np.float32(0.6):
                ok = False
                break
        if ok:
            keep.append(i)
    assert O.nms(boxes, scores, 0.6).tolist() == keep


def test_roi_align_known_answers():
    # constant feature map -> every bin equals the constant; linear ramp -> bilinear sampling is exact
    feat = torch.full((1, 2, 10, 12), 3.0)
    rois = torch.tensor([[0, 16., 16, 100, 90], [0, 0, 0, 8, 8]])
    out = O.roi_align(feat, rois, 7, 1 / 16.0, 0)
    np.testing.assert_allclose(out.numpy(), 3.0, rtol=1e-6)
    yy, xx = torch.meshgrid(torch.arange(10.), torch.arange(12.), indexing="ij")
    ramp = (2 * xx + 3 * yy).view(1, 1, 10, 12)
    roi = torch.tensor([[0, 32., 16, 96, 80]])      # [2,6] x [1,5] on the stride-16 map
    out = O.roi_align(ramp, roi, 2, 1 / 16.0, 2)
    # bin centres: x in {3,5}, y in {2,4}  (mean of the 2x2 sample grid inside each bin)
    np.testing.assert_allclose(out.numpy()[0, 0], [[2 * 3 + 3 * 2, 2 * 5 + 3 * 2], [2 * 3 + 3 * 4, 2 * 5 + 3 * 4]], rtol=1e-6)
    # a roi smaller than one bin is widened to 1 px (max(., 1)); a roi outside the map reads zeros
    tiny = O.roi_align(feat, torch.tensor([[0, 40., 40, 41, 41]]), 7, 1 / 16.0, 0)
    np.testing.assert_allclose(tiny.numpy(), 3.0, rtol=1e-6)
    outside = O.roi_align(feat, torch.tensor([[0, 4000., 4000, 4100, 4100]]), 7, 1 / 16.0, 0)
    np.testing.assert_array_equal(outside.numpy(), 0.0)


def test_proposal_layer_shapes_and_order():
    g = torch.Generator().manual_seed(0)
    h, w, a = 6, 9, 25
    anchors = torch.from_numpy(O.generate_anchors_pre(h, w, 16, SCALES, RATIOS)[0])
    prob = torch.rand(1, h, w, 2 * a, generator=g)
    deltas = torch.randn(1, h, w, 4 * a, generator=g) * 0.3
    info = np.array([0, 144, 0, 96, 0, 0, 1.0], dtype=np.float32)
    rois, scores, dbg = O.proposal_layer(prob, deltas, info, anchors, a, 600, 50, 0.7, return_debug=True)
    assert rois.shape[1] == 5 and rois.shape[0] == scores.shape[0] <= 50
    assert (rois[:, 0] == 0).all()
    assert (scores[:-1, 0] >= scores[1:, 0]).all()
    assert rois[:, 1].min() >= 0 and rois[:, 3].max() <= 143 and rois[:, 4].max() <= 95


# ------------------------------------------------------------------------------------------------
# LiDAR variant + training-target layers against vectors produced by the reference
# (tests/golden/make_golden_lidar_train.py)
# ------------------------------------------------------------------------------------------------
def _lt(golden_dir):
    return np.load(os.path.join(golden_dir, "lidar_train.npz"))


def test_3d_anchors_and_bev_boxes_match_reference(golden_dir):
    z = _lt(golden_dir)
    for tag, (h, w, fs) in {"25x22_fs0.5": (25, 22, 0.5), "50x44_fs1": (50, 44, 1.0), "7x5_fs0.3": (7, 5, 0.3)}.items():
        n, a3 = O.generate_anchors_3d(h, w, 16, frame_scale=fs)
        assert n == z["a3d_" + tag].shape[0]
        np.testing.assert_array_equal(a3, z["a3d_" + tag])
        np.testing.assert_array_equal(O.bbaa_graphics_gems(a3), z["a2d_" + tag])


def test_lidar_codec_matches_reference(golden_dir):
    z = _lt(golden_dir)
    rois, anc, d, gt = (torch.from_numpy(z[k]) for k in ("l_rois", "l_anchors", "l_deltas", "l_gt"))
    np.testing.assert_array_equal(O.lidar_3d_bbox_transform_inv(rois, anc, d).numpy(), z["l_inv"])
    np.testing.assert_array_equal(O.lidar_3d_bbox_transform_inv(rois, anc, d, 0.5).numpy(), z["l_inv_scale0.5"])
    np.testing.assert_array_equal(O.lidar_3d_bbox_transform(rois, anc, gt).numpy(), z["l_fwd"])
    np.testing.assert_array_equal(np.asarray(O.lidar_extents(), np.float32), z["l_extents"])
    got = O.bbox_voxel_grid_to_pc(z["l_inv"][:, 7:14].copy(), O.lidar_extents(), z["l_info"])
    np.testing.assert_array_equal(got, z["l_vg_to_pc"])


def test_proposal_top_layer_matches_reference(golden_dir):
    z = _lt(golden_dir)
    blob, sc, anc = O.proposal_top_layer(torch.from_numpy(z["top_prob"]), torch.from_numpy(z["top_deltas"]),
                                         z["top_info"], torch.from_numpy(z["top_anchors"]), 25, rpn_top_n=120)
    np.testing.assert_array_equal(blob.numpy(), z["top_blob"])
    np.testing.assert_array_equal(sc.numpy(), z["top_scores"])
    np.testing.assert_array_equal(anc.numpy(), z["top_sel_anchors"])


def test_anchor_target_layer_matches_reference(golden_dir):
    z = _lt(golden_dir)
    h, w = (int(v) for v in z["atl_hw"])
    lab, tgt, inw, outw = O.anchor_target_layer(torch.from_numpy(z["atl_gt"]), z["atl_info"],
                                                torch.from_numpy(z["atl_anchors"]), 25, h, w, rpn_batchsize=10 ** 7)
    np.testing.assert_array_equal(lab.numpy(), z["atl_labels"])
    np.testing.assert_array_equal(tgt.numpy(), z["atl_targets"])
    np.testing.assert_array_equal(inw.numpy(), z["atl_inside"])
    np.testing.assert_array_equal(outw.numpy(), z["atl_outside"])
    assert (z["atl_labels"] == 1).sum() >= 6 and (z["atl_labels"] == 0).sum() > 100   # the case is not degenerate


def test_proposal_target_layer_matches_reference(golden_dir):
    z = _lt(golden_dir)
    rois, sc, gt = (torch.from_numpy(z[k]) for k in ("ptl_rois", "ptl_scores", "ptl_gt"))
    lab, r, _, s, bt, biw, bow = O.proposal_target_layer(rois, sc, torch.zeros(rois.shape[0], 7), gt,
                                                         torch.zeros(4, 8), 2, 4,
                                                         generator=torch.Generator().manual_seed(1))
    packed = torch.cat((r, lab, s.view(-1, 1), bt, biw, bow), 1).numpy()
    packed = packed[np.lexsort(packed.T[::-1])]
    # every candidate is taken (40 fg == quota share, 216 bg == the rest), so only the row order is random
    np.testing.assert_array_equal(packed, z["ptl_packed_sorted"])
    assert int((lab > 0).sum()) == int(z["ptl_num_fg"][0]) == 40


def test_proposal_target_layer_lidar_matches_reference(golden_dir):
    """NET_TYPE 'lidar': 7-of-7K targets = lidar_3d_bbox_transform(roi, 3-D anchor, true gt) / LiDAR stds."""
    z = _lt(golden_dir)
    rois, sc, gt, a3, tgt = (torch.from_numpy(z[k]) for k in ("ptl_rois", "ptl_scores", "ptl_gt", "ptl_lidar_anchors_3d",
                                                             "ptl_lidar_true_gt"))
    lab, r, a3s, s, bt, biw, bow = O.proposal_target_layer(rois, sc, a3, gt, tgt, 2, 7, net_type="lidar",
                                                           generator=torch.Generator().manual_seed(1))
    packed = torch.cat((r, lab, s.view(-1, 1), a3s, bt, biw, bow), 1).numpy()
    packed = packed[np.lexsort(packed.T[::-1])]
    np.testing.assert_array_equal(packed, z["ptl_lidar_packed_sorted"])


def test_losses_match_reference(golden_dir):
    z = _lt(golden_dir)
    p, t, iw, ow = (torch.from_numpy(a) for a in z["sl1_rpn_in"])
    assert abs(O.smooth_l1_loss("RPN", p, t, iw, ow, dim=(1, 2, 3)).item() - z["sl1_rpn"][0]) <= 1e-7
    p, t, iw, ow = (torch.from_numpy(a) for a in z["sl1_det_in"])
    assert abs(O.smooth_l1_loss("DET", p, t, iw, ow).item() - z["sl1_det"][0]) <= 1e-7
    p, t = (torch.from_numpy(a) for a in z["huber_in"])
    np.testing.assert_array_equal(O.huber_loss(p, t).numpy(), z["huber"])
    np.testing.assert_array_equal(O.huber_loss(p, t, sin_en=True).numpy(), z["huber_sin"])
    p, t, iw, ow = (torch.from_numpy(a) for a in z["sl1_lidar_in"])
    assert abs(O.smooth_l1_loss("DET", p, t, iw, ow, net_type="lidar").item() - z["sl1_lidar"][0]) <= 1e-7


def test_uncertainty_statistics_and_aleatoric_loss_match_reference(golden_dir):
    """loss_utils.py:82-85,114-141 (imported reference -> lidar_train.npz): MC variance / entropy / mutual information
    and the aleatoric smooth-L1 with its gradients w.r.t. the prediction and the log-variance."""
    z = _lt(golden_dir)
    np.testing.assert_array_equal(O.compute_bbox_var(torch.from_numpy(z["mc_bbox_samples"])).numpy(), z["mc_bbox_var"])
    cls = torch.from_numpy(z["mc_cls_samples"])
    np.testing.assert_array_equal(O.categorical_mutual_information(cls).numpy(), z["mc_mutual_info"])
    np.testing.assert_array_equal(O.categorical_entropy(torch.softmax(cls, 2).mean(0)).numpy(), z["mc_entropy"])
    p, t, v, iw, ow = (torch.from_numpy(a) for a in z["sl1_al_in"])
    p.requires_grad_(True)
    v.requires_grad_(True)
    loss = O.smooth_l1_loss("DET", p, t, iw, ow, bbox_var=v)
    assert abs(loss.item() - z["sl1_al"][0]) <= 1e-7
    loss.backward()
    np.testing.assert_array_equal(p.grad.numpy(), z["sl1_al_dpred"])
    np.testing.assert_array_equal(v.grad.numpy(), z["sl1_al_dvar"])


def test_uncertainty_inverse_transforms_pinned(golden_dir):
    """O.uncertainty_transform_inv / O.lidar_3d_uncertainty_transform_inv equal the imported reference bit for bit
    (lib/model/bbox_transform.py:107-169); the image form's missing unsqueeze is pinned as well: with N == K == 2 boxes the
    reference scales class j by the size of BOX j."""
    import os
    z = np.load(os.path.join(golden_dir, "uc_inv.npz"))
    r, a, d, u = [torch.from_numpy(z[k]) for k in ("rois", "anchors", "deltas", "uc")]
    for tag, sc in (("", None), ("_scale0.5", 0.5)):
        np.testing.assert_array_equal(O.lidar_3d_uncertainty_transform_inv(r, a, d, u, sc).numpy(), z["lidar" + tag])
        np.testing.assert_array_equal(O.uncertainty_transform_inv(r, d, u, sc).numpy(), z["bev" + tag])
    q = z["bev_quirk_n2"]                     # what the reference returns for two boxes at once
    ln = z["rois"][:2, 2] - z["rois"][:2, 0] + 1
    np.testing.assert_allclose(q[:, 0::4], (z["uc"][:2, 0::7] * ln[None, :]) ** 2, rtol=1e-6)      # box j, not box i
    assert not np.allclose(q, z["bev"][:2])
