"""cfg.UC.* uncertainty heads on the device against the CPU oracle (which replays the device's counter-based random
draws): dropout masks bit-exact, distorted logits / losses / statistics within fp32 tolerances written below.

Reference anchors: module names and rates lib/nets/imagenet.py:52-91, lib/nets/lidarnet.py:56-102; protocol
lib/model/test.py:74-77,151-159,260-270; arithmetic lib/utils/loss_utils.py:82-85,114-169; per-detection gather
lib/utils/filter_predictions.py:23-43,113-124.
"""
import numpy as np
import pytest
import torch

from oracle import frcnn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ALL_FLAGS = dict(EN_BBOX_EPISTEMIC=True, EN_CLS_EPISTEMIC=True, EN_BBOX_ALEATORIC=True, EN_CLS_ALEATORIC=True)


def _set_flags(C, flags):
    for k, v in flags.items():
        C.cfg.UC[k] = v


def test_dropout_masks_and_normal_draws_replay(hip):
    from faster_rcnn_pytorch_multimodal_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(300, 1024, generator=g)
    for p, seed, stream, repeat in ((0.1, 3, 11, 1), (0.3, 99, 13, 10), (0.5, 7, 12, 4)):
        y = ops.dropout(x.to(DEV), p, seed, stream, repeat=repeat).cpu()
        ref = O.dropout_replay(x, p, seed, stream, repeat=repeat)
        assert torch.equal(y, ref)                                     # same masks, same fp32 scaling
        keep = float((y != 0).float().mean())
        assert abs(keep - (1 - p)) < 0.01
        # backward: gradient through the same masks, summed over the copies
        dy = torch.randn(y.shape, generator=g)
        dx = ops.dropout_bwd(dy.to(DEV), p, seed, stream, repeat=repeat).cpu()
        mask = (ref != 0).float() / (1 - np.float32(p))
        want = (dy * mask).reshape(repeat, *x.shape).sum(0) if repeat > 1 else dy * mask
        np.testing.assert_allclose(dx.numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
    score, logvar = torch.randn(300, 2, generator=g), torch.randn(300, 2, generator=g) * 0.5
    samples, var = ops.logit_distort(score.to(DEV), logvar.to(DEV), 200, 5, 15, var_is_log=True)
    ref, eps = O.logit_distort_replay(score, torch.exp(logvar), 200, 5)
    np.testing.assert_allclose(var.cpu().numpy(), torch.exp(logvar).numpy(), rtol=2e-6)
    np.testing.assert_allclose(samples.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)      # logf / cosf / expf last-ulp differences
    assert abs(float(eps.mean())) < 0.01 and abs(float(eps.std()) - 1.0) < 0.01          # the draws are standard normal


def test_bayesian_cross_entropy_matches_oracle_autograd(hip):
    from faster_rcnn_pytorch_multimodal_amd import ops
    g = torch.Generator().manual_seed(1)
    n, k, s = 256, 2, 200
    score = torch.randn(n, k, generator=g)
    logvar = torch.randn(n, k, generator=g) * 0.7 - 0.5
    labels = torch.randint(0, k, (n,), generator=g).float()
    loss, dscore, dlogvar = ops.bayesian_cross_entropy(score.to(DEV), logvar.to(DEV), labels.to(DEV), s, 11, 16, grad=1.0,
                                                       var_is_log=True)
    sc = score.double().requires_grad_(True)
    lv = logvar.double().requires_grad_(True)
    ref, _ = O.bayesian_cross_entropy(sc, torch.exp(lv), labels, s, 11)
    ref.backward()
    assert abs(float(loss.item()) - float(ref.detach())) < 2e-5
    np.testing.assert_allclose(dscore.cpu().numpy(), sc.grad.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(dlogvar.cpu().numpy(), lv.grad.numpy(), rtol=0, atol=2e-6)
    # variances given directly (var_is_log = False) take the other derivative branch
    var = torch.exp(logvar)
    loss2, _, dvar = ops.bayesian_cross_entropy(score.to(DEV), var.to(DEV), labels.to(DEV), s, 11, 16, var_is_log=False)
    v = var.double().requires_grad_(True)
    ref2, _ = O.bayesian_cross_entropy(score.double(), v, labels, s, 11)
    ref2.backward()
    assert abs(float(loss2.item()) - float(ref2.detach())) < 2e-5
    np.testing.assert_allclose(dvar.cpu().numpy(), v.grad.numpy(), rtol=0, atol=5e-6)


def _build_uc_net(flags, lidar=False, seed=31):
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.utils.init_utils import seeded_state_dict
    C.reset_cfg()
    C.cfg.NET_TYPE = "lidar" if lidar else "image"
    _set_flags(C, flags)
    if lidar:
        from faster_rcnn_pytorch_multimodal_amd.nets.lidarnet import lidarnet
        net = lidarnet(num_layers=101)
        net.create_architecture(2, tag="default", anchor_scales=C.cfg.LIDAR.ANCHOR_SCALES[0],
                                anchor_ratios=C.cfg.LIDAR.ANCHOR_ANGLES)
    else:
        from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
        net = imagenet(num_layers=101)
        net.create_architecture(2, tag="default", anchor_scales=C.cfg.ANCHOR_SCALES, anchor_ratios=C.cfg.ANCHOR_RATIOS)
    sd = seeded_state_dict(net, seed, bn_mode="tame")
    g = torch.Generator().manual_seed(seed)
    for name in sd:                       # give the new heads non-degenerate weights (their reference init is N(0, 0.01))
        if any(t in name for t in ("_fc1", "_fc2", "al_var_net", "cls_score_net", "bbox_pred_net")) and name.endswith("weight"):
            sd[name] = torch.randn(sd[name].shape, generator=g) * (0.02 if "fc" in name else 0.004)
        if any(t in name for t in ("bbox_bn", "cls_bn")):
            if name.endswith("running_var"):
                sd[name] = torch.rand(sd[name].shape, generator=g) + 0.5
            elif name.endswith("running_mean") or name.endswith("bias"):
                sd[name] = torch.randn(sd[name].shape, generator=g) * 0.1
            elif name.endswith("weight"):
                sd[name] = torch.rand(sd[name].shape, generator=g) + 0.5
    net.load_state_dict(sd, strict=True)
    net.eval()
    net._device = DEV
    net.to(DEV)
    return net, sd, C


def _heads_oracle(sd, flags, lidar, bbox_elem):
    h = O.UcHeadsOracle(flags, num_classes=2, bbox_elem=bbox_elem, lidar=lidar)
    own = h.state_dict()
    h.load_state_dict({k: sd[k] for k in own}, strict=True)
    return h


@pytest.mark.parametrize("flags", [ALL_FLAGS, dict(EN_BBOX_EPISTEMIC=True, EN_CLS_EPISTEMIC=True),
                                   dict(EN_BBOX_ALEATORIC=True, EN_CLS_ALEATORIC=True), dict(EN_BBOX_ALEATORIC=True)])
def test_image_detector_uncertainty_heads_against_oracle(hip, flags):
    from faster_rcnn_pytorch_multimodal_amd.model.test import detect_frame_device, frame_detect
    from faster_rcnn_pytorch_multimodal_amd.nets import uncertainty as U
    net, sd, C = _build_uc_net(flags)
    try:
        epistemic = bool(flags.get("EN_BBOX_EPISTEMIC"))
        # module names / sizes of the reference (imagenet.py:52-91)
        keys = set(sd.keys())
        assert ("bbox_fc1.weight" in keys) == epistemic and ("cls_fc2.bias" in keys) == epistemic
        assert ("bbox_al_var_net.weight" in keys) == bool(flags.get("EN_BBOX_ALEATORIC"))
        assert ("cls_al_var_net.weight" in keys) == bool(flags.get("EN_CLS_ALEATORIC"))
        assert net.cls_score_net.in_features == (512 if epistemic else 2048)
        if epistemic:
            assert net.bbox_fc1.weight.shape == (1024, 2048) and net.bbox_fc2.weight.shape == (512, 1024)
            assert net.bbox_drop1.p == 0.1 and net.cls_drop1.p == 0.3 and net.bbox_drop1.training      # eval() keeps them stochastic
        data = (np.random.default_rng(4).standard_normal((1, 128, 192, 3)) * 50).astype(np.float32)
        info = np.array([0, 192, 0, 128, 0, 0, 1.0], np.float32)
        net.set_uc_seed(77)
        net.set_e_num_sample(C.cfg.UC.E_NUM_SAMPLE if epistemic else 1)
        cls_score, cls_prob, pred_boxes, rois, unc = net.test_frame(data, info)
        net.set_e_num_sample(1)
        n = rois.shape[0]
        assert list(unc.keys()) == [k for k in U.UNCERTAINTY_ORDER if (
            (k.startswith("a_") and "bbox" in k and flags.get("EN_BBOX_ALEATORIC")) or
            (k.startswith("a_") and "bbox" not in k and flags.get("EN_CLS_ALEATORIC")) or
            (k.startswith("e_") and "bbox" in k and flags.get("EN_BBOX_EPISTEMIC")) or
            (k.startswith("e_") and "bbox" not in k and flags.get("EN_CLS_EPISTEMIC")))]
        # the heads on the device's own fc7 (backbone / RoIAlign / layer4 parity is covered elsewhere)
        p = net._predictions
        fc7 = p["fc7_uc"].cpu()
        h = _heads_oracle(sd, flags, False, 4)
        cs_r, cp_r, bp_r, deltas_r, unc_r = h.test(fc7[:n], C.cfg.UC.E_NUM_SAMPLE if epistemic else 1, C.cfg.UC.A_NUM_CE_SAMPLE,
                                                   77, O.BBOX_NORMALIZE_STDS, O.BBOX_NORMALIZE_MEANS)
        rel = lambda ref: 2e-5 * max(1.0, float(ref.abs().max()))       # fp32 GEMM noise scales with the magnitude
        np.testing.assert_allclose(cls_prob.cpu().numpy(), cp_r.numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(cls_score.cpu().numpy(), cs_r.numpy(), rtol=0, atol=rel(cs_r))
        np.testing.assert_allclose(p["bbox_pred"][:n].cpu().numpy(), bp_r.numpy(), rtol=0, atol=rel(bp_r))
        assert 0.05 < float(cp_r[:, 1].std()) and float(cp_r.max()) < 0.9999    # the case is not saturated
        pb_r = O.bbox_transform_inv(rois[:, 1:5].cpu(), deltas_r, 1.0)
        np.testing.assert_allclose(pred_boxes.cpu().numpy(), pb_r.numpy(), rtol=0, atol=2e-3)
        for k, v in unc_r.items():
            got = unc[k].cpu().numpy()
            tol = 2e-5 * max(1.0, float(np.abs(v.numpy()).max()))
            np.testing.assert_allclose(got, v.numpy(), rtol=1e-4, atol=tol, err_msg=k)
        # per-detection columns: reference-shaped call and the device record agree with a host gather
        C.cfg.UC.E_NUM_SAMPLE = 10
        net.set_uc_seed(77)
        rois_np, all_boxes, all_unc = frame_detect(net, {"data": data, "info": info}, 2, 0.05)
        n_uc = U.num_uncertainty_pos(2, 4)
        assert n_uc == sum(v.shape[1] for v in all_unc[1].values())
        assert all(v.shape[0] == len(all_boxes[1]) for v in all_unc[1].values())
        net.set_uc_seed(77)
        dets, counts = detect_frame_device(net, data, info, 0.05, 100, 100)
        assert dets.shape == (2, 100, 5 + n_uc)
        c1 = int(counts[1])
        assert c1 == len(all_boxes[1]) or c1 == 100
        m = min(c1, len(all_boxes[1]))
        np.testing.assert_allclose(dets[1, :m, :5].cpu().numpy(), all_boxes[1][:m], rtol=0, atol=1e-6)
        from faster_rcnn_pytorch_multimodal_amd.model.test import stack_uncertainties
        stacked = stack_uncertainties(all_boxes[1], all_unc[1], n_uc)
        np.testing.assert_allclose(dets[1, :m].cpu().numpy(), stacked[:m], rtol=0, atol=1e-6)
        assert (dets[1, c1:] == 0).all()
    finally:
        C.reset_cfg()


def test_lidar_detector_uncertainty_heads_against_oracle(hip):
    """LiDAR variant: BatchNorm1d after each head Linear (lidarnet.py:85-92), rates 0.5 / 0.2, 7-element boxes."""
    net, sd, C = _build_uc_net(ALL_FLAGS, lidar=True)
    try:
        assert net.bbox_drop1.p == 0.5 and net.cls_drop1.p == 0.2 and net.bbox_bn1.num_features == 1024
        assert net.bbox_pred_net.out_features == 14 and net.bbox_al_var_net.out_features == 14
        rng = np.random.default_rng(2)
        data = (rng.random((1, 208, 176, 15)) * (rng.random((1, 208, 176, 15)) < 0.05)).astype(np.float32)
        info = np.array([0, 176, 0, 208, 0, 12, 0.5], np.float32)
        net.set_uc_seed(5)
        net.set_e_num_sample(10)
        cls_score, cls_prob, pred_boxes, rois, unc = net.test_frame(data, info)
        net.set_e_num_sample(1)
        n = rois.shape[0]
        fc7 = net._predictions["fc7_uc"].cpu()
        h = _heads_oracle(sd, ALL_FLAGS, True, 7)
        _, cp_r, bp_r, _, unc_r = h.test(fc7[:n], 10, C.cfg.UC.A_NUM_CE_SAMPLE, 5, O.LIDAR_BBOX_NORMALIZE_STDS,
                                         O.LIDAR_BBOX_NORMALIZE_MEANS)
        np.testing.assert_allclose(cls_prob.cpu().numpy(), cp_r.numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(net._predictions["bbox_pred"][:n].cpu().numpy(), bp_r.numpy(), rtol=0,
                                   atol=2e-5 * max(1.0, float(bp_r.abs().max())))
        assert unc["e_bbox_var"].shape == (n, 14) and unc["a_cls_var"].shape == (n, 2)
        for k, v in unc_r.items():
            tol = 2e-5 * max(1.0, float(np.abs(v.numpy()).max()))
            np.testing.assert_allclose(unc[k].cpu().numpy(), v.numpy(), rtol=1e-4, atol=tol, err_msg=k)
        # cfg.UC.EN_BBOX_EPISTEMIC_INV_TRANSFORM (lib/model/config.py:42): the same frame and draws, e_bbox_var carried to
        # box space by lidar_3d_uncertainty_transform_inv (lib/model/bbox_transform.py:132-169) on sqrt(variance)
        C.cfg.UC.EN_BBOX_EPISTEMIC_INV_TRANSFORM = True
        net.set_uc_seed(5)
        net.set_e_num_sample(10)
        _, _, _, rois2, unc2 = net.test_frame(data, info)
        net.set_e_num_sample(1)
        assert torch.equal(rois2, rois)
        a3 = net._predictions["roi_anchors_3d"][:n].cpu()
        want = O.lidar_3d_uncertainty_transform_inv(rois[:, 1:5].cpu(), a3, None, torch.sqrt(unc["e_bbox_var"].cpu()), 0.5)
        np.testing.assert_allclose(unc2["e_bbox_var"].cpu().numpy(), want.numpy(), rtol=2e-5, atol=1e-9)
        assert not torch.equal(unc2["e_bbox_var"], unc["e_bbox_var"])
        for k in unc:
            if k != "e_bbox_var":
                assert torch.equal(unc2[k], unc[k]), k
    finally:
        C.reset_cfg()


def test_uncertainty_inverse_transforms_match_reference_vectors(hip, golden_dir):
    """model.bbox_transform.uncertainty_transform_inv / lidar_3d_uncertainty_transform_inv (lib/model/bbox_transform.py:
    107-130, 132-169) against vectors produced by the imported reference (tests/golden/make_golden_uc_inv.py)."""
    import os
    from faster_rcnn_pytorch_multimodal_amd.model.bbox_transform import (lidar_3d_uncertainty_transform_inv,
                                                                         uncertainty_transform_inv)
    z = np.load(os.path.join(golden_dir, "uc_inv.npz"))
    dev = "cuda:0"
    r, a, d, u = [torch.from_numpy(z[k]).to(dev) for k in ("rois", "anchors", "deltas", "uc")]
    for tag, sc in (("", None), ("_scale0.5", 0.5)):
        got = lidar_3d_uncertainty_transform_inv(r, a, d, u, sc).cpu().numpy()
        np.testing.assert_allclose(got, z["lidar" + tag], rtol=3e-6, atol=1e-9)
        got = uncertainty_transform_inv(r, d, u, sc).cpu().numpy()
        np.testing.assert_allclose(got, z["bev" + tag], rtol=3e-6, atol=1e-9)
    # the variance-input form used by the detector: sqrt first
    got = lidar_3d_uncertainty_transform_inv(r, a, d, u, None).cpu().numpy()
    from faster_rcnn_pytorch_multimodal_amd import ops
    got_v = ops.uncertainty_transform_inv(r, (u * u).contiguous(), a, None, lidar=True, input_is_variance=True).cpu().numpy()
    np.testing.assert_allclose(got_v, got, rtol=2e-6, atol=1e-9)


def test_flag_combinations_the_snapshot_does_not_define_are_rejected(hip):
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.nets.imagenet import imagenet
    for flags in (dict(EN_BBOX_EPISTEMIC=True), dict(EN_RPN_CLS_ALEATORIC=True, EN_CLS_ALEATORIC=True),
                  dict(EN_BBOX_EPISTEMIC=True, EN_CLS_EPISTEMIC=True, EN_BBOX_EPISTEMIC_INV_TRANSFORM=True)):
        C.reset_cfg()
        C.cfg.NET_TYPE = "image"
        _set_flags(C, flags)
        net = imagenet(num_layers=101)
        with pytest.raises(NotImplementedError):
            net.create_architecture(2, tag="default", anchor_scales=C.cfg.ANCHOR_SCALES, anchor_ratios=C.cfg.ANCHOR_RATIOS)
    C.reset_cfg()


def test_train_step_with_uncertainty_losses(hip):
    """One training step with all four flags: the second-stage losses equal the oracle's (bayesian cross-entropy on
    replayed draws + variance-attenuated smooth-L1) on the device's own head outputs, every new parameter gets a
    gradient, and the gradients of the loss inputs equal torch autograd's."""
    from faster_rcnn_pytorch_multimodal_amd.nets import uncertainty as U
    net, sd, C = _build_uc_net(ALL_FLAGS, seed=41)
    try:
        C.cfg.TRAIN.USE_GT = True          # ground-truth boxes join the proposals: the batch has foreground RoIs
        net.train()
        rng = np.random.default_rng(8)
        data = (rng.standard_normal((1, 160, 224, 3)) * 50).astype(np.float32)
        info = np.array([0, 224, 0, 160, 0, 0, 1.0], np.float32)
        gt = np.array([[20, 30, 120, 110, 1], [100, 40, 200, 150, 1]], np.float32)
        net.set_uc_seed(123)
        net.zero_grad()
        net.forward(data, info, gt, None, mode="TRAIN")
        loss = net._losses["total_loss"]
        p, pt = net._predictions, net._proposal_targets
        cs = p["cls_score"].detach().cpu().double().requires_grad_(True)
        bp = p["bbox_pred"].detach().cpu().double().requires_grad_(True)
        bv = p["bbox_var"].detach().cpu().double().requires_grad_(True)
        cv = p["cls_var"].detach().cpu().double().requires_grad_(True)
        labels = pt["labels"].cpu()
        ce_ref, _ = O.bayesian_cross_entropy(cs, torch.exp(cv), labels, C.cfg.UC.A_NUM_CE_SAMPLE, p["uc_seed"])
        box_ref = O.smooth_l1_loss("DET", bp, pt["targets"].cpu().double(), pt["inside"].cpu().double(),
                                   pt["outside"].cpu().double(), dim=(1,), net_type="image", bbox_var=bv)
        assert int((labels > 0).sum()) > 0 and float(box_ref) != 0.0
        assert abs(float(net._losses["cross_entropy"]) - float(ce_ref)) < 5e-5
        assert abs(float(net._losses["loss_box"]) - float(box_ref)) < 5e-5 * max(1.0, abs(float(box_ref)))
        net.backward(loss)
        torch.cuda.synchronize()
        for name in ("bbox_fc1", "bbox_fc2", "cls_fc1", "cls_fc2", "bbox_al_var_net", "cls_al_var_net", "cls_score_net",
                     "bbox_pred_net"):
            gparam = getattr(net, name).weight.grad
            assert gparam is not None and torch.isfinite(gparam).all() and float(gparam.abs().max()) > 0, name
    finally:
        C.reset_cfg()
