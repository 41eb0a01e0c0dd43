"""datasets/waymo_eval.py (lib/datasets/waymo_eval.py:44-247 + the call-site contract of the missing eval_utils):
protocol properties and a cross-check against the VOC evaluator, which IS pinned to the imported reference
(tests/golden/eval.npz).  The reference module itself cannot be imported (shapely, eval_utils), so this file documents
the restated behaviour rather than pinning it."""
import os

import numpy as np

from faster_rcnn_pytorch_multimodal_amd.datasets import voc_eval as V
from faster_rcnn_pytorch_multimodal_amd.datasets import waymo_eval as Wm
from faster_rcnn_pytorch_multimodal_amd.model import config as C


def _scene(seed, frames=6, gt_per_frame=4):
    rng = np.random.default_rng(seed)
    recs, names, dets = [], [], []
    for f in range(frames):
        xy = rng.uniform(0, 800, (gt_per_frame, 2))
        wh = rng.uniform(40, 200, (gt_per_frame, 2))
        boxes = np.concatenate((xy, xy + wh), 1)
        name = "%06d.png" % f
        names.append(name)
        recs.append(Wm.make_rec(name, boxes, difficulty=rng.integers(1, 4, gt_per_frame)))
        for b in boxes:                                   # a jittered detection per box + a duplicate + clutter
            for _ in range(2):
                dets.append((f, name, rng.uniform(0.2, 1.0), b + rng.normal(0, 6, 4)))
        for _ in range(3):
            p = rng.uniform(0, 800, 2)
            dets.append((f, name, rng.uniform(0.05, 0.6), np.concatenate((p, p + rng.uniform(30, 120, 2)))))
    return recs, names, dets


def _write(path, dets):
    with open(path, "w") as fh:
        for f, name, score, b in dets:
            fh.write("%d %s %.6f %.3f %.3f %.3f %.3f\n" % (f, name, score, b[0], b[1], b[2], b[3]))


def test_coincides_with_voc_eval_when_the_protocols_agree(tmp_path):
    """All boxes difficulty 1, nothing ignored, no don't-care boxes, every frame has ground truth: the Waymo loop is
    the VOC loop (strict '>' threshold, first match wins, duplicates are false positives), so AP must be equal."""
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    recs, names, dets = _scene(3)
    for r in recs:
        r["difficulty"][:] = 1
    path = os.path.join(tmp_path, "det.txt")
    _write(path, dets)
    _, _, m, plain = Wm.waymo_eval(path, recs, ovthresh=0.5, eval_type="2d", d_levels=2, bbox_elem=4)
    # the file rounds scores / boxes: score the same rounded rows with the VOC evaluator
    idx, tok, score, box, _ = V.read_results_file(path, 4)
    voc_recs = {r["filename"]: {"bbox": r["boxes"], "difficult": np.zeros(len(r["boxes"]), bool)} for r in recs}
    rec, prec, ap = V.voc_eval_arrays(tok, score, box, voc_recs, ovthresh=0.5)
    assert abs(m[0] - ap) < 1e-12 and abs(m[1] - ap) < 1e-12
    assert 0.2 < ap < 1.0
    np.testing.assert_array_equal(plain["npos"].sum(0), [24, 24])


def test_difficulty_levels_ignore_flags_and_the_aliased_return(tmp_path):
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    recs, names, dets = _scene(5)
    recs[0]["ignore"][0] = True                         # a hit on it is dropped, it does not count as a positive
    recs[1]["ignore_frame"] = True                      # its detections are skipped without using a row
    path = os.path.join(tmp_path, "det.txt")
    _write(path, dets)
    mrec, mprec, m, plain = Wm.waymo_eval(path, recs, ovthresh=0.5, eval_type="2d", d_levels=2, bbox_elem=4)
    assert mrec is mprec is m                           # `map = mrec = mprec = np.zeros(...)` (waymo_eval.py:232)
    lvl1 = sum(int(((r["difficulty"] <= 1) & ~r["ignore"]).sum()) for r in recs if not r["ignore_frame"])
    lvl2 = sum(int(((r["difficulty"] <= 2) & ~r["ignore"]).sum()) for r in recs if not r["ignore_frame"])
    np.testing.assert_array_equal(plain["npos"].sum(0), [lvl1, lvl2])
    # level-1 true positives are a subset of level-2 ones; difficulty-3 boxes are positives at neither level
    assert (plain["tp"][:, 0] <= plain["tp"][:, 1]).all() and plain["tp"][:, 1].sum() <= lvl2
    used_rows = int(((plain["tp"].sum(1) + plain["fp"].sum(1)) > 0).sum())
    assert used_rows <= sum(1 for d in dets if d[1] != names[1])
    assert 0.0 <= m[0] <= 1.0 and 0.0 <= m[1] <= 1.0


def test_dont_care_boxes_and_frames_without_ground_truth(tmp_path):
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    gt = np.array([[100., 100, 200, 200]])
    dc = np.array([[400., 400, 500, 500]])
    recs = [Wm.make_rec("a", gt, boxes_dc=dc), Wm.make_rec("b", np.zeros((0, 4)), ignore_frame=False)]
    dets = [(0, "a", 0.9, gt[0]), (0, "a", 0.8, dc[0]), (1, "b", 0.7, np.array([10., 10, 50, 50]))]
    path = os.path.join(tmp_path, "det.txt")
    _write(path, dets)
    _, _, m_off, p_off = Wm.waymo_eval(path, recs, bbox_elem=4, ignore_dc=False)
    _, _, m_on, p_on = Wm.waymo_eval(path, recs, bbox_elem=4, ignore_dc=True)
    assert p_off["fp"][:, 0].sum() == 1 and p_on["fp"][:, 0].sum() == 0      # the detection on the don't-care box
    assert p_off["tp"][:, 0].sum() == 1 and p_on["tp"][:, 0].sum() == 1
    assert m_on[0] == 1.0 and m_off[0] == 1.0          # the false positive ranks below the only true positive


def test_overlap_flavours_known_answers():
    a = np.array([[0., 0, 9, 9]])
    assert abs(Wm.iou(a, np.array([5., 0, 14, 9]), "2d")[0] - 50.0 / 150.0) < 1e-12          # +1 pixel convention
    box = lambda xc, yc, l, w, ry, zc=0.0, h=2.0: np.array([xc, yc, zc, l, w, h, ry])
    g = box(0, 0, 4, 2, 0)[None]
    assert abs(Wm.iou(g, box(2, 0, 4, 2, 0), "bev_aa")[0] - 4.0 / 12.0) < 1e-12
    assert abs(Wm.iou(g, box(2, 0, 4, 2, 0), "bev")[0] - 4.0 / 12.0) < 1e-9
    assert abs(Wm.iou(g, box(0, 0, 4, 2, np.pi / 2), "bev")[0] - 4.0 / 12.0) < 1e-9           # 2x2 overlap of two 4x2 bars
    assert abs(Wm.iou(g, box(0, 0, 4, 2, np.pi), "bev")[0] - 1.0) < 1e-9
    assert abs(Wm.iou(g, box(0, 0, 4, 2, 0, zc=1.0), "3d")[0] - (8.0 * 1.0) / (16.0 + 16.0 - 8.0)) < 1e-9


def test_uncertainty_columns_of_a_results_line():
    C.reset_cfg()
    C.cfg.UC.EN_BBOX_ALEATORIC = True
    C.cfg.UC.EN_CLS_EPISTEMIC = True
    line = "0 f 0.9 1 2 3 4".split() + ["0.5", "0.25", "0.1", "0.2"] + ["7", "8", "9", "10"]
    _, unc = Wm.extract_uncertainties(4, [line])
    assert list(unc.keys()) == ["e_entropy", "e_mutual_info", "e_cls_var", "a_bbox_var"]
    np.testing.assert_array_equal(unc["e_cls_var"], [[0.1, 0.2]])
    np.testing.assert_array_equal(unc["a_bbox_var"], [[7, 8, 9, 10]])
    C.reset_cfg()
