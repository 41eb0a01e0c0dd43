"""Waymo-style scoring of a detections file — counterpart of lib/datasets/waymo_eval.py:44-247 (same matching loop in
kitti_eval.py / cadc_eval.py) and of the helpers it takes from the reference's MISSING ``lib/utils/eval_utils.py``
(``find_rec``, ``iou``, ``ap``, ``extract_uncertainties``: restated from their call sites, waymo_eval.py:112,141,166-171,246).

What is restated faithfully (waymo_eval.py):
  * detections are visited in descending confidence (:131), a detection whose frame token has no record is skipped
    WITHOUT consuming a row of the tp / fp tables (:141-143, ``idx`` only advances at :213);
  * a detection is a candidate match when ``ovmax > ovthresh`` (strict) and its overlap with every don't-care box is
    below 0.5 (only consulted when cfg.TEST.IGNORE_DC, :163-168,175);
  * a match to an ``ignore`` ground truth is dropped (neither tp nor fp, :177); the first match to a ground truth is a
    true positive at difficulty level 1 (``difficulty <= 1`` -> column 0) and level 2 (``<= 2`` -> column 1), a second
    match to the same box is a false positive at the SAME levels (:179-194);
  * an unmatched detection in a frame that has ground truth is a false positive at every level (:201-207); in a frame
    without any ground truth of the class it is not counted at all (the ``elif`` needs ``BBGT.size > 0``);
  * npos per level counts the non-ignored boxes of non-ignored frames (count_npos, :250-259);
  * ``rec, prec = zip(*sorted(zip(rec, prec)))`` (:243) sorts the PAIRS by recall before the area is taken;
  * ``map = mrec = mprec = np.zeros(...)`` (:232) aliases ONE array, so the three returned arrays all hold the AP
    (the last assignment wins) — kept, and the un-aliased mean recall / precision are returned as extra values.
Annotation loading (``load_recs`` / ``db._load_waymo_annotation``) is dataset parsing and out of scope: the caller
hands in the per-frame records.

Parity: unpinned (the reference module needs shapely and the missing eval_utils to import); ``ap`` and the '2d'
overlap are cross-checked against ``voc_eval`` (which IS pinned to the imported reference) on cases where the two
protocols coincide (tests/test_waymo_eval.py).
"""
import numpy as np

from ..model.config import cfg
from .voc_eval import voc_ap

UNCERTAINTY_ORDER = ('a_entropy', 'a_mutual_info', 'a_cls_var', 'e_entropy', 'e_mutual_info', 'e_cls_var',
                     'a_bbox_var', 'e_bbox_var')


def make_rec(filename, boxes, difficulty=None, ignore=None, boxes_dc=None, scene_idx=0, frame_idx=0, ignore_frame=None):
    """A per-frame, per-class ground-truth record with the fields the matching loop reads (load_recs, :270-309)."""
    boxes = np.asarray(boxes, dtype=np.float64).reshape(len(boxes), -1) if len(boxes) else np.zeros((0, 4))
    n = boxes.shape[0]
    return {'filename': filename, 'idx': filename, 'boxes': boxes,
            'boxes_dc': np.asarray(boxes_dc if boxes_dc is not None else np.zeros((0, boxes.shape[1])), dtype=np.float64),
            'difficulty': np.asarray(difficulty if difficulty is not None else np.ones(n), dtype=np.int64),
            'ignore': np.asarray(ignore if ignore is not None else np.zeros(n), dtype=bool),
            'hit': np.zeros(n, dtype=bool), 'scene_idx': scene_idx, 'frame_idx': frame_idx,
            'ignore_frame': (n == 0) if ignore_frame is None else bool(ignore_frame)}


# ---- eval_utils (missing file), restated from the call sites ---------------------------------------------------------
def find_rec(class_recs, token):
    """waymo_eval.py:141: the record of the frame a detection line names (results lines carry the frame's file name,
    lib/datasets/db.py:316-323); frames flagged ``ignore_frame`` yield no record."""
    for rec in class_recs:
        if rec.get('filename') == token and not rec.get('ignore_frame', False):
            return rec
    return None


def _poly_clip(subject, clip):
    """Sutherland-Hodgman: convex polygon ``subject`` clipped by convex ``clip`` (both (n,2), counter-clockwise)."""
    out = list(subject)
    for i in range(len(clip)):
        a, b = clip[i], clip[(i + 1) % len(clip)]
        inp, out = out, []
        if not inp:
            break
        for j in range(len(inp)):
            p, q = inp[j], inp[(j + 1) % len(inp)]
            side_p = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
            side_q = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
            if side_p >= 0:
                out.append(p)
            if (side_p >= 0) != (side_q >= 0):
                t = side_p / (side_p - side_q)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def _poly_area(poly):
    if len(poly) < 3:
        return 0.0
    x, y = np.array([p[0] for p in poly]), np.array([p[1] for p in poly])
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def _bev_corners(box):
    """[xc, yc, zc, l, w, h, ry] -> the 4 BEV corners, counter-clockwise."""
    xc, yc, l, w, ry = box[0], box[1], box[3], box[4], box[6]
    c, s = np.cos(ry), np.sin(ry)
    pts = [(l / 2, w / 2), (-l / 2, w / 2), (-l / 2, -w / 2), (l / 2, -w / 2)]
    return [(xc + c * px - s * py, yc + s * px + c * py) for px, py in pts]


def iou(bbgt, bb, eval_type='2d'):
    """Overlap of ONE detection ``bb`` with every ground-truth row.  '2d': [x1,y1,x2,y2] with the VOC +1 pixel convention
    (lib/datasets/voc_eval.py:155-169, the file this evaluator descends from); 'bev_aa': yaw-less rectangles
    xc -+ l/2, yc -+ w/2 of 7-element boxes; 'bev': rotated rectangles; '3d': rotated footprint x height overlap."""
    bbgt = np.asarray(bbgt, dtype=np.float64)
    bb = np.asarray(bb, dtype=np.float64)
    if eval_type == '2d':
        ixmin, iymin = np.maximum(bbgt[:, 0], bb[0]), np.maximum(bbgt[:, 1], bb[1])
        ixmax, iymax = np.minimum(bbgt[:, 2], bb[2]), np.minimum(bbgt[:, 3], bb[3])
        iw, ih = np.maximum(ixmax - ixmin + 1.0, 0.0), np.maximum(iymax - iymin + 1.0, 0.0)
        inters = iw * ih
        uni = ((bb[2] - bb[0] + 1.0) * (bb[3] - bb[1] + 1.0) + (bbgt[:, 2] - bbgt[:, 0] + 1.0) * (bbgt[:, 3] - bbgt[:, 1] + 1.0)
               - inters)
        return inters / uni
    if eval_type == 'bev_aa':
        def rect(b):
            return b[..., 0] - b[..., 3] / 2, b[..., 1] - b[..., 4] / 2, b[..., 0] + b[..., 3] / 2, b[..., 1] + b[..., 4] / 2
        gx1, gy1, gx2, gy2 = rect(bbgt)
        dx1, dy1, dx2, dy2 = rect(bb)
        iw = np.maximum(np.minimum(gx2, dx2) - np.maximum(gx1, dx1), 0.0)
        ih = np.maximum(np.minimum(gy2, dy2) - np.maximum(gy1, dy1), 0.0)
        inters = iw * ih
        return inters / ((gx2 - gx1) * (gy2 - gy1) + (dx2 - dx1) * (dy2 - dy1) - inters)
    if eval_type in ('bev', '3d'):
        out = np.zeros(bbgt.shape[0])
        det_poly, det_area = _bev_corners(bb), bb[3] * bb[4]
        for i, g in enumerate(bbgt):
            inter = _poly_area(_poly_clip(det_poly, _bev_corners(g)))
            if eval_type == 'bev':
                out[i] = inter / (det_area + g[3] * g[4] - inter)
            else:
                zlo, zhi = max(bb[2] - bb[5] / 2, g[2] - g[5] / 2), min(bb[2] + bb[5] / 2, g[2] + g[5] / 2)
                vol = inter * max(zhi - zlo, 0.0)
                out[i] = vol / (det_area * bb[5] + g[3] * g[4] * g[5] - vol)
        return out
    raise ValueError("eval_type %r" % (eval_type,))


def ap(rec, prec):
    """Area under the monotone precision envelope (the VOC continuous rule, lib/datasets/voc_eval.py:53-69)."""
    return voc_ap(np.asarray(rec, dtype=np.float64), np.asarray(prec, dtype=np.float64), use_07_metric=False)


def extract_uncertainties(bbox_elem, splitlines, num_classes=2):
    """Columns behind ``idx token score box...`` of each results line, split by the enabled cfg.UC.* flags in the order
    lib/model/test.py:260-270 stacked them.  Returns (per-scene accumulators, {key: (D, width) array})."""
    u = cfg.UC
    widths = []
    for key in UNCERTAINTY_ORDER:
        on = {'a_bbox_var': u.EN_BBOX_ALEATORIC, 'e_bbox_var': u.EN_BBOX_EPISTEMIC}.get(
            key, u.EN_CLS_ALEATORIC if key.startswith('a_') else u.EN_CLS_EPISTEMIC)
        if on:
            widths.append((key, bbox_elem if key.endswith('bbox_var') else (num_classes if key.endswith('cls_var') else 1)))
    uncertainties, col = {}, 3 + bbox_elem
    for key, w in widths:
        uncertainties[key] = np.array([[float(z) for z in x[col:col + w]] for x in splitlines]).reshape(len(splitlines), w)
        col += w
    uc_avg = {key: np.zeros((cfg.get('NUM_SCENES', 1), w)) for key, w in widths}
    return uc_avg, uncertainties


# ---- waymo_eval.py:44-247 -------------------------------------------------------------------------------------------------
def waymo_eval(detfile, class_recs, ovthresh=0.5, eval_type='2d', d_levels=2, bbox_elem=None, ignore_dc=None):
    """Returns (mrec, mprec, map) exactly as the reference does (three views of ONE array holding the AP per difficulty
    level, see the module docstring) plus a dict with the un-aliased quantities: {'ap', 'mean_recall', 'mean_precision',
    'tp', 'fp', 'npos'}."""
    ovthresh_dc = 0.5
    bbox_elem = bbox_elem if bbox_elem is not None else cfg[cfg.NET_TYPE.upper()].NUM_BBOX_ELEM
    ignore_dc = bool(cfg.TEST.get('IGNORE_DC', False)) if ignore_dc is None else ignore_dc
    with open(detfile, 'r') as f:
        splitlines = [x.strip().split(' ') for x in f.readlines() if x.strip()]
    frame_tokens = [x[1] for x in splitlines]
    confidence = np.array([float(x[2]) for x in splitlines])
    bb_all = np.array([[float(z) for z in x[3:3 + bbox_elem]] for x in splitlines]).reshape(len(splitlines), bbox_elem)
    n = len(splitlines)
    tp, fp = np.zeros((n, d_levels)), np.zeros((n, d_levels))
    npos = np.zeros((len(class_recs), d_levels))
    for i, rec in enumerate(class_recs):                                   # count_npos, :250-259
        if rec.get('ignore_frame') is False:
            for j, ig in enumerate(rec['ignore']):
                if not ig:
                    for lvl in range(d_levels):
                        if rec["difficulty"][j] <= lvl + 1:
                            npos[i, lvl] += 1
    for rec in class_recs:
        if 'hit' in rec:
            rec['hit'][:] = False
    idx = 0
    if bb_all.shape[0] > 0:
        for det_idx in np.argsort(-confidence):
            rec = find_rec(class_recs, frame_tokens[det_idx])
            if rec is None:
                continue
            bb = bb_all[det_idx, :].astype(float)
            ovmax, jmax = -np.inf, 0
            bbgt, bbgt_dc = rec['boxes'].astype(float), rec['boxes_dc'].astype(float)
            ovmax_dc = 0
            if bbgt_dc.size > 0 and ignore_dc:
                ovmax_dc = np.max(iou(bbgt_dc, bb, eval_type))
            if bbgt.size > 0:
                overlaps = iou(bbgt, bb, eval_type)
                ovmax, jmax = np.max(overlaps), int(np.argmax(overlaps))
            if ovmax > ovthresh and ovmax_dc < ovthresh_dc:
                if not rec['ignore'][jmax]:
                    table = fp if rec['hit'][jmax] else tp
                    for lvl in range(d_levels):
                        if rec['difficulty'][jmax] <= lvl + 1:
                            table[idx, lvl] += 1
                    rec['hit'][jmax] = True
            elif bbgt.size > 0 and ovmax_dc < ovthresh_dc:
                fp[idx, :] += 1
            idx += 1
    shared = np.zeros((d_levels,))                                          # map = mrec = mprec (:232)
    plain = {'ap': np.zeros(d_levels), 'mean_recall': np.zeros(d_levels), 'mean_precision': np.zeros(d_levels),
             'tp': tp, 'fp': fp, 'npos': npos}
    fp_sum, tp_sum, npos_sum = np.cumsum(fp, axis=0), np.cumsum(tp, axis=0), np.sum(npos, axis=0)
    for i in range(d_levels):
        npos_d = npos_sum[i] if npos_sum[i] != 0 else 1.0
        rec_c = tp_sum[:, i] / float(npos_d)
        prec_c = tp_sum[:, i] / np.maximum(tp_sum[:, i] + fp_sum[:, i], np.finfo(np.float64).eps)
        if len(rec_c):
            rec_c, prec_c = zip(*sorted(zip(rec_c, prec_c)))
        plain['mean_precision'][i] = np.average(prec_c) if len(prec_c) else 0.0
        plain['mean_recall'][i] = np.average(rec_c) if len(rec_c) else 0.0
        plain['ap'][i] = ap(rec_c, prec_c)
        shared[i] = plain['ap'][i]
    return shared, shared, shared, plain
