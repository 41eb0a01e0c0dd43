"""Scoring side of the eval loop: the detections text format the reference writes per class
(lib/datasets/db.py:305-367) and the PASCAL-VOC precision/recall/AP it is scored with (lib/datasets/voc_eval.py:38-219).
Host-side numpy, exactly where the reference runs it: the detections arrive on the host once per frame
(``model/test.detect_frame_device``), everything after that is bookkeeping on a few hundred boxes.

Dataset parsing (xml annotations, Waymo/KITTI/CADC label readers) stays with the caller: ``voc_eval_arrays`` takes the
ground truth as ``{frame_id: {'bbox': (G,4), 'difficult': (G,) bool}}``.
"""
import numpy as np


def voc_ap(rec, prec, use_07_metric=False):
    """lib/datasets/voc_eval.py:38-69."""
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def voc_eval_arrays(image_ids, confidence, boxes, class_recs, ovthresh=0.5, use_07_metric=False, use_diff=False):
    """The matching loop of voc_eval (lib/datasets/voc_eval.py:135-219) on in-memory detections.
    image_ids: list of frame ids (one per detection), confidence (D,), boxes (D,4) [x1,y1,x2,y2] (+1 area convention).
    Returns (rec, prec, ap)."""
    recs, npos = {}, 0
    for key, r in class_recs.items():
        bbox = np.asarray(r['bbox'], dtype=float).reshape(-1, 4)
        difficult = (np.zeros(len(bbox), dtype=bool) if use_diff
                     else np.asarray(r.get('difficult', np.zeros(len(bbox))), dtype=bool))
        npos += int(np.sum(~difficult))
        recs[key] = {'bbox': bbox, 'difficult': difficult, 'det': [False] * len(bbox)}
    confidence = np.asarray(confidence, dtype=float)
    BB = np.asarray(boxes, dtype=float).reshape(-1, 4) if len(confidence) else np.zeros((0, 4))
    nd = len(image_ids)
    tp, fp = np.zeros(nd), np.zeros(nd)
    if BB.shape[0] > 0:
        sorted_ind = np.argsort(-confidence)
        BB = BB[sorted_ind, :]
        image_ids = [image_ids[x] for x in sorted_ind]
        for d in range(nd):
            R = recs[image_ids[d]]
            bb = BB[d, :]
            ovmax, jmax = -np.inf, -1
            BBGT = R['bbox']
            if BBGT.size > 0:
                iw = np.maximum(np.minimum(BBGT[:, 2], bb[2]) - np.maximum(BBGT[:, 0], bb[0]) + 1., 0.)
                ih = np.maximum(np.minimum(BBGT[:, 3], bb[3]) - np.maximum(BBGT[:, 1], bb[1]) + 1., 0.)
                inters = iw * ih
                uni = ((bb[2] - bb[0] + 1.) * (bb[3] - bb[1] + 1.) +
                       (BBGT[:, 2] - BBGT[:, 0] + 1.) * (BBGT[:, 3] - BBGT[:, 1] + 1.) - inters)
                overlaps = inters / uni
                ovmax, jmax = np.max(overlaps), int(np.argmax(overlaps))
            if ovmax > ovthresh:
                if not R['difficult'][jmax]:
                    if not R['det'][jmax]:
                        tp[d] = 1.
                        R['det'][jmax] = True
                    else:
                        fp[d] = 1.
            else:
                fp[d] = 1.
    fp, tp = np.cumsum(fp), np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def format_image_rows(im_ind, name, dets):
    """The text lines of one frame's detections of one class: ``idx token score x1 y1 x2 y2 [uc ...]``
    (lib/datasets/db.py:305-334)."""
    dets = np.asarray(dets)
    if dets.size == 0:
        return ''
    name = str(name)
    out = []
    for row in dets.tolist():
        line = '{:d} {:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}'.format(im_ind, name, row[4], row[0], row[1], row[2], row[3])
        if len(row) > 5:
            line += ''.join(' {:.10f}'.format(v) for v in row[5:])
        out.append(line)
    return '\n'.join(out) + '\n'


def format_lidar_rows(ind, name, dets):
    """LiDAR rows (n, 8+u) [xc,yc,zc,l,w,h,ry,score,...] -> ``idx token score xc yc zc l w h ry [uc ...]``
    (lib/datasets/db.py:336-367)."""
    dets = np.asarray(dets)
    if dets.size == 0:
        return ''
    name = str(name)
    out = []
    for row in dets.tolist():
        line = '{:d} {:s} {:.3f} {:.3f} {:.3f} {:.3f} {:.3f} {:.3f} {:.3f} {:.5f}'.format(
            ind, name, row[7], row[0], row[1], row[2], row[3], row[4], row[5], row[6])
        if len(row) > 8:
            line += ''.join(' {:.10f}'.format(v) for v in row[8:])
        out.append(line)
    return '\n'.join(out) + '\n'


def write_image_results_file(dets_per_frame, frame_names, filename):
    """One class of ``all_boxes`` (list over frames of (n, 5+u) [x1,y1,x2,y2,score,uncertainties...]) in the
    reference's text format (lib/datasets/db.py:305-334)."""
    with open(filename, 'wt') as f:
        for im_ind, img in enumerate(frame_names):
            f.write(format_image_rows(im_ind, img, dets_per_frame[im_ind]))


def write_lidar_results_file(dets_per_frame, frame_names, filename):
    """LiDAR counterpart (lib/datasets/db.py:336-367)."""
    with open(filename, 'wt') as f:
        for ind, frame in enumerate(frame_names):
            f.write(format_lidar_rows(ind, frame, dets_per_frame[ind]))


def read_results_file(filename, num_box_values=4):
    """Inverse of the writers: returns (frame indices, tokens, scores, boxes (D, num_box_values), extras (D, u))."""
    idx, tok, score, box, extra = [], [], [], [], []
    with open(filename, 'rt') as f:
        for line in f:
            parts = line.strip().split(' ')
            if len(parts) < 3 + num_box_values:
                continue
            idx.append(int(parts[0]))
            tok.append(parts[1])
            score.append(float(parts[2]))
            box.append([float(v) for v in parts[3:3 + num_box_values]])
            extra.append([float(v) for v in parts[3 + num_box_values:]])
    return (np.asarray(idx, dtype=np.int64), tok, np.asarray(score), np.asarray(box).reshape(-1, num_box_values),
            np.asarray(extra).reshape(len(idx), -1) if idx else np.zeros((0, 0)))
