"""Per-frame detection loop body — counterpart of lib/model/test.py:68-116 (frame_detect) and :210-228
(the ``max_dets`` cut that test_net applies per class), for the image detector.

``frame_detect`` keeps the reference's call sequence (test_frame -> filter_and_draw_prep).
``detect_frame_device`` is the throughput form used by bench.py and the 8-GPU eval collate: everything up to
the per-class, max_dets-limited detections stays on the device and a frame ends with ONE device->host
copy of a fixed-size record.  ``test_net`` is the eval loop of :138-257 over a caller-supplied frame source,
sharded one frame per rank per step.
"""
import numpy as np
import torch

from ..model.config import cfg, get_output_dir
from ..roi_data_layer import minibatch
from ..utils.filter_predictions import filter_and_draw_prep, filter_device


def _get_blobs(filename):
    """lib/model/test.py:32-44: ``filename`` is a one-element list; returns {'data': (1,H,W,C) device blob, 'info'}."""
    blobs = {}
    scale = cfg.TEST.SCALES[0]
    if cfg.NET_TYPE == 'image':
        infos, blobs['data'], _ = minibatch._get_image_blob(filename, scale, augment_en=cfg.TEST.AUGMENT_EN, mode='test')
        blobs['info'] = infos[0]
    elif cfg.NET_TYPE == 'lidar':
        infos, blobs['data'], _ = minibatch._get_lidar_blob(filename, lidar_extents(), scale, augment_en=False, mode='test')
        blobs['info'] = infos[0]
    return blobs


class ReferenceDb:
    """Adaptor from the reference's dataset protocol to the frame source ``test_net`` iterates.

    The reference's loop (lib/model/test.py:141-147,183-206) reads ``db._val_index`` / ``db._test_index`` for the frame
    count, ``db.path_at(i, mode)`` for the file of frame i (lib/datasets/db.py:139-148), loads it with ``_get_blobs``
    and finally calls ``db.evaluate_detections(all_boxes, output_dir, 'val')`` (:256-257).  Any object with those members
    (the reference's ``waymo_imdb`` / ``kitti_lidb`` / ... classes; dataset classes themselves are outside this package)
    can be passed to ``test_net`` directly - it is wrapped in this class."""

    def __init__(self, db, get_blobs=None):
        self.inner = db
        self._get_blobs = get_blobs or _get_blobs

    @property
    def num_classes(self):
        return self.inner.num_classes

    @property
    def name(self):
        return self.inner.name

    def _index(self, mode):
        return self.inner._test_index if mode == 'test' else self.inner._val_index if mode == 'val' else []

    def num_frames(self, mode):
        return len(self._index(mode))

    def blobs_at(self, i, mode):
        return self._get_blobs([self.inner.path_at(i, mode)])

    def name_at(self, i, mode):
        return str(self._index(mode)[i])

    def evaluate_detections(self, all_boxes, out_dir, mode):
        return self.inner.evaluate_detections(all_boxes, out_dir, mode)

    @staticmethod
    def wraps(db):
        return not hasattr(db, 'blobs_at') and (hasattr(db, '_val_index') or hasattr(db, '_test_index'))


def frame_detect(net, blobs, num_classes, thresh):
    """lib/model/test.py:68-93: the epistemic heads run cfg.UC.E_NUM_SAMPLE stochastic passes per frame."""
    epistemic = cfg.UC.EN_BBOX_EPISTEMIC or cfg.UC.EN_CLS_EPISTEMIC
    if epistemic:
        net.set_e_num_sample(cfg.UC.E_NUM_SAMPLE)
    try:
        _, probs, bbox_pred, rois, uncertainties = net.test_frame(blobs['data'], blobs['info'])
    finally:
        if epistemic:
            net.set_e_num_sample(1)
    return filter_and_draw_prep(rois, probs, bbox_pred, uncertainties, blobs['info'], num_classes, thresh,
                                cfg.NET_TYPE)


def stack_uncertainties(cls_bbox, cls_uncertainties, num_uc_pos):
    """lib/model/test.py:260-270: detection rows followed by their uncertainty columns, in dict order."""
    out = np.zeros((cls_bbox.shape[0], cls_bbox.shape[1] + num_uc_pos))
    out[:, 0:cls_bbox.shape[1]] = cls_bbox
    ptr = cls_bbox.shape[1]
    for _, uncert in cls_uncertainties.items():
        end = ptr + uncert.shape[1]
        out[:, ptr:end] = uncert[:, :]
        ptr = end
    return out


def apply_max_dets(cls_boxes, max_dets):
    """test.py:213-221: keep every detection scoring >= the max_dets-th best (ties stay)."""
    if max_dets > 0 and len(cls_boxes) > max_dets:
        cut = np.sort(cls_boxes[:, -1])[-max_dets]
        cls_boxes = cls_boxes[np.where(cls_boxes[:, -1] >= cut)[0], :]
    return cls_boxes


def detect_frame_device(net, data, info, thresh=0.5, max_dets=100, max_out=None):
    """One frame, asynchronous: returns (dets (K, max_out, 5), det_count (K,)) device tensors holding,
    per class, the detections test_net would store in all_boxes[cls][frame]."""
    epistemic = cfg.UC.EN_BBOX_EPISTEMIC or cfg.UC.EN_CLS_EPISTEMIC
    if epistemic:
        net.set_e_num_sample(cfg.UC.E_NUM_SAMPLE)          # lib/model/test.py:74-77
    try:
        with torch.no_grad():
            net.forward(data, info, None, None, mode='TEST')
    finally:
        if epistemic:
            net.set_e_num_sample(1)
    p = net._predictions
    max_out = max_out if max_out is not None else p['cls_prob'].shape[0]
    return filter_device(p['rois_count'], p['cls_prob'], p['pred_boxes'], info, thresh, max_dets, max_out,
                         db_type=cfg.NET_TYPE, uncertainties=p.get('uncertainties'))


def lidar_extents():
    """[x1,y1,z1,x2,y2,z2] of the LiDAR scan in metres (lib/datasets/db.py passes cfg.LIDAR.*_RANGE this way)."""
    return [cfg.LIDAR.X_RANGE[0], cfg.LIDAR.Y_RANGE[0], cfg.LIDAR.Z_RANGE[0],
            cfg.LIDAR.X_RANGE[1], cfg.LIDAR.Y_RANGE[1], cfg.LIDAR.Z_RANGE[1]]


def bbox_voxel_grid_to_pc(bboxes, bev_extents, info):
    """lib/utils/bbox.py:140-162 (called at lib/model/test.py:223-224 on the host copy of the detections):
    voxel-grid [xc,yc,zc,l,w,h,ry,...] rows -> metres, in place on the numpy array."""
    scale = info[6]
    s_info = np.asarray(info[0:6]) * 1 / scale
    kx = (bev_extents[3] - bev_extents[0]) / (s_info[1] - s_info[0])
    ky = (bev_extents[4] - bev_extents[1]) / (s_info[3] - s_info[2])
    bboxes[:, 0] = bboxes[:, 0] * kx + bev_extents[0]
    bboxes[:, 1] = bboxes[:, 1] * ky + bev_extents[1]
    bboxes[:, 3] = bboxes[:, 3] * kx
    bboxes[:, 4] = bboxes[:, 4] * ky
    return bboxes


def test_net(net, db, out_dir, max_dets=100, thresh=0.1, mode='test', draw_det=False, eval_det=False, timers=None):
    """Eval loop of lib/model/test.py:138-257 over a frame source, sharded one frame per rank per step when
    torch.distributed is initialised (BASELINE.json configs[4]).

    Execution (cfg.TEST.FRAME_GRAPHS, default on): frame s is replayed as a captured hipGraph on HIP stream
    s % cfg.TEST.FRAMES_IN_FLIGHT (``model/frame_graph.FramePool`` attached to the net: one graph per stream and frame
    problem, captured at first use, eager launches for frame sizes it has not captured) - the arrangement bench.py times.
    Records are bit-equal to ``detect_frame_device`` on the same blob (tests/test_reference_names.py).
    ``timers``: optional dict that receives 'loop_s' (first frame queued -> last record on the host), 'frames' and
    'pool' (replays / eager frames / captures) - the counterpart of the reference's _t timers (:171,196-250).

    ``db``: either an object with the reference's dataset protocol (``_val_index`` / ``_test_index``, ``path_at``,
    ``num_classes``, ``name``, ``evaluate_detections``; wrapped in ``ReferenceDb``, frames loaded by ``_get_blobs``), or a
    frame source with ``num_classes``, ``num_frames(mode)``, ``blobs_at(i, mode)`` -> {'data': (1,H,W,C) blob or None,
    'info': 7-vector}, optional ``name_at(i, mode)``, optional ``evaluate_detections(all_boxes, out_dir, mode)``.
    ``out_dir`` None / '' -> ``get_output_dir(db, mode='test')`` like the reference (:166), which ignores the argument.  Per frame everything stays on the device up to the
    per-class, max_dets-limited record (``detect_frame_device``); records are collated in blocks of
    ``collate.EVAL_GATHER_EVERY`` frames (one all-gather over the ranks + one device-to-host copy per block, on a stream of
    their own: ``collate.RecordRing``), so every rank ends with the complete ``all_boxes`` and no frame waits for the host.
    LiDAR detections are converted from the voxel grid to metres (:223-224).  Writes ``detections.pkl`` like the
    reference (:246-248) plus the per-class text files of lib/datasets/db.py:305-367 (rank 0 only) and returns
    ``all_boxes[cls][frame]`` (rows [box..., score])."""
    import contextlib
    import os
    import pickle
    import torch.distributed as dist
    from ..datasets import voc_eval
    from . import collate
    if draw_det:
        raise NotImplementedError("drawing is dataset tooling, outside the accelerated path")
    np.random.seed(cfg.RNG_SEED)
    if ReferenceDb.wraps(db):
        db = ReferenceDb(db)
    if not out_dir:
        out_dir = get_output_dir(db, mode='test')
    num_images, k = db.num_frames(mode), db.num_classes
    lidar = cfg.NET_TYPE == 'lidar'
    from ..nets.uncertainty import num_uncertainty_pos
    # rows carry the uncertainty columns of cfg.UC.* behind the box and the score (lib/model/test.py:151-159,222-226)
    elem = (8 if lidar else 5) + num_uncertainty_pos(k, 7 if lidar else 4)
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if distributed else 0
    world = dist.get_world_size() if distributed else 1
    all_boxes = [[np.empty(0) for _ in range(num_images)] for _ in range(k)]
    mine = collate.shard_frames(num_images, rank, world)
    steps = (num_images + world - 1) // world
    dev = torch.device(net._device)
    # rows per class in the exchanged record: the max_dets cut keeps every detection that TIES with the max_dets-th best
    # score (lib/model/test.py:213-221), so a record of max_dets rows could truncate; one row per RoI cannot
    max_out = max(max_dets, int(cfg.TEST.RPN_POST_NMS_TOP_N)) if max_dets > 0 else int(cfg.TEST.RPN_POST_NMS_TOP_N)
    infos = {}
    numel = collate.record_numel(k, max_out, elem)
    gather_dev = dev if (not distributed or dist.get_backend() == 'nccl') else torch.device('cpu')
    # Frames are queued without a host round trip each: a frame's record is packed into a device ring and a block of
    # EVAL_GATHER_EVERY frames is collated (one all-gather over the ranks) and copied to the host on the ring's own stream
    # (collate.RecordRing); the host unpacks a chunk of blocks at a time.
    chunk = collate.EVAL_GATHER_EVERY * 8
    # (an injected RPN output - the evaluation hook Network._rpn_override - is a per-call host decision: eager path)
    pool = net.frame_pool() if (cfg.TEST.FRAME_GRAPHS and dev.type == 'cuda' and hasattr(net, 'frame_pool')
                                and getattr(net, '_rpn_override', None) is None) else None
    lanes = pool.n_streams if pool is not None else 1
    if pool is not None:
        pool.sync_weights()
        cur = torch.cuda.current_stream(dev)
        for st in pool.streams:
            st.wait_stream(cur)
    import time
    t_loop = time.perf_counter()
    # rank 0 streams the per-class text files of lib/datasets/db.py:305-367 while the device works on the next chunk
    writers = None
    fmt = voc_eval.format_lidar_rows if lidar else voc_eval.format_image_rows
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        writers = {j: open(os.path.join(out_dir, 'det_%s_cls%d.txt' % (mode, j)), 'wt') for j in range(1, k)}

    def queue_chunk(c0, n, ring):
        for s in range(c0, c0 + n):
            # the loader runs inside the lane's stream context: device-side producers (prep_im_for_blob, the BEV voxeliser)
            # launch on the stream that consumes their blob
            with (torch.cuda.stream(pool.stream(s)) if pool is not None else contextlib.nullcontext()):
                blobs = db.blobs_at(mine[s], mode) if s < len(mine) else None
                slot = ring.slot(s - c0)
                if blobs is not None and blobs.get('data') is not None:
                    infos[mine[s]] = blobs['info']
                    runner = (pool.runner(blobs['data'].shape, blobs['info'], thresh, max_dets, max_out, lane=s % lanes)
                              if pool is not None else None)
                    if runner is not None:
                        dets, counts = runner.run(blobs['data'])
                    else:
                        dets, counts = detect_frame_device(net, blobs['data'], blobs['info'], thresh, max_dets, max_out)
                    collate.pack_record(dets, counts, slot)
                else:
                    slot.zero_()          # a frame without data (the reference skips it, :198-203) or a padding step
                ring.commit(s - c0)

    def collect_chunk(c0, n, ring):
        host = ring.drain()
        for s in range(c0, c0 + n):
            rows = collate.unpack_records(host[s - c0], k, max_out, elem)
            if distributed:
                frames_of_step = [(r, s * world + r) for r in range(world) if s * world + r < num_images]
            else:
                frames_of_step = [(0, mine[s])] if s < len(mine) else []
            for r, i in frames_of_step:
                info = None
                if lidar:
                    # the voxel-grid geometry depends on cfg and the frame scale only: another rank's frame needs no reload
                    info = infos[i] if i in infos else minibatch.lidar_frame_geometry(cfg.TEST.SCALES[0])[2]
                name = (db.name_at(i, mode) if hasattr(db, 'name_at') else '%06d' % i) if writers is not None else None
                for j in range(1, k):
                    cls_boxes = rows[r][j]
                    if lidar and cls_boxes.size:
                        cls_boxes = bbox_voxel_grid_to_pc(cls_boxes, lidar_extents(), info)
                    all_boxes[j][i] = cls_boxes if cls_boxes.size else np.empty(0)
                    if writers is not None and cls_boxes.size:
                        writers[j].write(fmt(i, name, cls_boxes))
        ring.reset()

    # Two rings alternate: chunk c+1 is queued on the device before chunk c's records are waited for, unpacked into
    # all_boxes and written out, so the host-side work of a chunk overlaps the device work of the next one (frames are
    # visited in order, so the text files are written in frame order like the reference's)
    rings = [collate.RecordRing(numel, min(chunk, max(steps, 1)), every=collate.EVAL_GATHER_EVERY, device=dev,
                                distributed=distributed, gather_device=gather_dev) for _ in range(2 if steps > chunk else 1)]
    prev = None
    for ci, c0 in enumerate(range(0, steps, chunk)):
        n = min(chunk, steps - c0)
        ring = rings[ci % len(rings)]
        queue_chunk(c0, n, ring)
        if prev is not None:
            collect_chunk(*prev)
        prev = (c0, n, ring)
    if prev is not None:
        collect_chunk(*prev)
    if pool is not None:
        cur = torch.cuda.current_stream(dev)
        for st in pool.streams:
            cur.wait_stream(st)
    if timers is not None:
        timers['loop_s'] = time.perf_counter() - t_loop
        timers['frames'] = len(mine)
        timers['pool'] = dict(pool.stats) if pool is not None else None
    if rank == 0:
        for f in writers.values():
            f.close()
        with open(os.path.join(out_dir, 'detections.pkl'), 'wb') as f:
            pickle.dump(all_boxes, f, pickle.HIGHEST_PROTOCOL)
        if eval_det and hasattr(db, 'evaluate_detections'):
            db.evaluate_detections(all_boxes, out_dir, mode)
    return all_boxes
