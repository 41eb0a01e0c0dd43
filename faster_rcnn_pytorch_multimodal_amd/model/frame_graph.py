"""One frame of the detector as a replayable hipGraph, and the pool of such graphs behind the drop-in API.

The reference runs its per-frame loop (lib/model/test.py:183-228) eagerly: ~115 kernel launches from
Python plus one host copy per class.  Here the whole frame — channel pad, ResNet-101 head, RPN, proposal
sort/NMS, RoIAlign, layer4 on the RoIs, detection tail, per-class filter — is captured ONCE into a
hipGraph (every kernel of libfrcnn_hip.so is asynchronous, allocation-free and keeps data-dependent
counts on the device) and replayed per frame, so the launch-bound small layers (layer3's 69 convs of
~10 us each) are not paced by the host.

Several runners may share one ``net`` and replay concurrently on different HIP streams (4 frames in
flight): everything a frame WRITES — activations, workspaces, proposal buffers, the detection
record — is allocated during the capture and therefore private to the runner's graph; what runners share
(filters, folded BatchNorm scale/shift, anchors, tuned plans) is read-only after the eager warm-up.
``tests/test_timed_path.py`` replays distinct frames through that arrangement and compares every record
with the eager path bit for bit and with the CPU oracle.

``FramePool`` is what the product's callers use: ``model.test.test_net`` (lib/model/test.py:138-257), ``Network.test_frame``
(lib/model/test.py:75) and ``Network.run_eval`` (lib/model/train_val.py:411-412) ask it for the runner of a frame's
problem (shape, info, thresholds, cfg fingerprint); it captures on demand, keeps ``streams`` runners per problem (one per
HIP stream), notices changed weights, and hands back None (-> the eager path) for shapes it has not decided to capture.
"""
import contextlib
import gc

import numpy as np
import torch

from .config import cfg
from .test import detect_frame_device


@contextlib.contextmanager
def capture(graph, **kwargs):
    """``torch.cuda.graph(graph, **kwargs)`` with Python's cyclic collector held off until the capture has ended.

    A capture runs in HIP's global capture mode: a call the runtime counts as unsafe (destroying another graph, releasing
    its memory pool, ...) made from ANY thread while it lasts is an error, and the destructors such calls sit in treat
    an error as fatal.  Reference cycles that hold a captured graph exist (a net and its cached TrainStepRunner point at
    each other), and the collector may wake in any thread that allocates Python objects - e.g. autograd's backward thread
    in the middle of a captured training step, which is where a round-5 GPU test run aborted.  So: collect first, keep
    the collector off while capturing, and let whatever became garbage meanwhile be freed after the capture."""
    was_enabled = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        with torch.cuda.graph(graph, **kwargs):
            yield
    finally:
        if was_enabled:
            gc.enable()


def _as_i32(seed):
    seed = int(seed) & 0xFFFFFFFF
    return seed - (1 << 32) if seed >= (1 << 31) else seed


class FrameRunner:
    """Fixed-shape frame pipeline: ``run(frame)`` -> (dets (K, max_out, 5 + U), det_count (K,)) device tensors
    that are overwritten by the next ``run``.

    ``rpn_override_shape`` = (1, H/16, W/16, ld): the runner owns a static buffer of that shape which replaces the
    RPN head's output [bg logits | fg logits | deltas | pad] (the evaluation hook ``Network._rpn_override``, used by
    the parity tests and bench.py's mAP leg to make the proposal stage well-conditioned); ``run(frame, rpn=...)``
    fills it.
    ``max_out``: rows per class of the record (default ``max_dets``; ``test_net`` passes one row per RoI because the
    ``max_dets`` cut keeps ties, lib/model/test.py:213-221).
    ``with_filter=False``: capture ``Network.forward(mode='TEST')`` only (``Network.test_frame``: the caller runs
    ``filter_and_draw_prep`` itself, lib/model/test.py:75-93); ``run`` then returns None and the frame's tensors are in
    ``predictions``.
    cfg.UC.* (uncertainty heads): the seed of the counter-based draws reaches the captured launches through a device
    word (``Network.uc_seed_args``) that ``run`` rewrites with ``net.next_uc_seed()`` - the value the eager call of the
    same forward would have passed as a scalar, so replayed and eager frames draw identical masks."""

    def __init__(self, net, height, width, channels, info, thresh=0.5, max_dets=100, use_graph=True, warmup=2,
                 autotune=True, rpn_override_shape=None, max_out=None, with_filter=True, e_num_sample=None):
        from ..nets import uncertainty
        self.net = net
        self.info = np.asarray(info, dtype=np.float32)
        self.thresh, self.max_dets = thresh, max_dets
        self.max_out = int(max_out) if max_out is not None else max_dets
        self.with_filter = bool(with_filter)
        self.e_num_sample = e_num_sample          # forward-only runners: the caller's set_e_num_sample value at capture
        dev = torch.device(net._device)
        self.dev = dev
        self.static_in = torch.zeros((1, height, width, channels), dtype=torch.float32, device=dev)
        self.static_rpn = (torch.zeros(tuple(rpn_override_shape), dtype=torch.float32, device=dev)
                           if rpn_override_shape is not None else None)
        self.seed_dev = torch.zeros((1,), dtype=torch.int32, device=dev) if uncertainty.enabled() else None
        self.graph = None
        self.out = None
        self.predictions = None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        from .. import ops
        with torch.cuda.stream(side):
            # lazy work happens here: weight layout, anchors, LDS attributes and — with autotune — the timing of the
            # (tile, split-K) candidates of every convolution shape (the plan cache is process-wide)
            ops.set_conv_autotune(autotune)
            try:
                for _ in range(max(warmup, 1)):
                    self.out = self._frame()
            finally:
                torch.cuda.synchronize(dev)
                ops.set_conv_autotune(False)
            if autotune:
                self.out = self._frame()   # one more eager frame with the tuned plans (allocator warm-up)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        if use_graph:
            self.graph = torch.cuda.CUDAGraph()
            with capture(self.graph):
                self.out = self._frame()

    def _frame(self):
        net = self.net
        net._rpn_override = self.static_rpn
        net._uc_seed_dev = self.seed_dev
        try:
            if self.with_filter:
                out = detect_frame_device(net, self.static_in, self.info, self.thresh, self.max_dets, self.max_out)
            else:
                with torch.no_grad():
                    net.forward(self.static_in, self.info, None, None, mode='TEST')
                out = None
        finally:
            net._rpn_override = None
            net._uc_seed_dev = None
        self.predictions = net._predictions     # this runner's (graph-private) intermediate tensors
        return out

    def run(self, frame, rpn=None, poison=False):
        """frame: (1,H,W,C) float32 tensor - device, or (pinned) host memory: the copy into the graph's input buffer is then
        the asynchronous host->device upload on the current stream - or a numpy blob.
        ``poison``: overwrite the graph's output buffers (NaN detections, -1 counts) before the replay, so that a replay
        which did not execute cannot leave a plausible record behind (bench.py's verification)."""
        if isinstance(frame, np.ndarray):
            frame = torch.from_numpy(np.ascontiguousarray(frame, dtype=np.float32))
        self.static_in.copy_(frame, non_blocking=True)
        if self.static_rpn is not None:
            if rpn is None:
                raise ValueError("this runner was built with an RPN override buffer: run(frame, rpn=...)")
            self.static_rpn.copy_(rpn, non_blocking=True)
        if self.seed_dev is not None:
            self.seed_dev.fill_(_as_i32(self.net.next_uc_seed()))     # a fill launch: stream-ordered, no host copy
        if self.graph is not None:
            if poison and self.out is not None:
                self.out[0].fill_(float('nan'))
                self.out[1].fill_(-1)
            self.graph.replay()
        else:
            self.out = self._frame()
        return self.out


# ---------------------------------------------------------------------------------------------------------------------
# the pool behind test_net / test_frame / run_eval
# ---------------------------------------------------------------------------------------------------------------------
def cfg_fingerprint(net):
    """Everything outside the frame's shape that a captured frame bakes in: proposal / NMS / pooling settings, the
    uncertainty flags and sample counts, the process-wide kernel switches (the Python-level ones by value, the library's -
    forced tile, convolution algorithm / staging, RoIAlign and filter variants, memops mode, NMS tie rule - through
    ``frcnn_settings_signature``, a hash of their current values), the modules' train / eval state."""
    from .. import _hip, ops
    from ..nets import network as N
    t, u = cfg.TEST, cfg.UC
    uc = tuple(bool(u.get(k, False)) for k in ('EN_BBOX_ALEATORIC', 'EN_CLS_ALEATORIC', 'EN_BBOX_EPISTEMIC', 'EN_CLS_EPISTEMIC',
                                               'EN_BBOX_EPISTEMIC_INV_TRANSFORM'))
    modes = tuple(m.training for m in net.modules())
    return (cfg.NET_TYPE, int(t.RPN_PRE_NMS_TOP_N), int(t.RPN_POST_NMS_TOP_N), float(t.RPN_NMS_THRESH), float(t.NMS_THRESH),
            str(t.get('MODE', 'nms')), int(t.get('RPN_TOP_N', 0)), str(cfg.POOLING_MODE), int(cfg.POOLING_SIZE),
            bool(cfg.ENABLE_CUSTOM_TAIL), uc, int(u.E_NUM_SAMPLE), int(u.A_NUM_CE_SAMPLE), ops.nms_suppress_at_equal(),
            ops._CONV_ALGO_MODE, ops._CONV_ALGO_FLAGS, bool(N.PROJECT_BEFORE_POOLING), bool(N.FUSE_PROJECTIONS),
            int(N.ROI_ALIGN_SAMPLING_RATIO), int(_hip.load().frcnn_settings_signature()), hash(modes))


class FramePool:
    """Captured frames of ONE net, keyed by problem, ``streams`` runners (one per HIP stream) per key.

    ``runner(...)`` returns the runner of (key, lane) - capturing it when the key has been seen often enough - or None: the
    caller then runs the eager path for that frame (``detect_frame_device`` / ``Network.forward``), on the lane's stream.
    Capture policy: the first problem is captured at once (a dataset of one frame size, the common case: every frame is a
    replay); a further problem after ``capture_after`` sightings (a one-off odd size is not worth ~1 s of warm-up + capture
    and ~1.3 GB of private activations per lane); at most ``max_keys`` problems are held, the least recently used one is
    dropped for a newcomer.
    Weights: a graph reads filters, folded BatchNorm terms and the heads' parameters BY ADDRESS.  ``sync_weights`` (called
    once per ``test_net`` / ``test_frame``) compares the version counters of every parameter and buffer with those at
    the last call: values changed in place (optimizer step, ``load_state_dict``) -> the derived tensors are re-derived in
    place (``refresh_derived_weights``), the graphs stay valid; storage replaced (``net.to``, a re-created parameter) ->
    every graph is dropped and re-captured on demand."""

    def __init__(self, net, streams=4, max_keys=4, capture_after=2, autotune=True, warmup=2, use_graph=True):
        self.net = net
        self.dev = torch.device(net._device)
        self.n_streams = max(1, int(streams))
        self.max_keys = max(1, int(max_keys))
        self.capture_after = max(0, int(capture_after))
        self.autotune, self.warmup, self.use_graph = bool(autotune), int(warmup), bool(use_graph)
        from .streams import concurrent_streams      # streams that overlap by measurement, not by hope (model/streams.py)
        self.streams, self.distinct_queues = concurrent_streams(self.n_streams, self.dev)
        self.runners = {}             # key -> [runner or None] * streams
        self.sightings = {}           # key -> frames seen
        self.uncapturable = set()     # problems whose capture failed: eager from then on
        self.clock = 0
        self.last_use = {}
        self.stats = {'replays': 0, 'eager': 0, 'captures': 0, 'refreshes': 0, 'invalidations': 0}
        self._stamp = self._weights_stamp()

    # ---- weights -----------------------------------------------------------------------------------------------------
    def _weights_stamp(self):
        vers, ptrs = [], []
        for t in list(self.net.parameters()) + list(self.net.buffers()):
            vers.append(t._version)
            ptrs.append(t.data_ptr())
        stats = sum(m.__dict__.get('_frcnn_stats_version', 0) for m in self.net.modules())
        return tuple(vers), tuple(ptrs), stats

    def sync_weights(self):
        stamp = self._weights_stamp()
        if stamp == self._stamp:
            return
        old, self._stamp = self._stamp, stamp
        if stamp[1] != old[1]:
            self.invalidate()
        elif self.runners:
            from ..nets.hip_modules import refresh_derived_weights
            with torch.no_grad():
                refresh_derived_weights(self.net)
            self.stats['refreshes'] += 1

    def invalidate(self):
        """Drop every captured frame (their memory returns to the allocator once the graphs are collected)."""
        if self.runners:
            torch.cuda.synchronize(self.dev)
            self.stats['invalidations'] += 1
        self.runners.clear()
        self.last_use.clear()

    # ---- runners -----------------------------------------------------------------------------------------------------
    def key_of(self, shape, info, thresh, max_dets, max_out, with_filter=True):
        h, w, c = int(shape[1]), int(shape[2]), int(shape[3])
        e = None if with_filter else int(self.net._e_num_sample)
        return (h, w, c, tuple(float(v) for v in np.asarray(info, dtype=np.float32)), float(thresh), int(max_dets),
                int(max_out), bool(with_filter), e, cfg_fingerprint(self.net))

    def stream(self, lane):
        return self.streams[lane % self.n_streams]

    def runner(self, shape, info, thresh=0.5, max_dets=100, max_out=None, lane=0, with_filter=True):
        """The captured frame for this problem on lane ``lane``, or None (run this frame eagerly)."""
        max_out = max_dets if max_out is None else max_out
        key = self.key_of(shape, info, thresh, max_dets, max_out, with_filter)
        lane %= self.n_streams
        self.clock += 1
        seen = self.sightings.get(key, 0) + 1
        self.sightings[key] = seen
        if key in self.uncapturable:
            self.stats['eager'] += 1
            return None
        lanes = self.runners.get(key)
        if lanes is None:
            # the very first problem of this pool is captured at first sight
            need = 1 if (not self.runners and len(self.sightings) == 1) else max(self.capture_after, 1)
            if seen < need:
                self.stats['eager'] += 1
                return None
            if len(self.runners) >= self.max_keys:
                victim = min(self.runners, key=lambda k: self.last_use.get(k, 0))
                torch.cuda.synchronize(self.dev)
                del self.runners[victim]
                self.last_use.pop(victim, None)
            lanes = self.runners[key] = [None] * self.n_streams
        self.last_use[key] = self.clock
        r = lanes[lane]
        if r is None:
            torch.cuda.synchronize(self.dev)      # captures happen with the device idle (no replay of another lane in flight)
            tuned = any(x is not None for ls in self.runners.values() for x in ls)
            try:
                r = lanes[lane] = FrameRunner(self.net, key[0], key[1], key[2], np.asarray(key[3], np.float32), thresh, max_dets,
                                              use_graph=self.use_graph, warmup=self.warmup if not tuned else 1,
                                              autotune=self.autotune and not any(x is not None for x in lanes),
                                              max_out=max_out, with_filter=with_filter, e_num_sample=key[8])
            except Exception as e:      # a forward that cannot be captured (a host read-back inside): this problem runs eagerly
                import warnings
                warnings.warn("FramePool: frames of shape %s run EAGERLY, their forward pass could not be captured: %s: %s"
                              % (key[:3], type(e).__name__, e))
                torch.cuda.synchronize(self.dev)
                self.uncapturable.add(key)
                if not any(x is not None for x in lanes):
                    self.runners.pop(key, None)
                self.stats['eager'] += 1
                return None
            self.stats['captures'] += 1
            self._stamp = self._weights_stamp()   # the warm-up may have created derived tensors; versions are unchanged
        self.stats['replays'] += 1
        return r
