"""One frame of the detector as a replayable hipGraph.

The reference runs its per-frame loop (lib/model/test.py:183-228) eagerly: ~115 kernel launches from
Python plus one host copy per class.  Here the whole frame — channel pad, ResNet-101 head, RPN, proposal
sort/NMS, RoIAlign, layer4 on the RoIs, detection tail, per-class filter — is captured ONCE into a
hipGraph (every kernel of libfrcnn_hip.so is asynchronous, allocation-free and keeps data-dependent
counts on the device) and replayed per frame, so the launch-bound small layers (layer3's 69 convs of
~10 us each) are not paced by the host.

Several runners may share one ``net`` and replay concurrently on different HIP streams (bench.py keeps 4
frames in flight): everything a frame WRITES — activations, workspaces, proposal buffers, the detection
record — is allocated during the capture and therefore private to the runner's graph; what runners share
(filters, folded BatchNorm scale/shift, anchors, tuned plans) is read-only after the eager warm-up.
``tests/test_timed_path.py`` replays distinct frames through that arrangement and compares every record
with the eager path bit for bit and with the CPU oracle.
"""
import numpy as np
import torch

from .test import detect_frame_device


class FrameRunner:
    """Fixed-shape frame pipeline: ``run(frame)`` -> (dets (K, max_out, 5), det_count (K,)) device tensors
    that are overwritten by the next ``run``.

    ``rpn_override_shape`` = (1, H/16, W/16, ld): the runner owns a static buffer of that shape which replaces the
    RPN head's output [bg logits | fg logits | deltas | pad] (the evaluation hook ``Network._rpn_override``, used by
    the parity tests and bench.py's mAP leg to make the proposal stage well-conditioned); ``run(frame, rpn=...)``
    fills it."""

    def __init__(self, net, height, width, channels, info, thresh=0.5, max_dets=100, use_graph=True, warmup=2,
                 autotune=True, rpn_override_shape=None):
        from ..nets import uncertainty
        if uncertainty.enabled():
            raise NotImplementedError("FrameRunner with cfg.UC.*: the seed of the counter-based draws is a launch argument, "
                                      "a replayed graph would repeat the masks of the captured frame; use detect_frame_device")
        self.net = net
        self.info = np.asarray(info, dtype=np.float32)
        self.thresh, self.max_dets = thresh, max_dets
        dev = torch.device(net._device)
        self.static_in = torch.zeros((1, height, width, channels), dtype=torch.float32, device=dev)
        self.static_rpn = (torch.zeros(tuple(rpn_override_shape), dtype=torch.float32, device=dev)
                           if rpn_override_shape is not None else None)
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
            with torch.cuda.graph(self.graph):
                self.out = self._frame()

    def _frame(self):
        self.net._rpn_override = self.static_rpn
        try:
            out = detect_frame_device(self.net, self.static_in, self.info, self.thresh, self.max_dets, self.max_dets)
        finally:
            self.net._rpn_override = None
        self.predictions = self.net._predictions     # this runner's (graph-private) intermediate tensors
        return out

    def run(self, frame, rpn=None, poison=False):
        """frame: (1,H,W,C) float32 tensor - device, or (pinned) host memory: the copy into the graph's input buffer is then
        the asynchronous host->device upload on the current stream - or a numpy blob.
        ``poison``: overwrite the graph's output buffers (NaN detections, -1 counts) before the replay, so that a replay
        which did not execute cannot leave a plausible record behind (bench.py's verification)."""
        if isinstance(frame, np.ndarray):
            frame = torch.from_numpy(frame)
        self.static_in.copy_(frame, non_blocking=True)
        if self.static_rpn is not None:
            if rpn is None:
                raise ValueError("this runner was built with an RPN override buffer: run(frame, rpn=...)")
            self.static_rpn.copy_(rpn, non_blocking=True)
        if self.graph is not None:
            if poison:
                self.out[0].fill_(float('nan'))
                self.out[1].fill_(-1)
            self.graph.replay()
        else:
            self.out = self._frame()
        return self.out
