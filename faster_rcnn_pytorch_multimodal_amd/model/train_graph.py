"""One training step of the detector as a replayable hipGraph.

``Network.train_step`` (lib/model/train_val.py:458 -> forward, losses, ``loss.backward()``) launches ~1000 kernels per
frame from Python autograd: on the res101+FPN step the device is idle 13-29 % of the wall time waiting for launches
(profiles/r03_train_step.md), and the independent filter-gradient kernels never overlap the data-gradient chain because
the host cannot feed two streams fast enough.  Here the whole step - channel pad, backbone, FPN, RPN, proposal layer,
anchor / proposal target layers, RoIAlign, tail, the four losses and the complete backward pass with the filter
gradients on a side stream - is captured ONCE per problem shape and replayed per frame:

  * everything data dependent already stays on the device (proposal counts, sampled RoIs, target counts), so a step has
    no host synchronisation inside;
  * the per-step sampling seeds of the target layers come through a two-word device tensor (``frcnn_*_target_layer``'s
    ``seed_dev``), rewritten before each replay - a replayed launch keeps its scalar arguments;
  * gradients ACCUMULATE in place into ``param.grad`` (pseudo batches of cfg.TRAIN.BATCH_SIZE frames,
    train_val.py:379-382); the optimizer step, the gradient clip and the data-parallel all-reduce stay outside the graph;
  * weight-derived tensors (KRSC / transposed / fused filters) live in storage-stable caches
    (``nets.hip_modules.stable_store``) that ``refresh_derived_weights`` re-derives in place after every optimizer step.

A graph is specific to (H, W, C, number of gt boxes); ``Network.train_step`` keeps one runner per key when
``net.enable_train_graphs()`` was called and falls back to the eager step for anything a graph cannot express
(BatchNorm on batch statistics, the uncertainty heads' counter-based draws, don't-care boxes).
"""
import numpy as np
import torch

from ..layer_utils.anchor_target_layer import _draw_seed
from ..nets import autograd_ops
from ..nets.hip_modules import refresh_derived_weights
from .config import cfg


def graphable(net, blobs):
    """Why this step cannot run as a graph (a string), or None."""
    from ..nets import uncertainty
    if uncertainty.enabled():
        return "uncertainty heads draw from a host-side counter"
    if cfg.NET_TYPE != 'image':
        return "LiDAR detector: BatchNorm layers train on batch statistics through host-tracked state"
    if cfg.RESNET.FIXED_BLOCKS == -1:
        return "FIXED_BLOCKS == -1: BatchNorm on batch statistics"
    if cfg.TRAIN.IGNORE_DC and blobs.get('gt_boxes_dc') is not None and len(blobs['gt_boxes_dc']) > 0:
        return "don't-care boxes"
    if len(blobs['gt_boxes']) == 0:
        return "no ground-truth boxes"
    return None


class TrainStepRunner:
    """``run(blobs)`` -> (loss (device scalar tensor), candidate counts (device int32)) with the gradients of this frame
    added to every ``param.grad``."""

    def __init__(self, net, height, width, channels, num_gt, info, warmup=2, autotune=True):
        self.net = net
        self.info = np.asarray(info, dtype=np.float32).copy()
        dev = torch.device(net._device)
        self.static_in = torch.zeros((1, height, width, channels), dtype=torch.float32, device=dev)
        self.static_gt = torch.zeros((num_gt, 5), dtype=torch.float32, device=dev)
        self.seed_dev = torch.zeros((2,), dtype=torch.int32, device=dev)
        self.key = (height, width, channels, num_gt, tuple(float(v) for v in self.info))
        from .. import ops
        # every gradient buffer exists before the capture: the captured backward then ACCUMULATES in place
        for p in net.parameters():
            if p.requires_grad and p.grad is None:
                p.grad = torch.zeros_like(p)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        saved = [p.grad.clone() for p in net.parameters() if p.requires_grad]
        with torch.cuda.stream(side):
            ops.set_conv_autotune(autotune)
            try:
                for _ in range(max(warmup, 1)):
                    self._step()
            finally:
                torch.cuda.synchronize(dev)
                ops.set_conv_autotune(False)
            self._step()              # once more with the tuned plans: allocator and caches warm
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        # Filter gradients are accumulated into param.grad by this package's own launches on the side stream
        # (autograd_ops._wgrad), never by autograd's AccumulateGrad nodes: those run on the stream the parameter was
        # created on (the default stream, outside the capture) and were measured to race with the captured backward
        # (up to 1 % gradient error); the side-stream form is exact and lets the filter gradients overlap the chain.
        self.graph = torch.cuda.CUDAGraph()
        prev = autograd_ops.ASYNC_WGRAD
        autograd_ops.ASYNC_WGRAD = True
        try:
            with torch.cuda.graph(self.graph):
                self.loss, self.counts = self._step()
        finally:
            autograd_ops.ASYNC_WGRAD = prev
        # the warm-up and capture passes ran on zero inputs: drop what they added to the gradients
        with torch.no_grad():
            for p, g in zip([p for p in net.parameters() if p.requires_grad], saved):
                p.grad.copy_(g)

    def _step(self):
        net = self.net
        net._seed_dev = self.seed_dev
        try:
            net.forward(self.static_in, self.info, self.static_gt, None, mode='TRAIN')
        finally:
            net._seed_dev = None
        loss = net._losses['total_loss']
        counts = net._proposal_targets.get('counts')
        net.backward(loss)
        self.losses = dict(net._losses)
        return loss, counts

    def run(self, blobs):
        data, gt = blobs['data'], blobs['gt_boxes']
        if isinstance(data, np.ndarray):
            data = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32))
        if isinstance(gt, np.ndarray):
            gt = torch.from_numpy(np.ascontiguousarray(gt, dtype=np.float32))
        self.static_in.copy_(data, non_blocking=True)
        self.static_gt.copy_(gt[:, :5], non_blocking=True)
        self.seed_dev.copy_(torch.tensor([_draw_seed(), _draw_seed()], dtype=torch.int32))    # 8 bytes, host -> device
        self.graph.replay()
        return self.loss, self.counts


def after_optimizer_step(net):
    """The parameters changed in place: re-derive the cached filters the captured graphs read."""
    return refresh_derived_weights(net)
