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


def inline_graphs_supported():
    """May a training step be captured as ONE chain (filter gradients in line)?  Only when the HIP runtime was started with
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: ROCm 7's packet-captured replay of single-chain graphs replays this step wrongly from
    the second replay on (wrong filter gradients; profiles/r03_train_step.md), the general replay path is correct."""
    import os
    return os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') == '0'


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

    def __init__(self, net, height, width, channels, num_gt, info, warmup=2, autotune=True, grads=None, inline=False):
        """``grads``: gradient buffers (one per trainable parameter, in net.parameters() order) the captured backward
        accumulates into; default: the parameters' own ``.grad`` (created as zeros when missing).  A pipeline slot passes
        its private buffers (``TrainPipeline``).  ``inline``: capture the filter gradients in line (one chain, see
        ``inline_graphs_supported``) instead of on a side stream."""
        self.net = net
        self.info = np.asarray(info, dtype=np.float32).copy()
        dev = torch.device(net._device)
        self.static_in = torch.zeros((1, height, width, channels), dtype=torch.float32, device=dev)
        self.static_gt = torch.zeros((num_gt, 5), dtype=torch.float32, device=dev)
        self.seed_dev = torch.zeros((2,), dtype=torch.int32, device=dev)
        self.key = (height, width, channels, num_gt, tuple(float(v) for v in self.info))
        from .. import ops
        # every gradient buffer exists before the capture: the captured backward then ACCUMULATES in place
        params = [p for p in net.parameters() if p.requires_grad]
        for p in params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        own = [p.grad for p in params]
        if grads is not None:                       # warm-up and capture see the slot's buffers as the gradients
            for p, g in zip(params, grads):
                p.grad = g
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
        prev = (autograd_ops.ASYNC_WGRAD, autograd_ops.WGRAD_ON_SIDE_STREAM, autograd_ops.GROUP_WGRAD)
        autograd_ops.ASYNC_WGRAD = True
        autograd_ops.WGRAD_ON_SIDE_STREAM = not inline
        # in line, the grouped filter gradients of a stage (autograd_ops.GROUP_WGRAD) cost nothing in overlap - there is no side
        # chain to trail - and save 1.3 ms of kernel time per step
        autograd_ops.GROUP_WGRAD = bool(inline) or prev[2]
        self.inline = bool(inline)
        try:
            with torch.cuda.graph(self.graph):
                self.loss, self.counts = self._step()
        finally:
            autograd_ops.ASYNC_WGRAD, autograd_ops.WGRAD_ON_SIDE_STREAM, autograd_ops.GROUP_WGRAD = prev
        # the warm-up and capture passes ran on zero inputs: drop what they added to the gradients
        with torch.no_grad():
            for p, g in zip(params, saved):
                p.grad.copy_(g)
        for p, g in zip(params, own):
            p.grad = g

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


class TrainPipeline:
    """Several frames of ONE pseudo batch in flight.

    Between two optimizer steps the weights do not change (lib/model/train_val.py:379-382: the optimizer steps every
    cfg.TRAIN.BATCH_SIZE frames), so the frames of a pseudo batch are independent.  Slot s owns a HIP stream, its own
    captured graph(s) and its own set of gradient buffers (190 MB each; 288 GB of HBM make that free); ``submit`` replays a
    frame on the next slot without waiting (the host never blocks on a loss it does not need yet), ``collect`` returns
    losses in submission order, ``flush`` adds the slots' gradients into ``param.grad`` (one multi-tensor add per slot)
    before the optimizer step.  Same arithmetic as the sequential loop except for the order in which the frames'
    gradients are summed.
    MEASURED (profiles/r03_train_step.md): on the res101+FPN 1000x600 step 2 / 4 frames in flight run at 16.7 / 17.1 ms per
    frame against 17.0 ms for one - no gain.  Unlike the inference frame, the captured training step already runs two
    chains side by side (data-gradient chain || filter gradients: 24.6 -> 17.0 ms) and that fills the chip; what is left is
    the efficiency of the individual small-GEMM kernels, not idle CUs.  The class stays as the host-side pipelining of the
    solver loop (cfg.TRAIN.FRAMES_IN_FLIGHT, default 1)."""

    def __init__(self, net, slots=4, max_graphs=8, inline=None):
        """``inline``: capture every slot's step as one chain so that the slots' replays overlap (default: when
        ``inline_graphs_supported()``); a forked graph (filter gradients on a side stream) is correct everywhere but its
        replays do not overlap - then the pipeline only hides the host's launch / read-back time."""
        self.inline = inline_graphs_supported() if inline is None else bool(inline)
        self._checked = False
        self.net = net
        self.dev = torch.device(net._device)
        self.slots = max(1, int(slots))
        self.max_graphs = int(max_graphs)
        self.params = [p for p in net.parameters() if p.requires_grad]
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        self.grads = [[torch.zeros_like(p) for p in self.params] for _ in range(self.slots)]
        self.streams = [torch.cuda.Stream(device=self.dev) for _ in range(self.slots)]
        self.runners = [dict() for _ in range(self.slots)]
        self.pending = [None] * self.slots         # per slot: [runner, event, loss value or None, counts or None]
        self.order = []                            # slots in submission order, not collected yet
        self.next_slot = 0

    def _finish(self, s):
        ent = self.pending[s]
        if ent is not None and ent[2] is None:
            ent[1].synchronize()
            ent[2] = float(ent[0].loss.item())
            ent[3] = [int(v) for v in ent[0].counts[:2].cpu()] if ent[0].counts is not None else None
        return ent

    def submit(self, blobs):
        """Queue one frame (forward + backward, gradients into the slot's buffers).  Returns the slot index."""
        why = graphable(self.net, blobs)
        if why is not None:
            raise RuntimeError("this frame cannot run as a captured training step: " + why)
        s = self.next_slot
        self.next_slot = (s + 1) % self.slots
        if self.pending[s] is not None and self.pending[s][2] is None:
            raise RuntimeError("TrainPipeline: slot %d still holds an uncollected frame (collect() before submitting more than "
                               "%d frames)" % (s, self.slots))
        data, info = blobs['data'], np.asarray(blobs['info'], dtype=np.float32)
        key = (int(data.shape[1]), int(data.shape[2]), int(data.shape[3]), int(len(blobs['gt_boxes'])),
               tuple(float(v) for v in info))
        runner = self.runners[s].get(key)
        if runner is None:
            if len(self.runners[s]) >= self.max_graphs:
                raise RuntimeError("TrainPipeline: more than %d distinct frame shapes" % self.max_graphs)
            torch.cuda.synchronize(self.dev)       # captures happen with the device idle
            runner = TrainStepRunner(self.net, key[0], key[1], key[2], key[3], info, grads=self.grads[s],
                                     autotune=not any(self.runners), inline=self.inline)
            self.runners[s][key] = runner
            if self.inline and not self._checked:
                self._check_replays(runner, blobs, self.grads[s])
                self._checked = True
        st = self.streams[s]
        st.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(st):
            runner.run(blobs)
            ev = torch.cuda.Event()
            ev.record(st)
        self.pending[s] = [runner, ev, None, None]
        self.order.append(s)
        return s

    def _check_replays(self, runner, blobs, grads):
        """A single-chain graph must give the same gradients on every replay (the runtime fault described at
        ``inline_graphs_supported`` shows from the second replay on): replay the first captured step three times on this
        frame with the same sampling seeds and compare the increments.  Raises instead of training on wrong gradients."""
        torch.cuda.synchronize(self.dev)
        held = [g.clone() for g in grads]
        incs = []
        with torch.no_grad():
            for _ in range(3):
                torch._foreach_zero_(grads)
                state = torch.random.get_rng_state()
                runner.run(blobs)
                torch.random.set_rng_state(state)               # the same two sampling seeds for every replay
                torch.cuda.synchronize(self.dev)
                incs.append([g.clone() for g in grads])
            for g, h in zip(grads, held):
                g.copy_(h)
        scale = max(float(a.abs().max()) for a in incs[0]) or 1.0
        worst = max(float((a - b).abs().max()) for k in (1, 2) for a, b in zip(incs[0], incs[k])) / scale
        if not worst <= 1e-3:
            raise RuntimeError("TrainPipeline: a replayed single-chain training graph does not reproduce its own gradients "
                               "(deviation %.3e of their scale).  Start the process with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, or "
                               "build the pipeline with inline=False." % worst)

    def in_flight(self):
        return len(self.order)

    def collect(self):
        """(loss, [fg, bg] candidate counts) of the OLDEST uncollected frame (blocks until that frame is done)."""
        s = self.order.pop(0)
        ent = self._finish(s)
        self.net._losses = ent[0].losses
        return ent[2], ent[3]

    def flush(self):
        """All submitted frames are done and their gradients are added to ``param.grad``; the slot buffers are zeroed.  Call
        with every frame collected, before clipping / the optimizer step."""
        if self.order:
            raise RuntimeError("TrainPipeline.flush: %d frames not collected" % len(self.order))
        cur = torch.cuda.current_stream(self.dev)
        for st in self.streams:
            cur.wait_stream(st)
        own = [p.grad for p in self.params]
        with torch.no_grad():
            for g in self.grads:
                torch._foreach_add_(own, g)
                torch._foreach_zero_(g)
        for st in self.streams:
            st.wait_stream(cur)
