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

A graph is specific to (H, W, C, capacity of the gt buffer); ``Network.train_step`` keeps one runner per key when
``net.enable_train_graphs()`` was called and falls back to the eager step for what a graph cannot express (``graphable``:
BatchNorm with momentum=None, more don't-care boxes than the buffer holds).
"""
import numpy as np
import torch

from ..layer_utils.anchor_target_layer import _draw_seed
from ..nets import autograd_ops
from ..nets.hip_modules import refresh_derived_weights
from .config import cfg
from .frame_graph import capture


GT_CAPACITY_MIN = 32      # rows of a captured step's gt buffer: max(32, next power of two of the frame's boxes)
DC_CAPACITY = 64          # rows of a captured step's don't-care buffer (cfg.TRAIN.IGNORE_DC); unused rows hold DC_FAR
# a box no proposal overlaps: IoU 0 < cfg.TRAIN.DC_THRESH, so a padding row never masks a RoI (proposal_target_layer.py:180-187
# drops RoIs by their BEST overlap with a don't-care box) - the buffer needs no device-side count
DC_FAR = (-1.0e6, -1.0e6, -1.0e6 + 1.0, -1.0e6 + 1.0)


def gt_capacity(num_gt):
    cap = GT_CAPACITY_MIN
    while cap < num_gt:
        cap *= 2
    return cap


def capture_switches():
    """The process-wide switches a captured training step bakes in, as a hashable part of every runner key: a holder of
    captured steps (Network._train_graphs, SolverWrapper, TrainPipeline) then never replays a step that was captured under
    other settings."""
    from .. import _hip, ops
    return (bool(cfg.TRAIN.IGNORE_DC), bool(ops.BN_FUSED_FINAL), bool(autograd_ops.BN_GRADS_IN_KERNEL),
            int(autograd_ops.WGRAD_SIDE_STREAMS), int(_hip.load().frcnn_settings_signature()))


def packet_capture_disabled():
    """Was the HIP runtime started with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (the general replay path for every graph)?"""
    import os
    return os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE') == '0'


def inline_graphs_supported():
    """May a training step be captured as ONE chain (filter gradients in line)?  Yes.  Root cause of the round-3 fault
    ("wrong filter gradients from the second replay on" under ROCm 7's packet-captured replay of single-chain graphs):
    the hipMemsetAsync / hipMemcpyAsync NODES this library's launch sequences put into the chain (anchor target layer,
    strided 1x1 data gradient, RPN loss) - tools/train_graph_trace.py: with them the chain diverges from replay 2, with the
    same initialisations as kernel launches (csrc/common.hip fill_bytes / copy_bytes, the default since library version
    107) it replays correctly on the default runtime path.  ``TrainStepRunner`` additionally refuses an in-line capture
    that still contains a memset node (``InlineCaptureUnsafe``: e.g. a torch reduction's semaphore memset) and
    ``TrainPipeline`` checks every captured runner's replays against each other.
    The root cause is EMPIRICAL (an in-situ A/B, profiles/r04_graph_replay_root_cause.md; the stand-alone reproducer
    tools/graph_replay_repro.hip does not trigger, and torch's own memcpy nodes remain in the chain): what protects a run is
    the per-runner replay check.  ``TrainPipeline(inline=False)`` / ``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0`` are the
    conservative settings (forked graphs / the runtime's general replay path)."""
    return True


class InlineCaptureUnsafe(RuntimeError):
    """An in-line (single-chain) capture holds node kinds the packet-captured replay path is known to mishandle."""


_NODE_KINDS = {0: 'kernel', 1: 'memcpy', 2: 'memset', 3: 'host', 4: 'graph', 5: 'empty', 6: 'wait_event', 7: 'event_record',
               8: 'ext_sem_signal', 9: 'ext_sem_wait', 10: 'mem_alloc', 11: 'mem_free', 12: 'memcpy_from_symbol',
               13: 'memcpy_to_symbol'}


def _hip_runtime():
    """The libamdhip64 instance THIS process's torch runs on (torch ships its own copy: opening another one by name, or
    picking another mapped copy, would hand its entry points graph handles of a different runtime)."""
    import ctypes
    import os
    with open('/proc/self/maps') as f:
        paths = sorted({line.split()[-1] for line in f if 'libamdhip64' in line})
    if not paths:
        raise RuntimeError("libamdhip64 is not loaded in this process")
    torch_lib = os.path.join(os.path.dirname(os.path.abspath(torch.__file__)), 'lib')
    own = [p for p in paths if os.path.dirname(os.path.realpath(p)) == os.path.realpath(torch_lib)]
    if len(paths) > 1 and not own:
        raise RuntimeError("several libamdhip64 copies are mapped (%s) and none is torch's: cannot tell which runtime owns "
                           "torch's graphs" % ", ".join(paths))
    return ctypes.CDLL((own or paths)[0])


def _hip_ok(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with hipError %d" % (what, rc))


def graph_node_kinds(graph):
    """(histogram {kind: count}, nodes, edges) of a ``torch.cuda.CUDAGraph`` captured with ``keep_graph=True`` (before or
    after ``instantiate()``), read through hipGraphGetNodes / hipGraphNodeGetType / hipGraphGetEdges.  A capture of one
    stream is a chain: edges == nodes - 1."""
    import ctypes
    hip = _hip_runtime()
    raw = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    _hip_ok(hip.hipGraphGetNodes(raw, None, ctypes.byref(n)), "hipGraphGetNodes (count)")
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    _hip_ok(hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)), "hipGraphGetNodes")
    hist = {}
    for i in range(n.value):
        t = ctypes.c_int(-1)
        _hip_ok(hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(t)), "hipGraphNodeGetType")
        kind = _NODE_KINDS.get(t.value, 'type%d' % t.value)
        hist[kind] = hist.get(kind, 0) + 1
    e = ctypes.c_size_t(0)
    _hip_ok(hip.hipGraphGetEdges(raw, None, None, ctypes.byref(e)), "hipGraphGetEdges")
    return hist, int(n.value), int(e.value)


def graphable(net, blobs):
    """Why this step cannot run as a graph (a string), or None."""
    if cfg.TRAIN.IGNORE_DC and blobs.get('gt_boxes_dc') is not None and len(blobs['gt_boxes_dc']) > DC_CAPACITY:
        return "more than %d don't-care boxes (the captured step's buffer)" % DC_CAPACITY
    if len(blobs['gt_boxes']) == 0:
        return "no ground-truth boxes"
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training and m.track_running_stats and m.momentum is None:
            return "BatchNorm with momentum=None (cumulative average: the factor is computed on the host per step)"
    return None


class TrainStepRunner:
    """``run(blobs)`` -> (loss (device scalar tensor), candidate counts (device int32)) with the gradients of this frame
    added to every ``param.grad``.

    One runner serves every frame of its (H, W, C, info) problem: ground-truth boxes go through a buffer of ``gt_cap`` rows
    (``gt_capacity``: 32, 64, ...) with the live row count in a device word that the target-layer kernels read
    (``frcnn_*_target_layer``'s ``num_gt_dev``), like the sampling seeds.  LiDAR frames carry two such buffers (BEV
    rectangles and 3-D rows, split on the host before the replay).  BatchNorm layers in train() mode (LiDAR backbone,
    FIXED_BLOCKS == -1) update their running statistics inside the captured launches; ``run`` bumps the modules' host-side
    statistics version afterwards so that an eval-mode forward re-folds them."""

    def __init__(self, net, height, width, channels, num_gt, info, warmup=2, autotune=True, grads=None, inline=False,
                 group_wgrad=None, debug_dump=None, defer_bn_stats=False):
        """``num_gt``: boxes of the first frame (fixes the buffer capacity).  ``grads``: gradient buffers (one per trainable
        parameter, in net.parameters() order) the captured backward accumulates into; default: the parameters' own
        ``.grad`` (created as zeros when missing).  A pipeline slot passes its private buffers (``TrainPipeline``).
        ``inline``: capture the filter gradients in line (one chain) instead of on a side stream.
        ``defer_bn_stats``: the captured BatchNorm launches leave the frame's batch mean / unbiased variance in buffers
        private to this runner instead of updating the modules' running statistics; ``fold_bn_stats()`` applies the update
        ``running = (1 - momentum) * running + momentum * stat`` afterwards.  Runners of different pipeline slots replay
        CONCURRENTLY: an in-kernel read-modify-write of the shared statistics would lose updates (and count); the
        pipeline folds in frame order instead, which is the reference's sequential per-frame update."""
        self.net = net
        self.info = np.asarray(info, dtype=np.float32).copy()
        dev = torch.device(net._device)
        self.lidar = cfg.NET_TYPE == 'lidar'
        self.gt_cap = gt_capacity(int(num_gt))
        self.static_in = torch.zeros((1, height, width, channels), dtype=torch.float32, device=dev)
        self.static_gt = torch.zeros((self.gt_cap, 5), dtype=torch.float32, device=dev)
        self.static_true_gt = torch.zeros((self.gt_cap, 8), dtype=torch.float32, device=dev) if self.lidar else None
        self.gt_count = torch.ones((1,), dtype=torch.int32, device=dev)
        # don't-care boxes (cfg.TRAIN.IGNORE_DC, lib/roi_data_layer/minibatch.py:168-176): fixed-capacity buffer padded with a
        # far-away box; the flag is part of the capture (a runner is built under one cfg)
        self.dc_far = torch.tensor([DC_FAR], dtype=torch.float32, device=dev).repeat(DC_CAPACITY, 1) if cfg.TRAIN.IGNORE_DC else None
        self.static_dc = self.dc_far.clone() if self.dc_far is not None else None
        self.seed_dev = torch.zeros((2,), dtype=torch.int32, device=dev)
        from .. import ops as _ops
        self.wgrad_counters = _ops.CounterArena(dev)      # tile counters of this runner's filter-gradient launches (_step)
        from ..nets import uncertainty
        self.uc_seed_dev = torch.zeros((1,), dtype=torch.int32, device=dev) if uncertainty.enabled() else None
        self.key = (height, width, channels, self.gt_cap, tuple(float(v) for v in self.info))
        self.bn_modules = [m for m in net.modules()
                           if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training and m.track_running_stats]
        self.bn_private = None
        if defer_bn_stats and self.bn_modules:
            # [mean, unbiased variance, "the captured step launched this module"]
            self.bn_private = {id(m): [torch.zeros_like(m.running_mean), torch.zeros_like(m.running_var), False]
                               for m in self.bn_modules}
        from .. import ops
        # warm-up / capture frame: zeros with ONE plausible box, so that the target layers see a regular problem
        self._fill_placeholder_gt(height, width)
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
        # the warm-up / capture steps run BatchNorm on batch statistics of a zero frame: keep the running statistics
        stats = [(m, m.running_mean.clone(), m.running_var.clone(),
                  m.num_batches_tracked.clone() if m.num_batches_tracked is not None else None) for m in self.bn_modules]
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
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)
        prev = (autograd_ops.ASYNC_WGRAD, autograd_ops.WGRAD_ON_SIDE_STREAM, autograd_ops.GROUP_WGRAD)
        autograd_ops.ASYNC_WGRAD = True
        autograd_ops.WGRAD_ON_SIDE_STREAM = not inline
        # The grouped filter gradients of a stage (autograd_ops.GROUP_WGRAD: the 22 equal Bottlenecks of layer3 in one unsplit
        # launch).  In line they cost nothing in overlap - there is no side chain to trail.  On the side stream they used to lose
        # (they start when their stage's backward is over); with the LDS-DMA kernel that adds into param.grad itself they win:
        # 14.7 -> 14.3 ms per captured FPN step (round 5, same-box A/B; groups of 8: 14.3, of 4: 14.6)
        autograd_ops.GROUP_WGRAD = True if group_wgrad is None else bool(group_wgrad)
        if debug_dump:
            self.graph.enable_debug_mode()
        self.inline = bool(inline)
        autograd_ops.BN_STAT_SINK = self.bn_private
        try:
            with capture(self.graph):
                self.loss, self.counts = self._step()
        finally:
            autograd_ops.ASYNC_WGRAD, autograd_ops.WGRAD_ON_SIDE_STREAM, autograd_ops.GROUP_WGRAD = prev
            autograd_ops.BN_STAT_SINK = None
        if debug_dump:
            self.graph.debug_dump(debug_dump)
        self.node_kinds, self.nodes, self.edges = graph_node_kinds(self.graph)
        # the warm-up and capture passes ran on a placeholder frame: drop what they added to the gradients and statistics
        with torch.no_grad():
            for p, g in zip(params, saved):
                p.grad.copy_(g)
            for m, mean, var, nbt in stats:
                m.running_mean.copy_(mean)
                m.running_var.copy_(var)
                if nbt is not None:
                    m.num_batches_tracked.copy_(nbt)
        for p, g in zip(params, own):
            p.grad = g
        if self.inline and self.node_kinds.get('memset', 0) and not packet_capture_disabled():
            raise InlineCaptureUnsafe("single-chain capture of the training step holds %d memset nodes (kinds %s): the runtime's "
                                      "packet-captured replay is not trusted with them" % (self.node_kinds['memset'], self.node_kinds))
        self.graph.instantiate()

    def _fill_placeholder_gt(self, height, width):
        box = torch.tensor([width * 0.25, height * 0.25, width * 0.6, height * 0.6, 1.0])
        self.static_gt.zero_()
        self.static_gt[0].copy_(box)
        if self.lidar:
            t = torch.tensor([width * 0.425, height * 0.425, 1.0, width * 0.35, height * 0.35, 2.0, 0.0, 1.0])
            self.static_true_gt.zero_()
            self.static_true_gt[0].copy_(t)
        self.gt_count.fill_(1)

    def _step(self):
        from .. import ops
        net = self.net
        net._seed_dev, net._gt_count_dev, net._uc_seed_dev = self.seed_dev, self.gt_count, self.uc_seed_dev
        # the filter-gradient launches of this runner count their pixel-split workgroups in tile counters of the runner's
        # own (ops.wgrad_counter_arena): the same range for the same launch in every pass, baked into the graph, shared
        # with no other graph or eager launch
        self.wgrad_counters.rewind()
        with ops.wgrad_counter_arena(self.wgrad_counters):
            try:
                gt = (self.static_gt, self.static_true_gt) if self.lidar else self.static_gt
                net.forward(self.static_in, self.info, gt, self.static_dc, mode='TRAIN')
            finally:
                net._seed_dev = net._gt_count_dev = net._uc_seed_dev = None
            loss = net._losses['total_loss']
            counts = net._proposal_targets.get('counts')
            net.backward(loss)
        self.losses = dict(net._losses)
        return loss, counts

    def fits(self, blobs):
        return len(blobs['gt_boxes']) <= self.gt_cap

    def run(self, blobs):
        data, gt = blobs['data'], blobs['gt_boxes']
        if isinstance(data, np.ndarray):
            data = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32))
        g = int(len(gt))
        if g < 1 or g > self.gt_cap:
            raise ValueError("TrainStepRunner: %d gt boxes, this runner holds 1..%d" % (g, self.gt_cap))
        if self.lidar:
            # blobs['gt_boxes'] rows [xc,yc,zc,l,w,h,ry,cls] (minibatch.py:147-167) -> the two forms the target layers take
            # (Network.forward does the same split for the eager step)
            from ..utils.bbox import bbaa_graphics_gems
            gt_np = gt.detach().cpu().numpy() if isinstance(gt, torch.Tensor) else np.asarray(gt, dtype=np.float32)
            if gt_np.ndim != 2 or gt_np.shape[1] != 8:
                raise ValueError("LiDAR gt_boxes must be (G, 8) [xc,yc,zc,l,w,h,ry,cls], got %s" % (gt_np.shape,))
            aabb = np.concatenate((bbaa_graphics_gems(gt_np[:, :7]), gt_np[:, 7:8]), 1).astype(np.float32)
            self.static_true_gt[:g].copy_(torch.from_numpy(np.ascontiguousarray(gt_np, dtype=np.float32)), non_blocking=True)
            self.static_gt[:g].copy_(torch.from_numpy(np.ascontiguousarray(aabb)), non_blocking=True)
        else:
            if isinstance(gt, np.ndarray):
                gt = torch.from_numpy(np.ascontiguousarray(gt, dtype=np.float32))
            self.static_gt[:g].copy_(gt[:, :5], non_blocking=True)
        if self.static_dc is not None:
            dc = blobs.get('gt_boxes_dc')
            n_dc = 0 if dc is None else int(len(dc))
            if n_dc > DC_CAPACITY:
                raise ValueError("TrainStepRunner: %d don't-care boxes, this runner holds %d" % (n_dc, DC_CAPACITY))
            self.static_dc.copy_(self.dc_far, non_blocking=True)
            if n_dc:
                dc_t = dc if isinstance(dc, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(dc, dtype=np.float32)))
                self.static_dc[:n_dc].copy_(dc_t[:, :4], non_blocking=True)
        elif cfg.TRAIN.IGNORE_DC:
            raise RuntimeError("TrainStepRunner: cfg.TRAIN.IGNORE_DC was switched on after this step was captured")
        self.static_in.copy_(data, non_blocking=True)
        self.gt_count.fill_(g)
        self.seed_dev.copy_(torch.tensor([_draw_seed(), _draw_seed()], dtype=torch.int32))    # 8 bytes, host -> device
        if self.uc_seed_dev is not None:
            seed = self.net.next_uc_seed()
            self.uc_seed_dev.fill_(seed - (1 << 32) if seed >= (1 << 31) else seed)
        self.graph.replay()
        if self.bn_private is None:
            for m in self.bn_modules:      # the captured launches updated the running statistics in place
                m.__dict__['_frcnn_stats_version'] = m.__dict__.get('_frcnn_stats_version', 0) + 1
        return self.loss, self.counts

    def fold_bn_stats(self):
        """Deferred statistics (``defer_bn_stats``): apply this replay's batch statistics to the modules' running statistics
        on the current stream - bn_fwd_final_kernel's expression (csrc/batchnorm.hip: two products and a sum, each rounded)
        as multi-tensor launches over all layers: running = (1 - m) * running + m * stat, num_batches_tracked += 1."""
        if self.bn_private is None:
            return
        if getattr(self, '_fold_groups', None) is None:
            groups = {}
            for m in self.bn_modules:
                pm, pv, used = self.bn_private[id(m)]
                if not used:          # a train()-mode module the step never calls keeps its statistics, as in the eager step
                    continue
                g = groups.setdefault(float(m.momentum), ([], [], []))
                g[0].extend((m.running_mean, m.running_var))
                g[1].extend((pm, pv))
                if m.num_batches_tracked is not None:
                    g[2].append(m.num_batches_tracked)
            self._fold_groups = groups
        with torch.no_grad():
            for mom, (running, stat, counts) in self._fold_groups.items():
                keep = float(np.float32(1.0) - np.float32(mom))
                torch._foreach_mul_(running, keep)
                torch._foreach_add_(running, torch._foreach_mul(stat, float(np.float32(mom))))
                if counts:
                    torch._foreach_add_(counts, 1)
        for m in self.bn_modules:
            if self.bn_private[id(m)][2]:
                m.__dict__['_frcnn_stats_version'] = m.__dict__.get('_frcnn_stats_version', 0) + 1


def after_optimizer_step(net):
    """The parameters changed in place: re-derive the cached filters the captured graphs read."""
    return refresh_derived_weights(net)


def check_replays(net, runner, blobs, grads):
    """A single-chain graph must give the same gradients on every replay (the runtime fault described at
    ``inline_graphs_supported`` shows from the second replay on): replay the first captured step three times on this
    frame with the same sampling seeds and compare the increments.  Raises instead of training on wrong gradients."""
    torch.cuda.synchronize(torch.device(net._device))
    held = [g.clone() for g in grads]
    stats = [(m, m.running_mean.clone(), m.running_var.clone(),
              m.num_batches_tracked.clone() if m.num_batches_tracked is not None else None) for m in runner.bn_modules]
    uc_calls = getattr(net, '_uc_calls', 0)
    incs = []
    with torch.no_grad():
        for _ in range(3):
            torch._foreach_zero_(grads)
            state = torch.random.get_rng_state()
            net._uc_calls = uc_calls                   # the same dropout masks / logit noise ...
            runner.run(blobs)
            torch.random.set_rng_state(state)               # ... and the same two sampling seeds for every replay
            torch.cuda.synchronize(torch.device(net._device))
            incs.append([g.clone() for g in grads])
        for g, h in zip(grads, held):
            g.copy_(h)
        net._uc_calls = uc_calls
        for m, mean, var, nbt in stats:                     # the check must not count as three training steps
            m.running_mean.copy_(mean)
            m.running_var.copy_(var)
            if nbt is not None:
                m.num_batches_tracked.copy_(nbt)
    # per tensor, relative to that tensor's own largest increment (a corrupted small-magnitude gradient must not hide
    # behind the largest one); tensors whose increment is below 1e-6 of the global scale are compared on that floor
    top = max(float(a.abs().max()) for a in incs[0]) or 1.0
    worst, worst_i = 0.0, -1
    for i, a in enumerate(incs[0]):
        scale = max(float(a.abs().max()), 1e-6 * top)
        dev = max(float((a - incs[k][i]).abs().max()) for k in (1, 2)) / scale
        if not dev <= worst:          # NaN counts as the worst
            worst, worst_i = dev, i
    if not worst <= 1e-3:
        raise RuntimeError("a replayed single-chain training graph does not reproduce its own gradients "
                           "(gradient tensor %d deviates by %.3e of its own scale; node kinds %s)."
                           % (worst_i, worst, runner.node_kinds))


def captured_step(net, height, width, channels, num_gt, info, blobs):
    """The runner behind ``Network.train_step`` (one captured step at a time): ONE chain of kernel nodes with the filter
    gradients in line and grouped per stage when the chain replays faithfully (no memset node, three replays reproduce each
    other: ``check_replays``), else the forked form with the filter gradients on a side stream.  Measured (round 5): the
    forked graph's side branch buys nothing on this runtime - 14.2 ms per FPN step against 13.75 ms for the single chain."""
    import warnings
    if inline_graphs_supported():
        try:
            runner = TrainStepRunner(net, height, width, channels, num_gt, info, inline=True)
            check_replays(net, runner, blobs, [p.grad for p in net.parameters() if p.requires_grad])
            return runner
        except (InlineCaptureUnsafe, RuntimeError) as e:
            warnings.warn("train_step: %s; capturing the forked graph (filter gradients on a side stream) instead" % e)
            return TrainStepRunner(net, height, width, channels, num_gt, info, autotune=False, inline=False)
    return TrainStepRunner(net, height, width, channels, num_gt, info, inline=False)


class TrainPipeline:
    """Several frames of ONE pseudo batch in flight.

    Between two optimizer steps the weights do not change (lib/model/train_val.py:379-382: the optimizer steps every
    cfg.TRAIN.BATCH_SIZE frames), so the frames of a pseudo batch are independent.  Slot s owns a HIP stream, its own
    captured graph(s) and its own set of gradient buffers (190 MB each; 288 GB of HBM make that free); ``submit`` replays a
    frame on the next slot without waiting (the host never blocks on a loss it does not need yet), ``collect`` returns
    losses in submission order, ``flush`` adds the slots' gradients into ``param.grad`` (one multi-tensor add per slot)
    before the optimizer step.  Same arithmetic as the sequential loop except for the order in which the frames'
    gradients are summed.
    MEASURED: forked graphs (filter gradients on a side stream) of different slots do not overlap - 2 / 4 frames in flight ran
    at 16.7 / 17.1 ms per res101+FPN 1000x600 step against 17.0 ms for one (profiles/r03_train_step.md); single-chain graphs
    (``inline``) do: 10.3 ms per step with 3 in flight against 14.7 ms one at a time (profiles/r04_bench.json).
    cfg.TRAIN.FRAMES_IN_FLIGHT (default 3) is the solver's slot count.
    BatchNorm on batch statistics (LiDAR backbone, FIXED_BLOCKS == -1): the slots' captured launches write their frame's
    batch statistics to slot-private buffers and ``submit`` folds them into the modules' running statistics in SUBMISSION
    order (``TrainStepRunner.fold_bn_stats`` chained by an event from frame to frame) - concurrent in-kernel updates of
    the shared statistics would lose updates, and the reference updates them once per frame, in order."""

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
        # one stream per slot, chosen by MEASUREMENT to overlap (model/streams.py: HIP streams share four hardware queues, and
        # which ones collide depends on the process's history - the same pipeline ran at 9.5 or 11.4 ms per step with pool streams)
        from .streams import concurrent_streams
        self.streams, self.distinct_queues = concurrent_streams(self.slots, self.dev)
        self.runners = [dict() for _ in range(self.slots)]
        self.pending = [None] * self.slots         # per slot: [runner, event, loss value or None, counts or None]
        self.order = []                            # slots in submission order, not collected yet
        self.next_slot = 0
        self.fold_event = None                     # end of the latest frame's running-statistics update (frame order)

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
        key = (int(data.shape[1]), int(data.shape[2]), int(data.shape[3]), gt_capacity(len(blobs['gt_boxes'])),
               tuple(float(v) for v in info), capture_switches())
        runner = self.runners[s].get(key)
        if runner is None:
            if len(self.runners[s]) >= self.max_graphs:
                raise RuntimeError("TrainPipeline: more than %d distinct frame shapes" % self.max_graphs)
            torch.cuda.synchronize(self.dev)       # captures happen with the device idle
            try:
                runner = TrainStepRunner(self.net, key[0], key[1], key[2], key[3], info, grads=self.grads[s],
                                         autotune=not any(self.runners), inline=self.inline, defer_bn_stats=True)
            except InlineCaptureUnsafe as e:
                import warnings
                warnings.warn("TrainPipeline: %s; capturing forked graphs instead (correct, but replays of different slots do "
                              "not overlap)" % e)
                self.inline = False
                runner = TrainStepRunner(self.net, key[0], key[1], key[2], key[3], info, grads=self.grads[s],
                                         autotune=False, inline=False, defer_bn_stats=True)
            self.runners[s][key] = runner
            if self.inline:
                # every newly captured single-chain runner proves that its replays reproduce each other (3 replays)
                try:
                    self._check_replays(runner, blobs, self.grads[s])
                except RuntimeError as e:
                    import warnings
                    warnings.warn(str(e) + "  Falling back to forked graphs.")
                    self.inline = False
                    runner = self.runners[s][key] = TrainStepRunner(self.net, key[0], key[1], key[2], key[3], info,
                                                                    grads=self.grads[s], autotune=False, inline=False,
                                                                    defer_bn_stats=True)
        st = self.streams[s]
        st.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(st):
            runner.run(blobs)
            if runner.bn_private is not None:
                # BatchNorm on batch statistics: the slots' replays overlap, the running statistics advance one frame at a
                # time in SUBMISSION order (the reference's sequential update, lib/model/train_val.py:458 per frame) - this
                # frame's fold waits for the previous frame's
                if self.fold_event is not None:
                    st.wait_event(self.fold_event)
                runner.fold_bn_stats()
                self.fold_event = torch.cuda.Event()
                self.fold_event.record(st)
            ev = torch.cuda.Event()
            ev.record(st)
        self.pending[s] = [runner, ev, None, None]
        self.order.append(s)
        return s

    def _check_replays(self, runner, blobs, grads):
        check_replays(self.net, runner, blobs, grads)

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
