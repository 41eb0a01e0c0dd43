"""Solver loop of the reference (lib/model/train_val.py:58-503) for the HIP Network, plus the data-parallel step the
reference does not have ("solver loop + DP all-reduce").

Same schedule as ``SolverWrapper.train_model``: iterations count frames, the optimizer steps every ``batch_size``
frames (train_val.py:379-382), the learning rate is multiplied by ``cfg.TRAIN.GAMMA`` at ``stepsize + 1``
(train_val.py:383-389), a snapshot (``.pth`` weights + ``.pkl`` sampler state) every ``cfg.TRAIN.SNAPSHOT_ITERS``
and at most ``cfg.TRAIN.SNAPSHOT_KEPT`` kept (train_val.py:476-485, 294-307).  Tensorboard writers, the dataset
classes and the drawing hooks are the caller's (out of scope); ``frames`` is any object with
``next()`` returning a blob dict and optionally ``get_pointer()/set_pointer()``.

Data parallel = one process per GPU, every rank draws its own frame, gradients accumulate locally for
``batch_size`` frames in ONE flat fp32 buffer (``GradientBucket``: every ``param.grad`` is a view into it) and a single
RCCL all-reduce of that buffer runs right before the optimizer step - one 190 MB collective per 16 frames instead of
one per layer per frame, sized for xGMI's per-link bound ring rather than for latency.
"""
import glob
import os
import pickle

import numpy as np
import torch
import torch.distributed as dist

from .config import cfg


def scale_lr(optimizer, scale):
    """lib/model/train_val.py:48-51."""
    for group in optimizer.param_groups:
        group['lr'] *= scale


def sgd_param_groups(net):
    """lib/model/train_val.py:188-208: one group per parameter; biases get ``lr * (DOUBLE_BIAS + 1)`` and no weight
    decay unless BIAS_DECAY."""
    lr = cfg.TRAIN.LEARNING_RATE
    groups = []
    for key, value in dict(net.named_parameters()).items():
        if not value.requires_grad:
            continue
        if 'bias' in key:
            groups.append({'params': [value], 'lr': lr * (cfg.TRAIN.DOUBLE_BIAS + 1),
                           'weight_decay': cfg.TRAIN.BIAS_DECAY and cfg.TRAIN.WEIGHT_DECAY or 0})
        else:
            groups.append({'params': [value], 'lr': lr,
                           'weight_decay': getattr(value, 'weight_decay', cfg.TRAIN.WEIGHT_DECAY)})
    return groups


class GradientBucket:
    """All trainable gradients of a network as views into one flat fp32 buffer resident in HBM."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        # one extra element behind the gradients: the FAULT slot.  A rank that hits a fatal frame adds 1 to it; the slot
        # rides in the same all-reduce, so every rank learns about it at the next optimizer step and all stop together
        self.flat = torch.zeros(total + 1, dtype=torch.float32, device=dev)
        self.fault = self.flat[total:]
        off = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("GradientBucket needs fp32 parameters on one device")
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None):
        """Average the accumulated gradients over the ranks (no-op for a single process).  Returns the number of faults
        the ranks recorded since the last ``zero()``."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            faults = float(self.fault.item())
            self.flat.div_(dist.get_world_size(group))
            return faults
        return 0.0          # single process: mark_fault() raises at once, the slot is never written (and no host sync here)


class DataParallelOptimizer:
    """The optimizer object handed to ``Network.train_step(blobs, optimizer, update_weights)``: ``step()`` first
    averages the flat gradient bucket over the ranks, ``zero_grad()`` clears the bucket in place (the views stay)."""

    def __init__(self, optimizer, bucket, group=None):
        self.optimizer = optimizer
        self.bucket = bucket
        self.group = group
        self._reduced = False
        self._fault_text = None

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def reduce(self):
        """Average the accumulated gradients over the ranks NOW.  ``Network.train_step`` calls this before it clips, so
        the clip acts on the gradient of the whole N-GPU batch (the value that is stepped), as in the single-process
        reference, instead of on each rank's share."""
        if not self._reduced:
            faults = self.bucket.all_reduce_mean(self.group)
            self._reduced = True
            if faults > 0:
                raise RuntimeError("a rank reported a fatal training frame (%d in this batch): %s"
                                   % (int(round(faults)), self._fault_text or "see that rank's log"))

    def _distributed(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def mark_fault(self, text=None):
        """Record a fatal frame on this rank.  Returns True when the error is deferred to the next ``reduce()`` (data
        parallel: all ranks raise there together), False when the caller should raise right away (single process)."""
        if not self._distributed():
            return False
        self.bucket.fault.add_(1.0)
        self._fault_text = text
        return True

    def check_faults(self):
        """All ranks learn NOW whether any of them recorded a fatal frame since the last weight update (the fault slot
        otherwise only travels with the next gradient all-reduce): called before a snapshot is written and at the end of
        training, so that a fault in the trailing frames of a run cannot end in a snapshot 'as if nothing happened'."""
        if not self._distributed():
            return
        flag = self.bucket.fault.clone()
        dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=self.group)
        if float(flag.item()) > 0:
            raise RuntimeError("a rank reported a fatal training frame (%d since the last weight update): %s"
                               % (int(round(float(flag.item()))), self._fault_text or "see that rank's log"))

    def step(self):
        self.reduce()
        self.optimizer.step()
        self._reduced = False

    def zero_grad(self, set_to_none=False):
        self.bucket.zero()
        self._reduced = False

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, state):
        self.optimizer.load_state_dict(state)


def broadcast_parameters(net, src=0, group=None):
    """Make every replica start from rank ``src``'s weights and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    for t in list(net.parameters()) + list(net.buffers()):
        dist.broadcast(t.data, src=src, group=group)


class SolverWrapper:
    """lib/model/train_val.py:58-503 without the dataset / tensorboard plumbing."""

    def __init__(self, network, num_classes, frames, val_frames=None, output_dir='.', sum_size=128, val_sum_size=0,
                 epoch_size=0, batch_size=None, val_batch_size=None, data_parallel=None, log=print):
        self.net = network
        self.num_classes = num_classes
        self.data_gen = frames
        self.data_gen_val = val_frames
        self.output_dir = output_dir
        self.sum_size = sum_size
        self.val_sum_size = val_sum_size
        self.epoch_size = epoch_size
        self.batch_size = batch_size or cfg.TRAIN.BATCH_SIZE
        self.val_batch_size = val_batch_size or cfg.TRAIN.VAL_BATCH_SIZE
        self.log = log
        if data_parallel is None:
            data_parallel = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.data_parallel = data_parallel
        self.rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        self.optimizer = None
        self.summaries = []          # (iter, name, value) instead of tensorboard events
        self.val_summaries = []

    # -- graph / optimizer -------------------------------------------------------------------------------------
    def construct_graph(self):
        """train_val.py:167-213.  The architecture is created here unless the caller already did."""
        torch.manual_seed(cfg.RNG_SEED)
        if not list(self.net.parameters()):
            if cfg.NET_TYPE == 'lidar':
                self.net.create_architecture(self.num_classes, tag='default', anchor_scales=cfg.LIDAR.ANCHOR_SCALES,
                                             anchor_ratios=cfg.LIDAR.ANCHOR_ANGLES)
            else:
                self.net.create_architecture(self.num_classes, tag='default', anchor_scales=cfg.ANCHOR_SCALES,
                                             anchor_ratios=cfg.ANCHOR_RATIOS)
        self.net.to(self.net._device)
        if self.data_parallel:
            broadcast_parameters(self.net)
        sgd = torch.optim.SGD(sgd_param_groups(self.net), momentum=cfg.TRAIN.MOMENTUM)
        self.bucket = GradientBucket([p for g in sgd.param_groups for p in g['params']])
        self.optimizer = DataParallelOptimizer(sgd, self.bucket)
        return cfg.TRAIN.LEARNING_RATE, self.optimizer

    # -- snapshots ---------------------------------------------------------------------------------------------
    def _snapshot_name(self, it, ext):
        return os.path.join(self.output_dir, '%s_%s_iter_%d%s' % (cfg.NET_TYPE, cfg.TRAIN.SNAPSHOT_PREFIX, it, ext))

    def _dist(self):
        return self.data_parallel and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _sampler_state(self):
        cur, perm = self.data_gen.get_pointer() if hasattr(self.data_gen, 'get_pointer') else (0, None)
        cur_val, perm_val = (self.data_gen_val.get_pointer() if hasattr(self.data_gen_val, 'get_pointer')
                             else (0, None))
        return (np.random.get_state(), cur, perm, cur_val, perm_val, int(getattr(self.net, '_uc_calls', 0)))

    def _sync_batchnorm_statistics(self):
        """Replicas see different frames, so BatchNorm running statistics (LiDAR detector, FIXED_BLOCKS == -1) drift
        apart; before a snapshot every rank takes the mean over the ranks, so the saved weights describe all of them and
        the replicas continue from identical buffers."""
        world = dist.get_world_size()
        for name, buf in self.net.named_buffers():
            if buf.dtype.is_floating_point and (name.endswith('running_mean') or name.endswith('running_var')):
                dist.all_reduce(buf.data, op=dist.ReduceOp.SUM)
                buf.data.div_(world)

    def snapshot(self, it):
        """train_val.py:100-129: weights as .pth, numpy RNG + sampler pointers + iteration as consecutive pickles
        (rank 0's, in the reference's order).  Data parallel: every rank contributes ITS sampler / RNG state (appended
        as a seventh object, a list over ranks), BatchNorm statistics are averaged first, rank 0 writes and everyone
        waits for the files."""
        sfile, nfile = self._snapshot_name(it, '.pth'), self._snapshot_name(it, '.pkl')
        if hasattr(self.optimizer, 'check_faults'):
            self.optimizer.check_faults()
        states = None
        if self._dist():
            self._sync_batchnorm_statistics()
            states = [None] * dist.get_world_size()
            dist.all_gather_object(states, self._sampler_state())
        if self.rank == 0:
            os.makedirs(self.output_dir, exist_ok=True)
            torch.save(self.net.state_dict(), sfile)
            mine = self._sampler_state()
            st0, cur, perm, cur_val, perm_val = mine[:5]
            with open(nfile, 'wb') as fid:
                for obj in (st0, cur, perm, cur_val, perm_val, it):
                    pickle.dump(obj, fid, pickle.HIGHEST_PROTOCOL)
                # seventh object (not in the reference's file): per-rank state incl. the uncertainty heads' draw counter
                pickle.dump(states if states is not None else [mine], fid, pickle.HIGHEST_PROTOCOL)
            self.log('Wrote snapshot to: %s' % sfile)
        if self._dist():
            dist.barrier()
        return sfile, nfile

    def from_snapshot(self, sfile, nfile):
        """train_val.py:131-150.  With a per-rank state list in the file every rank resumes ITS OWN sampler position and
        numpy RNG (restoring rank 0's everywhere would make the N-GPU batch N copies of one frame)."""
        self.net.load_state_dict(torch.load(str(sfile), map_location=self.net._device))
        with open(nfile, 'rb') as fid:
            st0, cur, perm, cur_val, perm_val, last = [pickle.load(fid) for _ in range(6)]
            try:
                states = pickle.load(fid)
            except EOFError:
                states = None
        if states is not None and self.rank < len(states):
            st0, cur, perm, cur_val, perm_val = states[self.rank][:5]
            if len(states[self.rank]) > 5 and hasattr(self.net, '_uc_calls'):
                self.net._uc_calls = int(states[self.rank][5])      # the masks continue where the snapshot left them
        elif self._dist() and self.rank > 0:
            # a single-process snapshot resumed on several GPUs: decorrelate the replicas deterministically
            np.random.seed((cfg.RNG_SEED + 7919 * self.rank) % (2 ** 32))
            st0 = np.random.get_state()
        np.random.set_state(st0)
        if hasattr(self.data_gen, 'set_pointer'):
            self.data_gen.set_pointer(cur, perm)
        if hasattr(self.data_gen_val, 'set_pointer'):
            self.data_gen_val.set_pointer(cur_val, perm_val)
        return last

    def find_previous(self):
        """train_val.py:215-241: snapshots by age, without the ones written just before a learning-rate drop."""
        red = [self._snapshot_name(s + 1, '.pth') for s in cfg.TRAIN.STEPSIZE]
        sfiles = sorted(glob.glob(self._snapshot_name(0, '.pth').replace('_iter_0.pth', '_iter_*.pth')),
                        key=os.path.getmtime)
        sfiles = [s for s in sfiles if s not in red]
        nfiles = sorted(glob.glob(self._snapshot_name(0, '.pkl').replace('_iter_0.pkl', '_iter_*.pkl')),
                        key=os.path.getmtime)
        nfiles = [n for n in nfiles if n not in [r.replace('.pth', '.pkl') for r in red]]
        assert len(nfiles) == len(sfiles)
        return len(sfiles), nfiles, sfiles

    def initialize(self):
        """train_val.py:243-262: a fresh run starts from ``pretrained_model`` when cfg.PRELOAD (backbone only) or
        cfg.PRELOAD_FULL is set, else from the seeded initialisation."""
        pretrained = getattr(self, 'pretrained_model', None)
        if pretrained is not None and cfg.PRELOAD:
            self.net.load_pretrained_cnn(torch.load(pretrained, map_location=self.net._device))
        elif pretrained is not None and cfg.PRELOAD_FULL:
            self.net.load_pretrained_full(torch.load(pretrained, map_location=self.net._device))
        return cfg.TRAIN.LEARNING_RATE, 0, list(cfg.TRAIN.STEPSIZE), [], []

    def restore(self, sfile, nfile):
        """train_val.py:264-278."""
        last = self.from_snapshot(sfile, nfile)
        lr_scale, stepsizes = 1.0, []
        for stepsize in cfg.TRAIN.STEPSIZE:
            if last > stepsize:
                lr_scale *= cfg.TRAIN.GAMMA
            else:
                stepsizes.append(stepsize)
        scale_lr(self.optimizer, lr_scale)
        return cfg.TRAIN.LEARNING_RATE * lr_scale, last, stepsizes, [nfile], [sfile]

    def remove_snapshot(self, np_paths, ss_paths):
        """train_val.py:280-294."""
        for paths in (np_paths, ss_paths):
            while len(paths) > cfg.TRAIN.SNAPSHOT_KEPT:
                victim = paths.pop(0)
                if self.rank == 0 and os.path.exists(victim):
                    os.remove(str(victim))

    # -- the loop ----------------------------------------------------------------------------------------------
    def train_model(self, max_iters):
        """train_val.py:296-503.  Returns the per-iteration losses of this rank."""
        lr, _ = self.construct_graph()
        lsf, nfiles, sfiles = self.find_previous()
        if lsf == 0:
            lr, last_snapshot_iter, stepsizes, np_paths, ss_paths = self.initialize()
        else:
            lr, last_snapshot_iter, stepsizes, np_paths, ss_paths = self.restore(str(sfiles[-1]), str(nfiles[-1]))
        it = last_snapshot_iter + 1
        stepsizes.append(max_iters)
        stepsizes.reverse()
        next_stepsize = stepsizes.pop()
        self.net.train()
        self.optimizer.zero_grad()
        losses = []
        pipe, pending = None, []
        # cfg.TRAIN.GRAPHS (default on): every step is a replayed hipGraph (model/train_graph.py); cfg.TRAIN.FRAMES_IN_FLIGHT
        # (default 3) frames of a pseudo batch run concurrently as single-chain graphs.  A frame a graph cannot express
        # (don't-care boxes) runs eagerly, with a warning from Network.train_step.
        on_device = torch.device(self.net._device).type == 'cuda' and hasattr(self.net, 'enable_train_graphs')
        if on_device and cfg.TEST.get('FRAME_GRAPHS', True) and self.data_gen_val is not None and self.val_sum_size:
            # validation frames (run_eval -> test_frame, train_val.py:411-412) replay the captured forward pass too; the pool
            # notices every weight update through the parameters' version counters and re-derives the cached filters in place
            self.net.enable_frame_graphs(True)
        if on_device and cfg.TRAIN.get('GRAPHS', True):
            from . import train_graph
            self.optimizer.zero_grad(set_to_none=False)
            self.net.enable_train_graphs(True)
            if int(cfg.TRAIN.get('FRAMES_IN_FLIGHT', 1)) > 1:
                pipe = train_graph.TrainPipeline(self.net, slots=int(cfg.TRAIN.FRAMES_IN_FLIGHT))
        while it < max_iters + 1:
            update_weights = (it % self.batch_size == 0 and it != 0)
            if it == next_stepsize + 1:
                self._drain(pipe, pending, losses)
                self.snapshot(it)
                lr *= cfg.TRAIN.GAMMA
                scale_lr(self.optimizer, cfg.TRAIN.GAMMA)
                next_stepsize = stepsizes.pop()
            blobs = self.data_gen.next()
            if self.val_sum_size and self.data_gen_val is not None and it % self.val_sum_size == 0:
                self._drain(pipe, pending, losses)
                for i in range(self.val_batch_size):
                    out = self.net.run_eval(self.data_gen_val.next(), self.val_batch_size,
                                            i == self.val_batch_size - 1)
                    self.val_summaries += [(it, k, v) for k, v in out[0]]
            want_summary = bool(self.sum_size and it % self.sum_size == 0)
            if pipe is not None and train_graph.graphable(self.net, blobs) is None and self._pipeline_takes(pipe, blobs):
                # cfg.TRAIN.FRAMES_IN_FLIGHT > 1: the frame is queued on the next pipeline slot; its loss is collected when
                # the slot comes round again, or right away where this iteration needs it (summary, weight update)
                if pipe.in_flight() >= pipe.slots:
                    self._collect_one(pipe, pending, losses)
                pipe.submit(blobs)
                pending.append((it, want_summary))
                if update_weights or want_summary:
                    self._drain(pipe, pending, losses)
                if update_weights:
                    pipe.flush()
                    self.net.apply_update(self.optimizer, in_place=True)
                total_loss = None
            else:
                self._drain(pipe, pending, losses)
                if pipe is not None:
                    pipe.flush()                # an eager frame adds to param.grad directly: merge the slots first
                if want_summary:
                    total_loss, summary = self.net.train_step_with_summary(blobs, self.optimizer, self.sum_size,
                                                                           update_weights)
                    self.summaries += [(it, k, v) for k, v in summary]
                else:
                    total_loss = self.net.train_step(blobs, self.optimizer, update_weights)
                losses.append(total_loss)
            if self.epoch_size and it % self.epoch_size == 0:
                self._drain(pipe, pending, losses)
                self.log('epoch average loss: %f' % (sum(losses[-self.epoch_size:]) / self.epoch_size))
            if it % cfg.TRAIN.SNAPSHOT_ITERS == 0:
                self._drain(pipe, pending, losses)
                last_snapshot_iter = it
                ss_path, np_path = self.snapshot(it)
                np_paths.append(np_path)
                ss_paths.append(ss_path)
                if len(np_paths) > cfg.TRAIN.SNAPSHOT_KEPT:
                    self.remove_snapshot(np_paths, ss_paths)
            it += 1
        self._drain(pipe, pending, losses)
        if pipe is not None:
            pipe.flush()          # frames since the last update: their gradients leave the slots' buffers for param.grad
        if last_snapshot_iter != it - 1:
            self.snapshot(it - 1)
        elif hasattr(self.optimizer, 'check_faults'):
            self.optimizer.check_faults()
        return losses

    @staticmethod
    def _pipeline_takes(pipe, blobs):
        """Frames of a geometry beyond the pipeline's graph budget go through train_step (its own graph or the eager path)."""
        from . import train_graph
        data, info = blobs['data'], np.asarray(blobs['info'], dtype=np.float32)
        key = (int(data.shape[1]), int(data.shape[2]), int(data.shape[3]), train_graph.gt_capacity(len(blobs['gt_boxes'])),
               tuple(float(v) for v in info), train_graph.capture_switches())
        slot = pipe.runners[pipe.next_slot]
        return key in slot or len(slot) < pipe.max_graphs

    def _collect_one(self, pipe, pending, losses):
        it, want_summary = pending.pop(0)
        loss, counts = pipe.collect()
        if counts is not None and counts[0] + counts[1] == 0:
            from ..nets.network import NO_CANDIDATES
            if not (hasattr(self.optimizer, 'mark_fault') and self.optimizer.mark_fault()):
                raise RuntimeError(NO_CANDIDATES)
        losses.append(loss)
        if want_summary:
            self.summaries += [(it, k, float(v.item())) for k, v in self.net._losses.items()]

    def _drain(self, pipe, pending, losses):
        while pipe is not None and pending:
            self._collect_one(pipe, pending, losses)


class _PointerFrames:
    """The two frame sources of ``train_net`` behind the iterator ``SolverWrapper`` pulls from."""

    def __init__(self, gen):
        self.gen = gen
        self.next = gen.next
        self.get_pointer = gen.get_pointer
        self.set_pointer = gen.set_pointer


def train_net(network, db, output_dir, tb_dir, pretrained_model=None, max_iters=40000, sum_size=128, val_sum_size=1000,
              batch_size=16, val_batch_size=16, val_thresh=0.1, augment_en=True, val_augment_en=False):
    """Reference entry point (lib/model/train_val.py:532-569; called at tools/trainval_net.py:~340): ``db.roidb`` /
    ``db.val_roidb`` lists of roidb entries ('filename', 'boxes', 'gt_classes', 'ignore', optional 'boxes_dc'),
    ``db.num_classes``.  ``tb_dir`` is accepted and unused (summaries are kept as (iter, name, value) tuples on the
    solver instead of tensorboard events).  cfg.PRELOAD / cfg.PRELOAD_FULL load ``pretrained_model`` through the
    detector's ``load_pretrained_cnn`` / ``load_pretrained_full`` like :249-253.  Returns the solver (losses in
    ``solver.losses``)."""
    from .data_layer_generator import data_layer_generator
    frames = _PointerFrames(data_layer_generator('train', db.roidb, augment_en, db.num_classes))
    val_roidb = getattr(db, 'val_roidb', None)
    val_frames = (_PointerFrames(data_layer_generator('val', val_roidb, val_augment_en, db.num_classes))
                  if val_roidb else None)
    sw = SolverWrapper(network, db.num_classes, frames, val_frames, output_dir=output_dir, sum_size=sum_size,
                       val_sum_size=val_sum_size if val_frames is not None else 0, epoch_size=len(db.roidb),
                       batch_size=batch_size, val_batch_size=val_batch_size)
    sw.val_thresh = val_thresh
    if pretrained_model is not None and (cfg.PRELOAD or cfg.PRELOAD_FULL):
        sw.pretrained_model = pretrained_model
    print('Solving...')
    sw.losses = sw.train_model(max_iters)
    print('done solving')
    return sw
