"""Eval collate across GPUs: one frame per rank, one all-gather of a fixed-size detection record.

The reference's eval loop (lib/model/test.py:183-257) is a single process that appends each frame's
``all_boxes[cls][frame]``; frames are independent, so here rank r runs frames r, r+N, ... and a step ends
with ONE ``all_gather_into_tensor`` (RCCL over xGMI on the GPUs, gloo in the CPU tests) of

    record = [ dets (K, max_dets, 5) fp32 flattened | counts (K,) as fp32 ]       (~4 KB for K = 2)

Counts up to 2**24 are exact in fp32, so one dtype and one collective suffice.  Rank 0 (or every rank)
rebuilds the reference's ``all_boxes[cls][frame]`` lists from the gathered matrix.
"""
import numpy as np
import torch

EVAL_GATHER_EVERY = 8      # frames per collate block of the eval loop (model/test.test_net, bench.py --gather-every)


def record_numel(num_classes, max_out, elem=5):
    """elem = 5 image rows [x1,y1,x2,y2,score], 8 LiDAR rows [xc,yc,zc,l,w,h,ry,score]."""
    return num_classes * max_out * elem + num_classes


def pack_record(dets, counts, out=None):
    """dets (K, max_out, E) fp32, counts (K,) int -> (K*max_out*E + K,) fp32 record (device-side, async)."""
    k, m, e = dets.shape
    if out is None:
        out = torch.empty(record_numel(k, m, e), dtype=torch.float32, device=dets.device)
    out[:k * m * e].copy_(dets.reshape(-1))
    out[k * m * e:].copy_(counts)
    return out


def gather_records(record, gathered=None, group=None):
    """All ranks contribute `record`; returns the (world, numel) matrix (same on every rank)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if gathered is None:
        gathered = torch.empty((world, record.numel()), dtype=record.dtype, device=record.device)
    dist.all_gather_into_tensor(gathered.view(-1), record.view(-1), group=group)  # flat: rank-major concat
    return gathered


def unpack_records(gathered, num_classes, max_out, elem=5):
    """(world, numel) host/device matrix -> list over ranks of per-class (n_j, elem) float32 arrays, i.e. the
    ``all_boxes[cls][frame]`` entries of lib/model/test.py:228 for the frames of this step."""
    g = gathered.detach().cpu().numpy()
    frames = []
    split = num_classes * max_out * elem
    for r in range(g.shape[0]):
        dets = g[r, :split].reshape(num_classes, max_out, elem)
        counts = np.rint(g[r, split:]).astype(np.int64)
        frames.append([dets[j, :counts[j]].copy() if j > 0 else np.empty((0, elem), np.float32)
                       for j in range(num_classes)])
    return frames


def shard_frames(num_frames, rank, world):
    """Frame indices of this rank: i with i % world == rank (lib/model/test.py:183 loop, sharded)."""
    return list(range(rank, num_frames, world))


class RecordRing:
    """Per-frame records collected on the device, exchanged in blocks off the frames' streams.

    The eval loop's only cross-GPU step (lib/model/test.py:183-257 sharded one frame per rank) is the collation of the
    per-frame records.  One all-gather per frame issued from the frame's stream makes every step a rendezvous of all
    ranks and queues frame i+1's gather behind frame i's on the process group's single stream.  Here a frame's stream only
    WRITES its record into a slot of a device ring; every ``every`` frames a dedicated collate stream waits for those
    frames' events, issues ONE ``all_gather_into_tensor`` of the block (``every * numel`` floats per rank - exactly the
    bytes the per-frame gathers would have moved) and copies the collated block to pinned host memory.  Ranks therefore
    rendezvous once per block, on a stream no frame waits for; a frame only waits when its ring slot's previous
    contents have not been gathered yet (two blocks of slack).

    ``host`` after ``drain()``: (steps, ranks, numel) - step i, row r = rank r's record of its i-th frame, i.e. global frame
    ``r + world * i`` of the sharded loop: collated order = frame order.
    Works on CPU tensors too (gloo tests, bench.py --rehearse-collate): streams and events are then absent."""

    def __init__(self, numel, steps, every=8, device='cuda', group=None, distributed=None, gather_device=None,
                 pin=None, blocks=2):
        import torch.distributed as dist
        self.numel, self.steps, self.every, self.blocks = int(numel), int(steps), max(1, int(every)), max(2, int(blocks))
        self.device = torch.device(device)
        self.cuda = self.device.type == 'cuda'
        self.group = group
        if distributed is None:
            distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        self.distributed = bool(distributed)
        self.world = dist.get_world_size(group) if self.distributed else 1
        self.gather_device = torch.device(gather_device) if gather_device is not None else self.device
        self.ring = torch.zeros((self.blocks, self.every, self.numel), dtype=torch.float32, device=self.device)
        self.gathered = ([torch.zeros((self.world, self.every * self.numel), dtype=torch.float32,
                                      device=self.gather_device) for _ in range(self.blocks)] if self.distributed else None)
        nblk = (self.steps + self.every - 1) // self.every
        pin = self.cuda if pin is None else pin
        self._host_blocks = torch.empty((max(nblk, 1) * self.every, self.world, self.numel), dtype=torch.float32,
                                        pin_memory=bool(pin))
        self.stream = torch.cuda.Stream(device=self.device) if self.cuda else None
        self._free = [None] * self.blocks           # event: block b's slots have been read by its gather / copy
        self._pending = []                          # events of the frames written since the last flush
        self._written = 0
        self.gathers = 0

    def slot(self, i):
        """The (numel,) device view frame ``i`` of this rank packs its record into.  Call on the frame's stream, before
        the pack: the stream first waits until the slot's previous block has been gathered."""
        b = (i // self.every) % self.blocks
        if self.cuda and self._free[b] is not None:
            torch.cuda.current_stream(self.device).wait_event(self._free[b])
        return self.ring[b, i % self.every]

    def commit(self, i):
        """Frame ``i``'s record is queued on the current stream; frames must be committed in order.  Flushes a block when
        it is complete."""
        if i != self._written:
            raise RuntimeError("RecordRing: frames are committed in order (expected %d, got %d)" % (self._written, i))
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._pending.append(ev)
        self._written += 1
        if self._written % self.every == 0:
            self._flush(self._written - self.every)

    def _flush(self, first):
        import contextlib
        b = (first // self.every) % self.blocks
        block = self.ring[b]
        with (torch.cuda.stream(self.stream) if self.cuda else contextlib.nullcontext()):
            if self.cuda:
                for ev in self._pending:
                    self.stream.wait_event(ev)
            self._pending = []
            dst = self._host_blocks[first:first + self.every]                       # (every, world, numel)
            if self.distributed:
                src = block.view(-1) if block.device == self.gather_device else block.to(self.gather_device).view(-1)
                gather_records(src, self.gathered[b], self.group)
                self.gathers += 1
                dst.copy_(self.gathered[b].view(self.world, self.every, self.numel).permute(1, 0, 2), non_blocking=True)
            else:
                dst[:, 0].copy_(block, non_blocking=True)
            if self.cuda:
                ev = torch.cuda.Event()
                ev.record(self.stream)
                self._free[b] = ev

    def drain(self):
        """Flush the last partial block, wait for the collate stream, return host records (steps, ranks, numel).  A
        partial block is gathered whole (its unused slots are ignored), so every rank issues the same collectives."""
        done = self._written - self._written % self.every
        if self._written > done:
            self._flush(done)
        if self.cuda:
            self.stream.synchronize()
        return self._host_blocks[:self._written]

    def reset(self):
        """Start a new sequence of frames (after drain())."""
        self._written, self._pending = 0, []
