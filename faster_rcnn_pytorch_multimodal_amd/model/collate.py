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
