"""N > 1 path on CPU: world_size-2 gloo run of the eval collate (model/collate.py) that bench.py and the
8-GPU eval use with RCCL.  Each rank fabricates the per-class detections of its frames; after the
all-gather every rank must rebuild the same all_boxes, equal to what a single process computes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from faster_rcnn_pytorch_multimodal_amd.model import collate

K, MAX_OUT, FRAMES = 3, 10, 5


def _fake_frame(i):
    rng = np.random.default_rng(100 + i)
    dets = np.zeros((K, MAX_OUT, 5), np.float32)
    counts = np.zeros((K,), np.int32)
    for j in range(1, K):
        counts[j] = rng.integers(0, MAX_OUT + 1)
        dets[j, :counts[j]] = rng.random((counts[j], 5), dtype=np.float32) * 600
    return dets, counts


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    all_frames = {}
    mine = collate.shard_frames(FRAMES, rank, world)
    steps = (FRAMES + world - 1) // world
    for s in range(steps):
        if s < len(mine):
            dets, counts = _fake_frame(mine[s])
        else:  # ragged tail: this rank has no frame left, contributes an empty record
            dets, counts = np.zeros((K, MAX_OUT, 5), np.float32), np.zeros((K,), np.int32)
        rec = collate.pack_record(torch.from_numpy(dets), torch.from_numpy(counts))
        gathered = collate.gather_records(rec)
        for r, per_class in enumerate(collate.unpack_records(gathered, K, MAX_OUT)):
            frame = s * world + r
            if frame < FRAMES:
                all_frames[frame] = per_class
    torch.save(all_frames, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_collate_world2_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(2)]
    for frame in range(FRAMES):
        dets, counts = _fake_frame(frame)
        for j in range(1, K):
            for r in range(2):   # every rank holds the full collate
                np.testing.assert_array_equal(got[r][frame][j], dets[j, :counts[j]])
            assert got[0][frame][0].shape == (0, 5)


def test_shard_frames_partition():
    for world in (1, 2, 3, 8):
        seen = sorted(i for r in range(world) for i in collate.shard_frames(11, r, world))
        assert seen == list(range(11))


class _FakeDb:
    """Frame source for model.test.test_net: 5 frames, frame 3 has no data (the reference skips such frames)."""
    num_classes = K

    def num_frames(self, mode):
        return FRAMES

    def blobs_at(self, i, mode):
        return {"data": None if i == 3 else np.full((1, 4, 4, 3), float(i), np.float32),
                "info": np.array([0, 4, 0, 4, 0, 0, 1.0], np.float32)}

    def name_at(self, i, mode):
        return "frame%02d" % i


def _fake_detect(net, data, info, thresh, max_dets, max_out):
    dets, counts = _fake_frame(int(data[0, 0, 0, 0]))
    # test_net asks for one row per RoI (max_out >= max_dets) so that ties at the max_dets-th score survive the record
    assert max_out >= max_dets
    padded = np.zeros((dets.shape[0], max_out, dets.shape[2]), np.float32)
    padded[:, :dets.shape[1]] = dets
    return torch.from_numpy(padded), torch.from_numpy(counts)


def _test_net_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from faster_rcnn_pytorch_multimodal_amd.model import config as C
    from faster_rcnn_pytorch_multimodal_amd.model import test as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    T.detect_frame_device = _fake_detect               # the device path is covered by the -m gpu tests

    class Net:
        _device = "cpu"

    all_boxes = T.test_net(Net(), _FakeDb(), os.path.join(out_dir, "eval"), max_dets=MAX_OUT, thresh=0.5)
    torch.save(all_boxes, os.path.join(out_dir, "boxes%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_test_net_eval_loop_world2_gloo(tmp_path):
    """model.test.test_net (lib/model/test.py:138-257) sharded over two ranks: every rank ends with the complete
    all_boxes, the skipped frame stays empty, rank 0 writes detections.pkl and the per-class text files."""
    import pickle
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_test_net_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [torch.load(os.path.join(str(tmp_path), "boxes%d.pt" % r), weights_only=False) for r in range(2)]
    for frame in range(FRAMES):
        dets, counts = _fake_frame(frame)
        for j in range(1, K):
            for r in range(2):
                if frame == 3 or counts[j] == 0:
                    assert got[r][j][frame].size == 0
                else:
                    np.testing.assert_array_equal(got[r][j][frame], dets[j, :counts[j]])
    with open(tmp_path / "eval" / "detections.pkl", "rb") as f:
        pk = pickle.load(f)
    assert len(pk) == K and len(pk[1]) == FRAMES
    lines = open(tmp_path / "eval" / "det_test_cls1.txt").read().splitlines()
    assert len(lines) == sum(int(_fake_frame(i)[1][1]) for i in range(FRAMES) if i != 3)
    assert all(l.split(" ")[1].startswith("frame") for l in lines)


def _ring_worker(rank, world, port, out_dir, every):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    numel, steps = 7, 11
    ring = collate.RecordRing(numel, steps, every=every, device="cpu")
    for i in range(steps):
        ring.slot(i).copy_(torch.arange(numel, dtype=torch.float32) + 100.0 * (rank + world * i))   # global frame id
        ring.commit(i)
    host = ring.drain().clone()
    torch.save((host, ring.gathers), os.path.join(out_dir, "ring%d_%d.pt" % (every, rank)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("every", [1, 4])
def test_record_ring_collates_blocks_in_frame_order(tmp_path, every):
    """RecordRing over gloo, world 2: one all-gather per block of `every` frames (the partial last block too), and the host
    matrix reads step i, row r = global frame r + 2*i - the order lib/model/test.py:183's loop visits the frames in."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_ring_worker, args=(2, port, str(tmp_path), every), nprocs=2, join=True)
    got = [torch.load(os.path.join(str(tmp_path), "ring%d_%d.pt" % (every, r))) for r in range(2)]
    assert torch.equal(got[0][0], got[1][0]) and got[0][0].shape == (11, 2, 7)
    assert got[0][1] == -(-11 // every)
    frame_ids = (got[0][0][:, :, 0] / 100.0).round().long()
    assert frame_ids.reshape(-1).tolist() == list(range(22))
