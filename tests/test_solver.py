"""Solver loop (model/train_val.py) on CPU: the schedule of lib/model/train_val.py:296-503 with a stub network, the
snapshot round trip, and the data-parallel gradient bucket on a world_size-2 gloo group (the N > 1 training path the
GPU job runs over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from faster_rcnn_pytorch_multimodal_amd.model import config as C
from faster_rcnn_pytorch_multimodal_amd.model import train_val


class _StubNet(torch.nn.Module):
    """Network protocol subset the solver touches; loss = sum((w*x - y)^2) over the frame."""

    def __init__(self):
        super().__init__()
        self._device = "cpu"
        self.lin = torch.nn.Linear(4, 2)
        self.calls = []

    def train_step(self, blobs, optimizer, update_weights=False):
        loss = ((self.lin(blobs["data"]) - blobs["y"]) ** 2).sum()
        loss.backward()
        self.calls.append((update_weights, optimizer.param_groups[0]["lr"]))
        if update_weights:
            optimizer.step()
            optimizer.zero_grad()
        return float(loss.item())

    def train_step_with_summary(self, blobs, optimizer, sum_size, update_weights=False):
        return self.train_step(blobs, optimizer, update_weights), [("total_loss", 0.0)]


class _Frames:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.cur = 0

    def next(self):
        self.cur += 1
        return {"data": torch.from_numpy(self.rng.standard_normal((3, 4)).astype(np.float32)),
                "y": torch.from_numpy(self.rng.standard_normal((3, 2)).astype(np.float32))}

    def get_pointer(self):
        return self.cur, None

    def set_pointer(self, cur, perm):
        self.cur = cur


@pytest.fixture
def cfg_solver():
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    C.cfg.TRAIN.STEPSIZE = [6, 12]
    C.cfg.TRAIN.SNAPSHOT_ITERS = 5
    C.cfg.TRAIN.SNAPSHOT_KEPT = 2
    yield C.cfg
    C.reset_cfg()


def test_param_groups_follow_reference_rules(cfg_solver):
    net = _StubNet()
    cfg_solver.TRAIN.DOUBLE_BIAS = True
    groups = train_val.sgd_param_groups(net)
    by_name = dict(zip([k for k, _ in net.named_parameters()], groups))
    assert by_name["lin.bias"]["lr"] == 2 * cfg_solver.TRAIN.LEARNING_RATE and by_name["lin.bias"]["weight_decay"] == 0
    assert by_name["lin.weight"]["lr"] == cfg_solver.TRAIN.LEARNING_RATE
    assert by_name["lin.weight"]["weight_decay"] == cfg_solver.TRAIN.WEIGHT_DECAY


def test_schedule_snapshots_and_resume(cfg_solver, tmp_path):
    net = _StubNet()
    solver = train_val.SolverWrapper(net, 2, _Frames(0), output_dir=str(tmp_path), batch_size=4, sum_size=0,
                                     log=lambda *_: None)
    losses = solver.train_model(14)
    assert len(losses) == 14
    # the optimizer steps on iterations 4, 8, 12 (train_val.py:379-382)
    assert [i + 1 for i, (u, _) in enumerate(net.calls) if u] == [4, 8, 12]
    # learning rate drops by GAMMA at stepsize + 1 = 7 and 13 (train_val.py:383-389)
    lrs = [lr for _, lr in net.calls]
    base = cfg_solver.TRAIN.LEARNING_RATE
    assert np.allclose(lrs[:6], base) and np.allclose(lrs[6:12], base * 0.1) and np.allclose(lrs[12:], base * 0.01)
    names = sorted(os.path.basename(p) for p in tmp_path.glob("*.pth"))
    # 5 and 10 periodic, 7 and 13 before the drops, 14 at the end; SNAPSHOT_KEPT only counts the periodic ones
    assert names == sorted("image_res101_faster_rcnn_iter_%d.pth" % i for i in (5, 7, 10, 13, 14))
    # resume: a new solver finds iteration 14, restores weights + sampler pointer and has nothing left to do
    net2 = _StubNet()
    frames2 = _Frames(0)
    solver2 = train_val.SolverWrapper(net2, 2, frames2, output_dir=str(tmp_path), batch_size=4, sum_size=0,
                                      log=lambda *_: None)
    assert solver2.train_model(14) == []
    assert frames2.cur == 14
    for a, b in zip(net.state_dict().values(), net2.state_dict().values()):
        assert torch.equal(a, b)
    assert np.isclose(solver2.optimizer.param_groups[0]["lr"], base * 0.01)


def test_gradient_bucket_views_survive_zero_grad():
    net = _StubNet()
    sgd = torch.optim.SGD(net.parameters(), lr=0.1)
    bucket = train_val.GradientBucket(net.parameters())
    opt = train_val.DataParallelOptimizer(sgd, bucket)
    net.train_step(_Frames(1).next(), opt, update_weights=False)
    assert bucket.flat.abs().sum() > 0
    assert net.lin.weight.grad.data_ptr() == bucket.flat.data_ptr()
    opt.zero_grad()
    assert bucket.flat.abs().sum() == 0 and net.lin.weight.grad.data_ptr() == bucket.flat.data_ptr()


def _dp_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    C.cfg.TRAIN.SNAPSHOT_ITERS = 1000
    torch.manual_seed(10 + rank)                  # replicas start different: construct_graph must broadcast rank 0
    net = _StubNet()
    solver = train_val.SolverWrapper(net, 2, _Frames(100 + rank), output_dir=os.path.join(out_dir, "snap"),
                                     batch_size=2, sum_size=0, log=lambda *_: None)
    solver.train_model(4)
    torch.save({k: v.clone() for k, v in net.state_dict().items()}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_data_parallel_step_matches_single_process_average(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0 = torch.load(tmp_path / "rank0.pt")
    w1 = torch.load(tmp_path / "rank1.pt")
    for k in w0:
        assert torch.equal(w0[k], w1[k])          # replicas stay identical
    # single-process restatement: start from rank 0's weights, average the two ranks' accumulated gradients
    C.reset_cfg()
    torch.manual_seed(10)
    ref = _StubNet()
    sgd = torch.optim.SGD(train_val.sgd_param_groups(ref), momentum=C.cfg.TRAIN.MOMENTUM)
    frames = [_Frames(100), _Frames(101)]
    for it in range(1, 5):
        for f in frames:
            blobs = f.next()
            (((ref.lin(blobs["data"]) - blobs["y"]) ** 2).sum() / 2).backward()
        if it % 2 == 0:
            sgd.step()
            sgd.zero_grad()
    for k, v in ref.state_dict().items():
        assert torch.allclose(v, w0[k], atol=1e-6), k


class _BnStubNet(_StubNet):
    """Adds a BatchNorm whose running statistics drift per rank, and the clip-after-reduce protocol of Network.train_step."""

    def __init__(self):
        super().__init__()
        self.bn = torch.nn.BatchNorm1d(4)

    def train_step(self, blobs, optimizer, update_weights=False):
        loss = ((self.lin(self.bn(blobs["data"])) - blobs["y"]) ** 2).sum()
        loss.backward()
        if update_weights:
            optimizer.reduce()
            self.clipped_input = self.lin.weight.grad.clone()        # what the clip sees = the averaged gradient
            self.lin.weight.grad.clamp_(-0.5, 0.5)
            optimizer.step()
            optimizer.zero_grad()
        return float(loss.item())


def _dp_resume_worker(rank, world, port, out_dir):
    import pickle
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    C.cfg.TRAIN.SNAPSHOT_ITERS = 3
    C.cfg.TRAIN.STEPSIZE = [1000]
    snap = os.path.join(out_dir, "snap")
    torch.manual_seed(3 + rank)
    net = _BnStubNet()
    net.train()
    frames = _Frames(200 + rank)
    frames.cur = 10 * rank                       # ranks sit at different sampler positions
    np.random.seed(500 + rank)
    solver = train_val.SolverWrapper(net, 2, frames, output_dir=snap, batch_size=2, sum_size=0, log=lambda *_: None)
    solver.train_model(3)
    np_state_at_snapshot = np.random.get_state()[1][:4].tolist()
    # a fresh process group member resumes: its OWN pointer and numpy state come back, not rank 0's
    net2 = _BnStubNet()
    frames2 = _Frames(0)
    np.random.seed(0)
    solver2 = train_val.SolverWrapper(net2, 2, frames2, output_dir=snap, batch_size=2, sum_size=0, log=lambda *_: None)
    solver2.construct_graph()
    _, nfiles, sfiles = solver2.find_previous()
    last = solver2.from_snapshot(sfiles[-1], nfiles[-1])
    out = {"rank": rank, "last": last, "cur_before": 10 * rank + 3, "cur_after": frames2.cur,
           "np_after": np.random.get_state()[1][:4].tolist(), "np_at_snapshot": np_state_at_snapshot,
           "bn_mean": net.bn.running_mean.clone(), "bn_mean_loaded": net2.bn.running_mean.clone(),
           "clipped_input": net.clipped_input, "weight": net.lin.weight.detach().clone()}
    with open(os.path.join(out_dir, "resume%d.pkl" % rank), "wb") as f:
        pickle.dump(out, f)
    dist.destroy_process_group()


def test_data_parallel_resume_restores_each_ranks_own_sampler_state(tmp_path):
    """a resume used to put rank 0's sampler / RNG state on every rank (the N-GPU batch became N
    copies of one frame), nothing waited for the snapshot, BatchNorm statistics diverged, and the gradient clip ran
    before the all-reduce."""
    import pickle
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_resume_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [pickle.load(open(tmp_path / ("resume%d.pkl" % i), "rb")) for i in range(2)]
    for x in r:
        assert x["last"] == 3 and x["cur_after"] == x["cur_before"]          # own pointer
        assert x["np_after"] == x["np_at_snapshot"]                           # own numpy RNG stream
    assert r[0]["cur_after"] != r[1]["cur_after"] and r[0]["np_after"] != r[1]["np_after"]
    # BatchNorm statistics were averaged before saving: identical on both ranks and equal to what the file holds
    assert torch.equal(r[0]["bn_mean"], r[1]["bn_mean"]) and torch.equal(r[0]["bn_mean"], r[0]["bn_mean_loaded"])
    # the clip acted on the averaged gradient: both ranks saw the same tensor and stepped to the same weights
    assert torch.equal(r[0]["clipped_input"], r[1]["clipped_input"]) and torch.equal(r[0]["weight"], r[1]["weight"])


class _FaultyStubNet(_StubNet):
    """Network.train_step's fatal-frame protocol: the frame is found bad AFTER its backward pass; single process raises at
    once, data parallel records the fault in the gradient bucket and every rank raises at the next all-reduce."""

    def __init__(self, bad_iteration):
        super().__init__()
        self.bad_iteration, self.it = bad_iteration, 0
        self._uc_calls = 0

    def train_step(self, blobs, optimizer, update_weights=False):
        self.it += 1
        self._uc_calls += 1
        loss = ((self.lin(blobs["data"]) - blobs["y"]) ** 2).sum()
        loss.backward()
        if self.it == self.bad_iteration:
            if not (hasattr(optimizer, "mark_fault") and optimizer.mark_fault("no candidates")):
                raise RuntimeError("no candidates")
        if update_weights:
            optimizer.reduce()
            optimizer.step()
            optimizer.zero_grad()
        return float(loss.item())


def _dp_fault_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    C.cfg.TRAIN.SNAPSHOT_ITERS = 1000
    net = _FaultyStubNet(bad_iteration=3 if rank == 1 else -1)        # only rank 1 meets the bad frame
    solver = train_val.SolverWrapper(net, 2, _Frames(300 + rank), output_dir=os.path.join(out_dir, "snap"),
                                     batch_size=2, sum_size=0, log=lambda *_: None)
    try:
        solver.train_model(8)
        outcome = "finished"
    except RuntimeError as err:
        outcome = "raised at iteration %d: %s" % (net.it, err)
    with open(os.path.join(out_dir, "fault%d.txt" % rank), "w") as f:
        f.write(outcome)
    dist.destroy_process_group()


def test_data_parallel_fatal_frame_stops_every_rank_together(tmp_path):
    """a frame without candidate RoIs raised on ONE rank and left the others blocked in the gradient
    all-reduce.  Now the fault travels in the bucket: both ranks raise at the optimizer step that follows (iteration 4)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_fault_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [open(tmp_path / ("fault%d.txt" % r)).read() for r in range(2)]
    assert all(g.startswith("raised at iteration 4: a rank reported a fatal training frame (1 in this batch)") for g in got), got
    # single process: the same frame raises immediately
    C.reset_cfg()
    C.cfg.NET_TYPE = "image"
    net = _FaultyStubNet(bad_iteration=3)
    solver = train_val.SolverWrapper(net, 2, _Frames(1), output_dir=str(tmp_path / "single"), batch_size=2, sum_size=0,
                                     log=lambda *_: None)
    with pytest.raises(RuntimeError, match="no candidates"):
        solver.train_model(8)
    assert net.it == 3
    C.reset_cfg()


def test_snapshot_carries_the_uncertainty_draw_counter(cfg_solver, tmp_path):
    """the counter behind the uncertainty heads' random draws (Network._uc_calls) is part of the per-rank
    snapshot state, so a resumed run continues the mask sequence instead of replaying it from step 0."""
    net = _FaultyStubNet(bad_iteration=-1)
    solver = train_val.SolverWrapper(net, 2, _Frames(0), output_dir=str(tmp_path), batch_size=4, sum_size=0, log=lambda *_: None)
    solver.train_model(5)
    assert net._uc_calls == 5
    net2 = _FaultyStubNet(bad_iteration=-1)
    solver2 = train_val.SolverWrapper(net2, 2, _Frames(0), output_dir=str(tmp_path), batch_size=4, sum_size=0,
                                      log=lambda *_: None)
    assert solver2.train_model(5) == [] and net2._uc_calls == 5
