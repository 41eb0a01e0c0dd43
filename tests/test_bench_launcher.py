"""bench.py's N > 1 control flow without a GPU: `python bench.py --gpus 2` must start its own ranks (no external
torch.distributed.run), exchange one record per rank per step, verify what the gathers delivered and print ONE JSON
line from rank 0.  `--rehearse-collate` swaps the device work for synthetic records and RCCL for gloo; everything else
(launcher, sharding, step loop, all-gather, barrier + max-over-ranks timing, verification, report) is the code the
8-GPU run executes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@pytest.mark.parametrize("gather_every", [1, 4])
def test_gpus_2_self_launches_and_collates_over_gloo(gather_every):
    """Records are collated in blocks of `gather_every` frames (one all-gather per block, a partial last block included);
    the verification inside bench.py requires step i, row r of what arrived on the host to be rank r's record of ITS
    i-th frame - collated order = frame order of the sharded loop - for both block sizes."""
    res = _run(["--gpus", "2", "--steps", "6", "--warmup", "2", "--rehearse-collate", "--gather-every", str(gather_every)])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout                      # rank 0 only
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "verification", "collective"):
        assert key in out, key                                # the driver's contract (device-only objects aside)
    assert out["metric"].startswith("frames/sec res101 Faster-RCNN") and out["unit"] == "frames/s"
    assert out["higher_is_better"] is True and out["vs_baseline"] is None and out["dtype"] == "f32"
    assert "workload" in out["config"] and "model" not in out["config"] and out["config"]["frames_per_step"] == 2
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["value"] is None and "rehearsal" in out["data"]          # never mistaken for a measurement
    col = out["collective"]
    assert col["ranks"] == 2 and col["backend"] == "gloo" and col["is_rccl"] is False and col["self_launched"] is True
    assert len(col["per_rank_frames_per_s"]) == 2
    assert col["gather_every_frames"] == gather_every and col["allgathers_per_region"] == -(-6 // gather_every)
    ver = out["verification"]
    assert ver["equal_to_eager_path"] is True and ver["timed_steps_checked"] == 6
    assert ver["distinct_expected_records"] == ver["distinct_frames_per_rank"] >= 2


def test_gpus_8_rehearsal_frame_sharding_blocks_and_device_binding():
    """The 8-rank run of BASELINE.json configs[4] rehearsed over gloo (the driver's `--gpus 8` can then only fail on RCCL
    itself): frame seed r + 8 j on rank r (seeds 0..7 in the first step), ONE all-gather per block of 8 frames, every rank
    bound to cuda:LOCAL_RANK, the audit fields of the `collective` object, all records verified in collated order."""
    res = _run(["--gpus", "8", "--steps", "16", "--warmup", "2", "--rehearse-collate"], timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    col = out["collective"]
    assert out["n_gpus"] == 8 and out["config"]["frames_per_step"] == 8 and out["scaling"] == "weak"
    assert col["ranks"] == 8 and col["self_launched"] is True and len(col["per_rank_frames_per_s"]) == 8
    assert col["gather_every_frames"] == 8 and col["allgathers_per_region"] == 2          # 16 steps = two blocks of 8 frames
    assert col["rank_devices"] == ["cuda:%d" % r for r in range(8)]
    assert [seeds[0] for seeds in col["rank_frame_seeds"]] == list(range(8))              # frame r of the step on rank r
    assert all(seeds[1] == seeds[0] + 8 for seeds in col["rank_frame_seeds"])
    assert col["record_bytes_per_rank"] * 8 == col["block_bytes_per_rank"]
    ver = out["verification"]
    assert ver["equal_to_eager_path"] is True and ver["timed_steps_checked"] == 16 and ver["regions_checked"] >= 1


def test_device_binding_rule():
    sys.path.insert(0, ROOT)
    import bench
    assert [bench.device_for_rank(r, "nccl", 8) for r in range(8)] == list(range(8))
    assert [bench.device_for_rank(r, "nccl", 1) for r in range(2)] == [0, 1]      # RCCL never shares a device between ranks
    assert [bench.device_for_rank(r, "gloo", 1) for r in range(3)] == [0, 0, 0]   # the rehearsal wraps around


def test_world_size_mismatch_is_rejected():
    res = _run(["--gpus", "2", "--rehearse-collate"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert res.returncode != 0 and "WORLD_SIZE" in (res.stderr + res.stdout)


def test_bench_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    res = _run(["--steps", "1", "--warmup", "0"])
    assert res.returncode != 0 and "MI355X" in (res.stderr + res.stdout)
