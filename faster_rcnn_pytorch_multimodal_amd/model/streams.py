"""HIP streams that really run side by side.

Frames in flight (model/frame_graph.FramePool, bench.py) and training steps in flight (model/train_graph.TrainPipeline) put
one hipGraph replay per stream and count on the streams overlapping.  A HIP stream is not a hardware queue: ROCclr multiplexes
a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues, the least-referenced queue at a stream's first use, and
two streams on one queue run strictly one after the other.  Which streams share a queue depends on how many streams the process
has used before (torch hands its pool out round robin; every FrameRunner / TrainStepRunner uses a warm-up stream of its own, the
library's throughput tuner three more): of the first twelve ``torch.cuda.Stream()`` of a fresh process, numbers 3 and 4 share a
queue, so do 5 and 2, 6 and 1 (tools/stream_concurrency_probe.py).  The same three-slot training pipeline therefore ran at 9.5 ms
per step inside bench.py and 11.4 ms in a process of its own - the "unexplained" two figures of rounds 4 and 5 - until its slots
were given streams chosen by measurement.

``concurrent_streams(n)`` measures instead of guessing: candidate streams each run a single-workgroup spin kernel
(``torch.cuda._sleep``) pairwise; a pair that takes as long as two kernels back to back shares a queue.  It returns ``n``
mutually concurrent streams when the process has that many queues, else the largest such set padded with the least-loaded
choices (the default stream's queue is idle during replays and counts as free).
"""
import itertools
import time

import torch

PROBE_CYCLES = 120000          # ~50 us per spin kernel
CANDIDATES = 10


def _pair_seconds(a, b, cycles):
    torch.cuda.synchronize(a.device)
    t0 = time.perf_counter()
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles)
    torch.cuda.synchronize(a.device)
    return time.perf_counter() - t0


def serialised_pairs(streams, cycles=PROBE_CYCLES, reps=3):
    """{(i, j): True if streams[i] and streams[j] do NOT overlap} for i < j."""
    for s in streams:                       # first use binds a stream to its hardware queue
        with torch.cuda.stream(s):
            torch.cuda._sleep(1000)
    one = min(_pair_seconds(streams[0], streams[0], cycles) for _ in range(reps)) / 2.0
    out = {}
    for i, j in itertools.combinations(range(len(streams)), 2):
        t = min(_pair_seconds(streams[i], streams[j], cycles) for _ in range(reps))
        out[(i, j)] = t > 1.6 * one
    return out


def choose_overlapping(serialised, count, n):
    """Indices of ``n`` of ``count`` candidates given ``serialised[(i, j)]`` (i < j): the largest mutually overlapping set of at
    most ``n`` first (smallest indices among equals), then - when the process has fewer queues than ``n`` - one more candidate per
    chosen queue in turn, so that the surplus is spread evenly.  Returns (indices, size of the overlapping set)."""
    def overlap(i, j):
        return not serialised[(min(i, j), max(i, j))]
    best = []
    for size in range(min(n, count), 0, -1):
        for combo in itertools.combinations(range(count), size):
            if all(overlap(i, j) for i, j in itertools.combinations(combo, 2)):
                best = list(combo)
                break
        if best:
            break
    chosen = list(best)
    rest = [i for i in range(count) if i not in chosen]
    k = 0
    while len(chosen) < n and rest:
        target = best[k % len(best)]
        pick = next((i for i in rest if not overlap(i, target)), rest[0])
        chosen.append(pick)
        rest.remove(pick)
        k += 1
    return chosen, len(best)


def concurrent_streams(n, device=None, candidates=CANDIDATES):
    """``n`` torch streams on ``device`` chosen so that as many of them as possible run concurrently (see the module text).
    Returns (streams, distinct) where ``distinct`` is the size of the mutually concurrent set found (== n when all overlap)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    n = int(n)
    if n <= 1:
        return [torch.cuda.Stream(device=device) for _ in range(max(n, 0))], max(n, 0)
    with torch.cuda.device(device):
        pool = [torch.cuda.Stream(device=device) for _ in range(max(candidates, n))]
        ser = serialised_pairs(pool)
    chosen, distinct = choose_overlapping(ser, len(pool), n)
    return [pool[i] for i in chosen], distinct
