#!/usr/bin/env python
"""Throughput sweep over convolution plans: the autotuner picks, per shape, the (tile, split-K) pair with the lowest
ISOLATED latency; bench.py's timed mode keeps 4 frames in flight, where the chip is full and what counts is CU-time
per layer, not latency.  This tool times bench.py's timed loop under plan tables derived from the autotuned one.

    python tools/plan_sweep.py [--steps 60] [--out gpurun_out/plan_sweep.json]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TILES = ["256x128", "128x256", "128x128", "128x64", "64x128", "64x64"]
BM = [256, 128, 128, 128, 64, 64]
BN = [128, 256, 128, 64, 128, 64]


def ksteps(row):
    n, h, w, c, k, r, s, st, pad, os_ = row[:10]
    return (r * s * c + 31) // 32


def m_of(row):
    n, h, w, c, k, r, s, st, pad, os_ = row[:10]
    return n * ((h + 2 * pad - r) // st + 1) * ((w + 2 * pad - s) // st + 1)


def with_plan(row, cfg, splits):
    ks = ksteps(row)
    splits = max(1, min(splits, ks))
    sps = (ks + splits - 1) // splits
    splits = (ks + sps - 1) // sps
    return list(row[:10]) + [cfg, splits, sps]


def fill_splits(row, cfg, waves=1.0):
    """smallest K split that gives at least `waves` workgroups per CU (256 CUs)"""
    m, k = m_of(row), row[4]
    tiles = ((m + BM[cfg] - 1) // BM[cfg]) * ((k + BN[cfg] - 1) // BN[cfg])
    ks = ksteps(row)
    sp = 1
    while tiles * sp < 256 * waves and ks // (sp + 1) >= 4 and sp < 16:
        sp += 1
    return sp


def variants(base):
    out = {"autotuned": base}
    out["all_64x64_nosplit"] = [with_plan(r, 5, 1) for r in base]
    out["all_128x128_fill"] = [with_plan(r, 2, fill_splits(r, 2)) for r in base]
    out["all_128x64_fill"] = [with_plan(r, 3, fill_splits(r, 3)) for r in base]
    out["autotuned_nosplit"] = [with_plan(r, r[10], 1) for r in base]
    out["big_m_128x128"] = [with_plan(r, 2, 1) if m_of(r) >= 9000 and r[4] >= 128 else list(r) for r in base]
    out["big_m_256x128"] = [with_plan(r, 0, 1) if m_of(r) >= 9000 and r[4] >= 128 and r[3] % 32 == 0 else list(r) for r in base]
    out["layer3_128x64_fill"] = [with_plan(r, 3, fill_splits(r, 3)) if m_of(r) == 2394 else list(r) for r in base]
    out["layer3_128x128_fill"] = [with_plan(r, 2, fill_splits(r, 2)) if m_of(r) == 2394 else list(r) for r in base]
    out["layer3_nosplit"] = [with_plan(r, r[10], 1) if m_of(r) == 2394 else list(r) for r in base]
    out["layer3_split2"] = [with_plan(r, r[10], min(r[11], 2)) if m_of(r) == 2394 else list(r) for r in base]
    return out


def run_bench(plans_path, steps, streams):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--plans", plans_path, "--steps", str(steps), "--warmup", "8",
           "--no-cpu-baseline", "--streams", str(streams)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    for line in res.stdout.splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            return d["value"], d["roofline"]["achieved"]
    raise RuntimeError(res.stderr[-2000:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "plan_sweep.json"))
    ap.add_argument("--only", default="")
    ap.add_argument("--streams", default="4,1", help="comma list of frames-in-flight settings to time")
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    base_path = args.out.replace(".json", "_base_plans.json")
    if os.path.exists(base_path):
        os.remove(base_path)
    fps, _ = run_bench(base_path, args.steps, 4)          # autotunes and writes the base table
    base = json.load(open(base_path))
    results = {}
    for name, rows in variants(base).items():
        if args.only and name not in args.only.split(","):
            continue
        path = args.out.replace(".json", "_%s.json" % name)
        json.dump(rows, open(path, "w"))
        res = {}
        for st in [int(v) for v in args.streams.split(",")]:
            res[st] = run_bench(path, args.steps, st)
        results[name] = {"fps_by_streams": {st: v[0] for st, v in res.items()}, "isolated_conv_tflops": list(res.values())[0][1]}
        print("%-24s %s   isolated conv %6.1f TFLOP/s" % (name, "  ".join("%d streams %6.1f fps" % (st, v[0]) for st, v in res.items()),
                                                         list(res.values())[0][1]), flush=True)
    json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
