#!/usr/bin/env python
"""Matrix-pipe busy time per frame of the TIMED mode (hipGraph replays on 4 streams) from one rocprofv3 PMC pass:

    rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d gpurun_out/pmc_timed -o pmc -- \
        python3 bench.py --plans <plans of the timed run> --only-timed --steps 60 > only_timed.json
    python tools/pmc_timed.py gpurun_out/pmc_timed only_timed.json "<command>" > profiles/r04_pmc_timed.json

``bench.py --only-timed`` runs the runners' warm-up frames and exactly --steps graph replays (4 frames in flight) and nothing
else, and prints how many frames that was; every frame - eager warm-up or replay - launches the same kernels with the same
plans, so  sum over ALL dispatches of (MfmaUtil x duration) / frames  is the matrix-pipe busy time of one frame in
milliseconds of a fully busy chip.  bench.py divides it by the ms_per_step of its timed run: the MFMA utilisation of the
timed mode (`roofline.mfma_util_timed_percent`).  The file also records how the dispatches overlapped under the counter
collection (sum of durations / wall span) and the duration-weighted MfmaUtil per kernel family in this mode."""
import csv
import glob
import json
import os
import re
import sys


def plans_sha(path):
    """bench.plans_sha of a saved plan table (hash independent of row order)."""
    import hashlib
    with open(path) as f:
        rows = json.load(f)
    return hashlib.sha1(json.dumps(sorted([int(v) for v in r] for r in rows)).encode()).hexdigest()[:16]


def main(directory, only_timed_json, command, plans=None, head=None, tag="r05"):
    paths = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not paths:
        raise SystemExit("no *counter_collection.csv under %s" % directory)
    with open(only_timed_json) as f:
        info = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])
    frames = int(info["frames_total"])
    rows = []
    with open(paths[0], newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == "MfmaUtil":
                rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"], float(row["Counter_Value"]),
                             row.get("Queue_Id", "")))
    rows.sort()
    dur = sum(e - s for s, e, _, _, _ in rows)
    busy = sum((e - s) * u / 100.0 for s, e, _, u, _ in rows)
    span = rows[-1][1] - rows[0][0]
    # overlap among the dispatches as collected: union of the busy intervals vs the sum of the durations
    union, cur_s, cur_e = 0, None, None
    for s, e, _, _, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    fam = {}
    for s, e, name, u, _ in rows:
        m = re.search(r"(conv_igemm\w*<[^>]*>|conv_\w+|wino_\w+|roi_\w+|nms_\w+|topk_\w+)", name)
        key = m.group(1) if m else name.split("(")[0][:48]
        n, sd, sb = fam.get(key, (0, 0, 0.0))
        fam[key] = (n + 1, sd + (e - s), sb + (e - s) * u / 100.0)
    out = {"file": "profiles/%s_pmc_timed.json" % tag, "command": command, "collected_at_head": head,
           "plans_sha": plans_sha(plans) if plans else None, "frames_total": frames,
           "frames_replayed_as_graphs": int(info["frames_replayed"]), "dispatches": len(rows),
           "queues": len({q for _, _, _, _, q in rows}),
           "kernel_ms_per_frame": dur / 1e6 / frames, "mfma_busy_ms_per_frame": busy / 1e6 / frames,
           "mfma_util_percent_of_kernel_time": 100.0 * busy / dur,
           "dispatch_overlap_under_collection": dur / union,
           "overlap_what": "sum of dispatch durations / union of their intervals: 1.0 = the counter collection serialised the "
                           "kernels (per-dispatch MfmaUtil is then each kernel's own); the busy time per frame does not depend on it",
           "by_kernel": {k: {"launches_per_frame": n / frames, "us_per_frame": sd / 1e3 / frames,
                             "mfma_util_percent": 100.0 * sb / sd if sd else 0.0}
                         for k, (n, sd, sb) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    opts = {a.split("=", 1)[0][2:]: a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--") and "=" in a}
    args = [a for a in sys.argv[1:] if not (a.startswith("--") and "=" in a)]
    main(args[0], args[1], " ".join(args[2:]), opts.get("plans"), opts.get("head"), opts.get("tag", "r05"))
