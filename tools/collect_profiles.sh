#!/bin/bash
# End-of-round evidence on a GPU box (run through gpurun): headline bench (incl. drop_in, extra_configs), kernel-trace stats
# of the same command, the PMC passes (one counter per run, never combined with other trace domains) over the eager
# one-stream mode AND over the timed mode (hipGraph replays on 4 streams).  The first run saves its tuned conv plans; the
# profiler runs load them, so they measure exactly the kernels that were timed.
#   gpurun --timeout 1150 -- "bash tools/collect_profiles.sh r05 $(git rev-parse --short HEAD)"
set -e -o pipefail
TAG=${1:-r05}
HEAD=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -f $O/${TAG}_plans.json
python3 $R/bench.py --retune --plans $O/${TAG}_plans.json > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
echo "bench done"; cut -c1-300 $O/${TAG}_bench.json
QUIET="--no-cpu-baseline --no-drop-in --no-extra-configs"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o ${TAG} -- python3 $R/bench.py --plans $O/${TAG}_plans.json --steps 30 --warmup 5 $QUIET > $O/${TAG}_prof.log 2>&1
echo "kernel trace done"
for C in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_pmc_$C -o pmc -- python3 $R/bench.py --plans $O/${TAG}_plans.json --steps 3 --warmup 1 --no-graph --streams 1 $QUIET > $O/${TAG}_$C.log 2>&1
  echo "$C done"
done
# the TIMED mode under the counters: warm-up frames + 60 graph replays on 4 streams, nothing else
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/${TAG}_pmc_timed -o pmc -- python3 $R/bench.py --plans $O/${TAG}_plans.json --only-timed --steps 60 > $O/${TAG}_only_timed.json 2> $O/${TAG}_only_timed.err
echo "timed-mode MfmaUtil done"; cat $O/${TAG}_only_timed.json
# training step (BASELINE.json configs[3] names a "rocprof MFMA capture"): MfmaUtil per kernel over the last steps, eager and
# as the replayed single-chain graphs of the pipeline
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/${TAG}_train_pmc -o pmc -- python3 $R/tools/bench_configs.py --train --steps 12 > $O/${TAG}_train_pmc.log 2>&1
echo "train MfmaUtil done"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_train_prof -o t -- python3 $R/tools/bench_configs.py --train --graph --steps 12 > $O/${TAG}_train_prof.log 2>&1
echo "train trace done"
# summaries (small, these are what gets committed under profiles/); the raw traces stay on the box
S=$O/${TAG}_summary
mkdir -p $S
cd $R
CMD="python3 bench.py --plans <plans of the timed run> --steps 30 --warmup 5 $QUIET (4 frames in flight, hipGraph replay)"
FLOPS=$(python3 -c "import json,sys; print(json.load(open(sys.argv[1]))['roofline']['flops_per_frame'])" $O/${TAG}_bench.json)
CALLS=$(python3 -c "import json,sys; print(json.load(open(sys.argv[1]))['roofline']['conv_calls_per_frame'])" $O/${TAG}_bench.json)
python3 tools/rocpd_summary.py $O/${TAG}_prof/${TAG}_results.db "$CMD" --conv-cross-check --flops-per-frame=$FLOPS --conv-calls=$CALLS > $S/kernel_stats.md
python3 tools/pmc_traffic.py $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE --mfma=$O/${TAG}_pmc_MfmaUtil --conv-calls=$CALLS --plans=$O/${TAG}_plans.json --head=$HEAD "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE|MfmaUtil> --output-format csv -- python3 bench.py --plans <plans of the timed run> --steps 3 --warmup 1 --no-graph --streams 1 $QUIET" > $S/pmc_traffic.json
python3 tools/pmc_timed.py --plans=$O/${TAG}_plans.json --head=$HEAD --tag=$TAG $O/${TAG}_pmc_timed $O/${TAG}_only_timed.json "rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -- python3 bench.py --plans <plans of the timed run> --only-timed --steps 60" > $S/pmc_timed.json
python3 tools/pmc_mfma.py $O/${TAG}_train_pmc "rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -- python3 tools/bench_configs.py --train --steps 12" --last-frac=0.4 > $S/train_pmc.json
python3 tools/rocpd_summary.py $O/${TAG}_train_prof/t_results.db "python3 tools/bench_configs.py --train --graph --steps 12 (res101+FPN 1000x600 forward+backward replayed as a hipGraph; includes the plan autotuning launches of the warm-up)" > $S/train_kernel_stats.md
python3 tools/train_busy.py $O/${TAG}_train_prof/t_results.db atl_overlap_kernel 8 > $S/train_busy.txt
cp $O/${TAG}_bench.json $O/${TAG}_plans.json $S/
rm -rf $O/${TAG}_prof $O/${TAG}_pmc_FETCH_SIZE $O/${TAG}_pmc_WRITE_SIZE $O/${TAG}_pmc_MfmaUtil $O/${TAG}_pmc_timed $O/${TAG}_train_pmc $O/${TAG}_train_prof
echo "summaries in $S"; ls -la $S
