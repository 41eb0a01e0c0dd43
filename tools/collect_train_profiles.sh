#!/bin/bash
# The training-step part of tools/collect_profiles.sh alone (the forward kernels and their PMC passes did not change):
#   gpurun --timeout 900 -- "bash tools/collect_train_profiles.sh r05"
set -e -o pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
S=$O/${TAG}_summary
mkdir -p $S
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/${TAG}_train_pmc -o pmc -- python3 $R/tools/bench_configs.py --train --steps 12 > $O/${TAG}_train_pmc.log 2>&1
echo "train MfmaUtil done"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_train_prof -o t -- python3 $R/tools/bench_configs.py --train --graph --steps 12 > $O/${TAG}_train_prof.log 2>&1
echo "train trace done"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_lidar_prof -o t -- python3 $R/tools/bench_configs.py --lidar-train --modes graph --steps 12 > $O/${TAG}_lidar_prof.log 2>&1
echo "lidar train trace done"
cd $R
python3 tools/pmc_mfma.py $O/${TAG}_train_pmc "rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -- python3 tools/bench_configs.py --train --steps 12" --last-frac=0.4 > $S/train_pmc.json
python3 tools/rocpd_summary.py $O/${TAG}_train_prof/t_results.db "python3 tools/bench_configs.py --train --graph --steps 12 (res101+FPN 1000x600 forward+backward replayed as a hipGraph; includes the plan autotuning launches of the warm-up)" > $S/train_kernel_stats.md
python3 tools/train_busy.py $O/${TAG}_train_prof/t_results.db atl_overlap_kernel 8 > $S/train_busy.txt
python3 tools/train_busy.py $O/${TAG}_lidar_prof/t_results.db atl_overlap_kernel 8 > $S/lidar_train_busy.txt
rm -rf $O/${TAG}_train_pmc $O/${TAG}_train_prof $O/${TAG}_lidar_prof
head -3 $S/train_busy.txt; head -3 $S/lidar_train_busy.txt
