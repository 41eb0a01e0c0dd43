#!/bin/bash
# RoIAlign in isolation under rocprofv3: kernel durations + counter passes (one group per run).
#   gpurun -- 'bash tools/roi_prof.sh r2x "5,7"'
set -o pipefail
TAG=${1:-r2x}
VARS=${2:-7}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 python3 $R/tools/roi_bench.py --variants $VARS --typical > $O/${TAG}_roi_bench.txt 2>&1 || exit $?
cat $O/${TAG}_roi_bench.txt | tail -12
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/${TAG}_roi_trace -o t -- python3 $R/tools/roi_bench.py --variants $VARS --reps 20 > $O/${TAG}_roi_trace.log 2>&1 || exit $?
python3 $R/tools/rocpd_summary.py $O/${TAG}_roi_trace/t_results.db roi | grep -i "roi_align\|kernel |" | head
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "VALUBusy MemUnitBusy" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_roi_pmc_$N -o pmc -- python3 $R/tools/roi_bench.py --variants $VARS --reps 5 > $O/${TAG}_roi_pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -3 $O/${TAG}_roi_pmc_$N.log; continue; }
  python3 $R/tools/pmc_kernel.py $O/${TAG}_roi_pmc_$N roi_align
done
