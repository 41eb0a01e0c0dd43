#!/bin/bash
# map-resident RoIAlign under rocprofv3: duration + SQ counters (one group per run).  gpurun -- 'bash tools/roi_prof2.sh TAG "MASKS"'
set -o pipefail
TAG=${1:-r3x}
MASKS=${2:-0}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/${TAG}_trace -o t -- python3 $R/tools/roi_ablate.py $MASKS > $O/${TAG}_trace.log 2>&1 || exit $?
python3 $R/tools/rocpd_summary.py $O/${TAG}_trace/t_results.db roi | grep -i "roi_align\|kernel |" | head
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/${TAG}_pmc_$N -o pmc -- python3 $R/tools/roi_ablate.py $MASKS > $O/${TAG}_pmc_$N.log 2>&1 || { echo "pmc $C failed"; tail -3 $O/${TAG}_pmc_$N.log; continue; }
  python3 $R/tools/pmc_kernel.py $O/${TAG}_pmc_$N roi_align_fwd_resident
done
