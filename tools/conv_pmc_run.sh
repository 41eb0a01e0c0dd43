#!/bin/bash
# gpurun -- 'bash tools/conv_pmc_run.sh': PMC passes over tools/conv_pmc_probe.py (never combined with other trace domains)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_pmc_probe
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python3 $R/tools/conv_pmc_probe.py > $O/probe.json
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
# (a fourth pass with TA_* / TCP_*_STALL counters hung rocprofv3 on this image: not collected)
n=0
for P in "$P1" "$P2" "$P3"; do
  n=$((n+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$n -o pmc -- python3 $R/tools/conv_pmc_probe.py > $O/p$n.log 2>&1 || { echo "pass $n failed"; tail -5 $O/p$n.log; }
  echo "pass $n done"
done
cd $R
python3 tools/conv_pmc_summary.py $O/probe.json $O/p1 $O/p2 $O/p3 > gpurun_out/r05_conv_pmc_probe.md || true
# keep the kernel-trace durations of pass 1 too
python3 - <<PY
import csv, glob, json
paths = glob.glob("$O/p1/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        if "conv_igemm" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
rows.sort()
json.dump(rows, open("gpurun_out/r05_conv_pmc_probe_durations.json", "w"))
print(len(rows), "conv dispatches")
PY
du -sh $O; rm -rf $O/p1 $O/p2 $O/p3
head -c 3000 gpurun_out/r05_conv_pmc_probe.md
