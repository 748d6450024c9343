#!/bin/bash
# Memory-pipeline counters (TA / TCP / TD) of one command on the GPU box, separate --pmc passes:
# (at most two counters of a block per pass: more "exceeds the capabilities of the hardware")
#   bash tools/collect_mem_counters.sh <variant> <program args...>   -> gpurun_out/mem_<variant>/pmc_*.csv
set -e
V=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/mem_$V
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
    local name=$1; shift
    rm -rf /tmp/rm_$name
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/rm_$name -o out -- python3 $REPO/${PROG[0]} "${PROG[@]:1}" > $OUT/$name.log 2>&1
    cp $(find /tmp/rm_$name -name '*counter_collection.csv' | head -1) $OUT/pmc_$name.csv
    echo "pass $name done"
}
PROG=("$@")
run ta1 GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
run ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
run tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp3 TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run tcp4 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
run td TD_TD_BUSY_sum TD_TC_STALL_sum
PMC_AGG= python3 $REPO/tools/pmc_summary.py $OUT ${RPT_PROFILE_KERNEL:-render_} | tee $OUT/summary.txt
