#!/bin/bash
# Collect the rocprofv3 evidence for one command on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <variant> [program args...]     -> gpurun_out/prof_<variant>/*.csv
# Default program: bench.py's N=1 line (the command whose summaries go to profiles/<round>/c2_bench/).  Another
# program (e.g. `tools/c4_run.py c5`) is given after the variant; it is run as `python3 <program args>` so that the
# profiler's child is the interpreter itself (no shell or env hop after `--`).
# One --kernel-trace --stats pass, then separate --pmc passes (FETCH_SIZE and WRITE_SIZE never together; SQ counters
# in groups of 8), as MI355X_MICROARCH.md prescribes.  Copy what is to be judged into profiles/.
set -e
V=${1:-current}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$V
mkdir -p $OUT
if [ $# -gt 0 ]; then PROG=("$@"); STATS_PROG=("$@"); KPAT=${RPT_PROFILE_KERNEL:-render_};
else PROG=(bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only); STATS_PROG=(bench.py --steps 10 --warmup 2 --no-cpu-baseline --headline-only); KPAT=render_small; fi
cd /tmp && export TMPDIR=/tmp
export RPT_LOADED_LIB_RECORD=$OUT/loaded_lib.txt      # rust-pathtracer_amd/_lib.py writes "<sha256> <path>" of the library the profiled process loads
rm -f $RPT_LOADED_LIB_RECORD
run() {   # name, rocprof args...
    local name=$1; shift
    rm -rf /tmp/rp_$name
    if [ $name = stats ]; then rocprofv3 "$@" --output-format csv -d /tmp/rp_$name -o out -- python3 $REPO/${STATS_PROG[0]} "${STATS_PROG[@]:1}" > $OUT/$name.log 2>&1
    else rocprofv3 "$@" --output-format csv -d /tmp/rp_$name -o out -- python3 $REPO/${PROG[0]} "${PROG[@]:1}" > $OUT/$name.log 2>&1; fi
    echo "pass $name done"
}
run stats --kernel-trace --stats
cp $(find /tmp/rp_stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
run fetch --kernel-trace --pmc FETCH_SIZE
cp $(find /tmp/rp_fetch -name '*counter_collection.csv' | head -1) $OUT/pmc_FETCH_SIZE.csv
run write --kernel-trace --pmc WRITE_SIZE
cp $(find /tmp/rp_write -name '*counter_collection.csv' | head -1) $OUT/pmc_WRITE_SIZE.csv
run sq1 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM SQ_WAIT_ANY
cp $(find /tmp/rp_sq1 -name '*counter_collection.csv' | head -1) $OUT/pmc_sq1.csv
run sq2 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES
cp $(find /tmp/rp_sq2 -name '*counter_collection.csv' | head -1) $OUT/pmc_sq2.csv
run sq3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64
cp $(find /tmp/rp_sq3 -name '*counter_collection.csv' | head -1) $OUT/pmc_sq3.csv
run sq4 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAIT_INST_LDS SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE
cp $(find /tmp/rp_sq4 -name '*counter_collection.csv' | head -1) $OUT/pmc_sq4.csv
cd $REPO
python3 tools/pmc_summary.py $OUT $KPAT | tee $OUT/summary.txt
