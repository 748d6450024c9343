#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's N=1 line on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <variant>     -> gpurun_out/prof_<variant>/*.csv  (copy into profiles/<round>/<variant>/)
# One --kernel-trace --stats pass of the default bench command, then separate --pmc passes (FETCH_SIZE and WRITE_SIZE
# never together; SQ counters in two groups), as MI355X_MICROARCH.md prescribes.
set -e
V=${1:-current}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$V
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, rocprof args...
    local name=$1; shift
    rm -rf /tmp/rp_$name
    rocprofv3 "$@" --output-format csv -d /tmp/rp_$name -o out -- python3 $REPO/bench.py $BENCH_ARGS --no-cpu-baseline > $OUT/$name.log 2>&1
    echo "pass $name done"
}
BENCH_ARGS="--steps 10 --warmup 2" run stats --kernel-trace --stats
cp $(find /tmp/rp_stats -name '*kernel_stats.csv' | head -1) $OUT/bench_n1_kernel_stats.csv
cp $(find /tmp/rp_stats -name '*domain_stats.csv' | head -1) $OUT/bench_n1_domain_stats.csv || true
BENCH_ARGS="--steps 2 --warmup 1"
run fetch --kernel-trace --pmc FETCH_SIZE
cp $(find /tmp/rp_fetch -name '*counter_collection.csv' | head -1) $OUT/bench_n1_pmc_FETCH_SIZE.csv
run write --kernel-trace --pmc WRITE_SIZE
cp $(find /tmp/rp_write -name '*counter_collection.csv' | head -1) $OUT/bench_n1_pmc_WRITE_SIZE.csv
run sq1 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM SQ_WAIT_ANY
cp $(find /tmp/rp_sq1 -name '*counter_collection.csv' | head -1) $OUT/bench_n1_pmc_SQ_ACTIVE_.csv
run sq2 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES
cp $(find /tmp/rp_sq2 -name '*counter_collection.csv' | head -1) $OUT/bench_n1_pmc_SQ_WAVESS.csv
cd $REPO
python3 tools/make_traffic_json.py $OUT
for f in $OUT/bench_n1_pmc_SQ_*.csv; do python3 tools/pmc_summary.py $f render_small; done
head -3 $OUT/bench_n1_kernel_stats.csv
tail -1 $OUT/stats.log
