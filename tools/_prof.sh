R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for c in c4 c5; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r1d_stats_$c -o run --output-format csv -- python3 $R/tools/c4_run.py $c > $R/gpurun_out/r1d_stats_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU -d $R/gpurun_out/r1d_pmc_$c -o run --output-format csv -- python3 $R/tools/c4_run.py $c > $R/gpurun_out/r1d_pmc_$c.log 2>&1
done
ls $R/gpurun_out/r1d_*
