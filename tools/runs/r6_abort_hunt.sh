# hunt the intermittent SIGABRT of the first probe launch.  N fresh processes, each one first launch; a failing
# run keeps its stderr (AMD_LOG_LEVEL per variant) under gpurun_out/r6_abort/.
O=gpurun_out/r6_abort; mkdir -p $O
N=${1:-150}
fail=0
for variant in plain blocking log3; do
  case $variant in
    plain) export AMD_LOG_LEVEL=0; unset HIP_LAUNCH_BLOCKING;;
    blocking) export AMD_LOG_LEVEL=0; export HIP_LAUNCH_BLOCKING=1;;
    log3) export AMD_LOG_LEVEL=3; unset HIP_LAUNCH_BLOCKING;;
  esac
  n=$N; [ $variant != plain ] && n=$((N / 3))
  ok=0
  for i in $(seq 1 $n); do
    python tools/probe_first_launch.py gen_ray > $O/out.txt 2> $O/err.txt; rc=$?
    if [ $rc -eq 0 ] && grep -q "FIRST-LAUNCH OK" $O/out.txt; then ok=$((ok + 1)); else fail=$((fail + 1)); cp $O/err.txt $O/fail_${variant}_$i.err; cp $O/out.txt $O/fail_${variant}_$i.out; echo "$variant run $i: rc=$rc" | tee -a $O/failures.txt; fi
    [ $((i % 25)) -eq 0 ] && echo "$variant: $i runs, $ok ok"
  done
  echo "$variant: $ok of $n ok" | tee -a $O/summary.txt
done
rm -f $O/out.txt $O/err.txt
echo "failures: $fail" | tee -a $O/summary.txt
exit 0
