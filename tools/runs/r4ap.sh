set -e
O=gpurun_out/r4ap; mkdir -p $O
for i in 1 2; do python tools/ab_time.py c2 6 >> $O/t.txt 2>&1; done
grep -v amdgpu $O/t.txt
python -m pytest tests -m gpu -x -q -k "parity or range_guards or dispatch or media" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
bash tools/collect_profiles.sh r4_c2_bench > $O/collect_c2.log 2>&1 || { tail -20 $O/collect_c2.log; exit 1; }
grep -n "VMEM\|HBM bytes\|kernel_stats\|lane util" gpurun_out/prof_r4_c2_bench/summary.txt
