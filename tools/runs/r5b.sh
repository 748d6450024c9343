# round 5, batch b: the refactored tree — GPU tests on the test build, timings of the three headline kernels
O=gpurun_out/r5b; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=25 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
for c in c2 c4 c5; do python tools/ab_time.py $c 5 2>/dev/null | tail -1 >> $O/base.txt; done
python tools/ab_time.py c5full 2 2>/dev/null | tail -1 >> $O/base.txt
python tools/compact_time.py 800 600 1 400 2>/dev/null | tail -1 >> $O/base.txt
cat $O/base.txt
