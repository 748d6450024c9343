set -e
O=gpurun_out/r4bs; mkdir -p $O
for cfg in c2 c4 c5; do for i in 1 2; do python tools/ab_time.py $cfg 6 2>&1 | grep -v amdgpu >> $O/t.txt; done; done
cat $O/t.txt
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
