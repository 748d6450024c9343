set -e
O=gpurun_out/r4bm; mkdir -p $O
python -m pytest tests/test_gpu_dispatch.py -m gpu -x -q > $O/d.log 2>&1 || { tail -40 $O/d.log; exit 1; }
tail -2 $O/d.log
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for cfg in c2 c4 c5; do python tools/ab_time.py $cfg 6 2>&1 | grep -v amdgpu >> $O/t.txt; done
RPT_NO_SIZED_KERNELS=1 python tools/ab_time.py c4 6 2>&1 | grep -v amdgpu >> $O/t.txt
cat $O/t.txt
