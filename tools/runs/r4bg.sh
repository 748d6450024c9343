set -e
O=gpurun_out/r4bg; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do python tools/ab_time.py c2 6 2>&1 | grep -v amdgpu >> $O/t.txt; RPT_NO_SIZED_KERNELS=1 python tools/ab_time.py c2 6 2>&1 | grep -v amdgpu >> $O/t.txt; done
python tools/compact_time.py 800 600 1 400 2>&1 | grep -v amdgpu >> $O/t.txt
RPT_NO_SIZED_KERNELS=1 python tools/compact_time.py 800 600 1 400 2>&1 | grep -v amdgpu >> $O/t.txt
python tools/compact_time.py 1920 1080 1 200 2>&1 | grep -v amdgpu >> $O/t.txt
RPT_NO_SIZED_KERNELS=1 python tools/compact_time.py 1920 1080 1 200 2>&1 | grep -v amdgpu >> $O/t.txt
cat $O/t.txt
