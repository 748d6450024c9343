set -e
O=gpurun_out/r4bw; mkdir -p $O; rm -f $O/t.txt
for i in 1 2; do
  python tools/ab_time.py c2 6 2>&1 | grep -v amdgpu >> $O/t.txt
done
python tools/ab_time.py c4 4 2>&1 | grep -v amdgpu >> $O/t.txt
python tools/ab_time.py c5 4 2>&1 | grep -v amdgpu >> $O/t.txt
cat $O/t.txt
python -m pytest tests/test_gpu_parity.py tests/test_gpu_probes.py tests/test_gpu_dispatch.py tests/test_gpu_range_guards.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
