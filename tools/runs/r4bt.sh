set -e
O=gpurun_out/r4bt; mkdir -p $O; rm -f $O/t.txt
for i in 1 2; do
  python tools/ab_time.py c2 6 2>&1 | grep -v amdgpu >> $O/t.txt
  for wh in "800 600" "1920 1080"; do
    python tools/compact_time.py $wh 1 2>&1 | grep -v amdgpu | tail -3 >> $O/t.txt
    RPT_NO_MATERIAL_TABLE=1 python tools/compact_time.py $wh 1 2>&1 | grep -v amdgpu | tail -3 | sed 's/^/no table: /' >> $O/t.txt
  done
done
cat $O/t.txt
python -m pytest tests/test_gpu_dispatch.py tests/test_gpu_range_guards.py tests/test_gpu_parity.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
