O=gpurun_out/r5j; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_multi.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for p in 1 0; do for c in c2 c4 c5; do RPT_PERSISTENT_WAVES=$p python tools/ab_time.py $c 5 2>/dev/null | tail -1; done; done | tee $O/times.txt
RPT_DISPATCH_TIMELINE=1 python tools/dispatch_timeline.py c2 2>&1 | grep -v amdgpu | tail -3 | cut -c1-330
