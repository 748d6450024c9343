O=gpurun_out/r5f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_multi.py > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for c in c2 c4 c5; do python tools/ab_time.py $c 5 2>/dev/null | tail -1; done | tee $O/times.txt
