O=gpurun_out/r5h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_denoise.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
bash tools/run_variants.sh tools/ab_time.py dn 10 2>&1 | tee $O/dn.txt
