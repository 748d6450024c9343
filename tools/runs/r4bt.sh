set -e
O=gpurun_out/r4bt; mkdir -p $O; rm -f $O/t.txt
python -m pytest tests/test_gpu_dispatch.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
