set -e
mkdir -p gpurun_out/r4bd
python -m pytest tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/r4bd/tests.log 2>&1 || { tail -60 gpurun_out/r4bd/tests.log; exit 1; }
tail -2 gpurun_out/r4bd/tests.log
