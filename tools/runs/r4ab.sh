set -e
mkdir -p gpurun_out/r4ab
python -m pytest tests/test_gpu_multi.py -m gpu -x -q > gpurun_out/r4ab/tests.log 2>&1 || { tail -60 gpurun_out/r4ab/tests.log; exit 1; }
tail -3 gpurun_out/r4ab/tests.log
