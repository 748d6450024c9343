set -e
mkdir -p gpurun_out/r4ax
python -m pytest tests -m gpu -x -q > gpurun_out/r4ax/tests.log 2>&1 || { tail -40 gpurun_out/r4ax/tests.log; exit 1; }
tail -2 gpurun_out/r4ax/tests.log
python tools/ab_time.py c2 6 2>&1 | grep -v amdgpu
