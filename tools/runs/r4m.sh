set -e
mkdir -p gpurun_out/r4m
python -m pytest tests -m gpu -x -q -k "large or grid or config5" > gpurun_out/r4m/tests.log 2>&1 || { tail -40 gpurun_out/r4m/tests.log; exit 1; }
tail -2 gpurun_out/r4m/tests.log
python tools/ab_time.py c5 8 2>&1 | grep -v amdgpu
python tools/ab_time.py c5 8 2>&1 | grep -v amdgpu
