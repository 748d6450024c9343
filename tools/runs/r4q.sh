set -e
mkdir -p gpurun_out/r4q
python -m pytest tests -m gpu -x -q -k "sdf or dispatch or media" > gpurun_out/r4q/tests.log 2>&1 || { tail -40 gpurun_out/r4q/tests.log; exit 1; }
tail -2 gpurun_out/r4q/tests.log
python tools/ab_time.py c4 8 2>&1 | grep -v amdgpu
python tools/ab_time.py c4 8 2>&1 | grep -v amdgpu
