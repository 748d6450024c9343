set -e
mkdir -p gpurun_out/r4y
python -m pytest tests -m gpu -x -q -k "denoise or scratch" > gpurun_out/r4y/tests.log 2>&1 || { tail -40 gpurun_out/r4y/tests.log; exit 1; }
tail -2 gpurun_out/r4y/tests.log
python tools/ab_time.py dn 20 2>&1 | grep -v amdgpu
python tools/ab_time.py dn 20 2>&1 | grep -v amdgpu
