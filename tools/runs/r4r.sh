set -e
mkdir -p gpurun_out/r4r
python -m pytest tests -m gpu -x -q > gpurun_out/r4r/tests.log 2>&1 || { tail -40 gpurun_out/r4r/tests.log; exit 1; }
tail -2 gpurun_out/r4r/tests.log
for c in c2 c4 c2s; do python tools/ab_time.py $c 8 2>&1 | grep -v amdgpu; done
python tools/compact_time.py 800 600 1 400 2>&1 | grep -v amdgpu
python tools/compact_time.py 1920 1080 1 200 2>&1 | grep -v amdgpu
