set -e
mkdir -p gpurun_out/r4al
for i in 1 2; do python tools/ab_time.py c4 8 >> gpurun_out/r4al/c4.txt 2>&1; done
grep -v amdgpu.ids gpurun_out/r4al/c4.txt
python -m pytest tests -m gpu -x -q -k "sdf" > gpurun_out/r4al/tests.log 2>&1 || { tail -40 gpurun_out/r4al/tests.log; exit 1; }
tail -2 gpurun_out/r4al/tests.log
