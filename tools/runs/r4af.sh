set -e
mkdir -p gpurun_out/r4af
for cfg in c2 c4 c5; do python tools/ab_time.py $cfg 6 >> gpurun_out/r4af/t.txt 2>&1; done
grep -v amdgpu.ids gpurun_out/r4af/t.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r4af/tests.log 2>&1 || { tail -60 gpurun_out/r4af/tests.log; exit 1; }
tail -3 gpurun_out/r4af/tests.log
