set -e
mkdir -p gpurun_out/r4ai
python -m pytest tests/test_gpu_range_guards.py -m gpu -x -q > gpurun_out/r4ai/guards.log 2>&1 || { tail -60 gpurun_out/r4ai/guards.log; exit 1; }
tail -3 gpurun_out/r4ai/guards.log
python -m pytest tests -m gpu -x -q > gpurun_out/r4ai/tests.log 2>&1 || { tail -60 gpurun_out/r4ai/tests.log; exit 1; }
tail -3 gpurun_out/r4ai/tests.log
for cfg in c2 c4 c5; do python tools/ab_time.py $cfg 6 >> gpurun_out/r4ai/t.txt 2>&1; done
python tools/compact_time.py 800 600 1 400 >> gpurun_out/r4ai/t.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4ai/t.txt
