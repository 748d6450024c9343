set -e
mkdir -p gpurun_out/r4au
for cfg in c2 c4 c5; do for i in 1 2; do python tools/ab_time.py $cfg 6 >> gpurun_out/r4au/t.txt 2>&1; done; done
python tools/compact_time.py 800 600 1 400 >> gpurun_out/r4au/t.txt 2>&1
python tools/compact_time.py 1920 1080 1 200 >> gpurun_out/r4au/t.txt 2>&1
grep -v amdgpu gpurun_out/r4au/t.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r4au/tests.log 2>&1 || { tail -40 gpurun_out/r4au/tests.log; exit 1; }
tail -2 gpurun_out/r4au/tests.log
