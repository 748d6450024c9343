set -e
mkdir -p gpurun_out/r4ah
for cfg in c2 c4 c5; do python tools/ab_time.py $cfg 6 >> gpurun_out/r4ah/t.txt 2>&1; done
python tools/compact_time.py 800 600 1 400 >> gpurun_out/r4ah/t.txt 2>&1
python tools/compact_time.py 1920 1080 1 200 >> gpurun_out/r4ah/t.txt 2>&1
python tools/compact_time.py 1920 1080 4 64 >> gpurun_out/r4ah/t.txt 2>&1
python tools/compact_time.py 1920 1080 32 16 >> gpurun_out/r4ah/t.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4ah/t.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r4ah/tests.log 2>&1 || { tail -60 gpurun_out/r4ah/tests.log; exit 1; }
tail -3 gpurun_out/r4ah/tests.log
