set -e
mkdir -p gpurun_out/r4g
python -m pytest tests -m gpu -x -q > gpurun_out/r4g/tests.log 2>&1 || { tail -40 gpurun_out/r4g/tests.log; exit 1; }
tail -3 gpurun_out/r4g/tests.log
for c in c2 c4 c5; do python tools/ab_time.py $c 8 >> gpurun_out/r4g/ab.txt 2>&1; done
python tools/tile_rows_time.py 1024 2 >> gpurun_out/r4g/ab.txt 2>&1
python tools/launch_size_time.py >> gpurun_out/r4g/ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4g/ab.txt
python bench.py --steps 10 --warmup 3 > gpurun_out/r4g/bench.json 2> gpurun_out/r4g/bench.err || { tail -20 gpurun_out/r4g/bench.err; exit 1; }
cat gpurun_out/r4g/bench.json
