set -e
mkdir -p gpurun_out/r4f
python -m pytest tests -m gpu -x -q > gpurun_out/r4f/tests.log 2>&1 || { tail -30 gpurun_out/r4f/tests.log; exit 1; }
tail -3 gpurun_out/r4f/tests.log
for r in 0 6 12 24 48; do
  for c in c2 c4 c5; do RPT_UNIT_ROUNDS=$r python tools/ab_time.py $c 8 >> gpurun_out/r4f/ab.txt 2>&1; done
  RPT_UNIT_ROUNDS=$r python tools/tile_rows_time.py 1024 2 >> gpurun_out/r4f/ab.txt 2>&1
done
RPT_UNIT_ROUNDS=12 python tools/launch_size_time.py >> gpurun_out/r4f/ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4f/ab.txt
