set -e
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/tests.log 2>&1 || { tail -30 gpurun_out/r4a/tests.log; exit 1; }
tail -3 gpurun_out/r4a/tests.log
for o in 0 1; do
  for c in c2 c2s c4 c5; do RPT_DISPATCH_ORDER=$o python tools/ab_time.py $c 6 >> gpurun_out/r4a/ab.txt 2>&1; done
  RPT_DISPATCH_ORDER=$o python tools/tile_rows_time.py 1024 2 >> gpurun_out/r4a/tile_rows.txt 2>&1
  RPT_DISPATCH_ORDER=$o python tools/launch_size_time.py >> gpurun_out/r4a/launch_size_$o.txt 2>&1
done
cat gpurun_out/r4a/ab.txt gpurun_out/r4a/tile_rows.txt gpurun_out/r4a/launch_size_*.txt
