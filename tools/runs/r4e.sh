set -e
mkdir -p gpurun_out/r4e
for k in 1 0; do
  for c in c2 c4 c5; do RPT_DISPATCH_KEY=$k python tools/ab_time.py $c 8 >> gpurun_out/r4e/ab.txt 2>&1; done
  RPT_DISPATCH_KEY=$k python tools/tile_rows_time.py 1024 2 >> gpurun_out/r4e/ab.txt 2>&1
done
echo "== c2 by max" >> gpurun_out/r4e/timeline.txt
RPT_DISPATCH_TIMELINE=1 RPT_DISPATCH_KEY=1 python tools/dispatch_timeline.py c2 >> gpurun_out/r4e/timeline.txt 2>&1
echo "== share by max" >> gpurun_out/r4e/timeline.txt
RPT_DISPATCH_TIMELINE=1 RPT_DISPATCH_KEY=1 python tools/dispatch_timeline.py share >> gpurun_out/r4e/timeline.txt 2>&1
grep -v amdgpu.ids gpurun_out/r4e/ab.txt; grep -v amdgpu.ids gpurun_out/r4e/timeline.txt
