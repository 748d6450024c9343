set -e
mkdir -p gpurun_out/r4c
RPT_DISPATCH_WAYS=7 python tools/dispatch_order_map.py c2 > gpurun_out/r4c/map_c2_w7.txt 2>&1
for cfg in "1 0" "3 1536" "7 1536" "7 0" "7 3000" "13 1536" "2 1536"; do
  set -- $cfg
  RPT_DISPATCH_WAYS=$1 RPT_DISPATCH_TAIL=$2 python tools/ab_time.py c2 8 >> gpurun_out/r4c/ab.txt 2>&1
done
RPT_DISPATCH_ORDER=0 python tools/ab_time.py c2 8 >> gpurun_out/r4c/ab.txt 2>&1
for cfg in "7 1536" "3 1536"; do
  set -- $cfg
  for c in c4 c5; do RPT_DISPATCH_WAYS=$1 RPT_DISPATCH_TAIL=$2 python tools/ab_time.py $c 6 >> gpurun_out/r4c/ab.txt 2>&1; done
  RPT_DISPATCH_WAYS=$1 RPT_DISPATCH_TAIL=$2 python tools/tile_rows_time.py 1024 2 >> gpurun_out/r4c/ab.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r4c/ab.txt; grep "launch\|corr" gpurun_out/r4c/map_c2_w7.txt
