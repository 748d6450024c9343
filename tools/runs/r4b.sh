set -e
mkdir -p gpurun_out/r4b
python tools/dispatch_order_map.py c2 > gpurun_out/r4b/map_c2.txt 2>&1
for o in 0 2 3 4 1; do
  RPT_DISPATCH_ORDER=$o python tools/ab_time.py c2 8 >> gpurun_out/r4b/ab.txt 2>&1
done
cat gpurun_out/r4b/ab.txt; cat gpurun_out/r4b/map_c2.txt
