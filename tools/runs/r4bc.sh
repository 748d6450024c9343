set -e
mkdir -p gpurun_out/r4bc
for tr in 1 2 4 8 16; do python tools/tile_rows_time.py 1024 $tr 2>&1 | grep -v amdgpu >> gpurun_out/r4bc/t.txt; done
cat gpurun_out/r4bc/t.txt
