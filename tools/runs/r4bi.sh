set -e
O=gpurun_out/r4bi; mkdir -p $O
for i in 1 2; do python tools/ab_time.py c2 6 2>&1 | grep -v amdgpu >> $O/t.txt; RPT_NO_SIZED_KERNELS=1 python tools/ab_time.py c2 6 2>&1 | grep -v amdgpu >> $O/t.txt; done
python tools/compact_time.py 800 600 1 400 2>&1 | grep -v amdgpu >> $O/t.txt
RPT_NO_SIZED_KERNELS=1 python tools/compact_time.py 800 600 1 400 2>&1 | grep -v amdgpu >> $O/t.txt
python tools/compact_time.py 1920 1080 1 200 2>&1 | grep -v amdgpu >> $O/t.txt
RPT_NO_SIZED_KERNELS=1 python tools/compact_time.py 1920 1080 1 200 2>&1 | grep -v amdgpu >> $O/t.txt
python tools/tile_rows_time.py 1024 8 2>&1 | grep -v amdgpu >> $O/t.txt
cat $O/t.txt
