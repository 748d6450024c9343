# round 6: the quantised cell records (host_grid.h, dev_scene_large.h): parity of everything that walks the grid, then timings
set -e
O=gpurun_out/r6d; mkdir -p $O
export AMD_LOG_LEVEL=1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_dispatch.py tests/test_gpu_media.py tests/test_gpu_sharing.py tests/test_gpu_multi.py -m gpu -x -q --capture=sys -k "grid or large or config5 or ground or sphere or media or sharing or light" > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
unset RPT_LIB
for k in 1 2; do python tools/ab_time.py c5 4; done 2>&1 | grep -v amdgpu.ids | tee $O/ab_c5.txt
python tools/ab_time.py c5full 1 2>&1 | grep -v amdgpu.ids | tee -a $O/ab_c5.txt
