set -e
mkdir -p gpurun_out/r4aa
RPT_LARGE_WALK_CAP=4 python -m pytest tests -m gpu -x -q -k "large or grid or config5 or spheres or every_kernel" > gpurun_out/r4aa/tests.log 2>&1 || { tail -40 gpurun_out/r4aa/tests.log; exit 1; }
tail -2 gpurun_out/r4aa/tests.log
for cap in 0 2 3 4 6 8 12 0; do RPT_LARGE_WALK_CAP=$cap python tools/ab_time.py c5 6 >> gpurun_out/r4aa/c5.txt 2>&1; done
for thr in 32 44; do for cap in 3 5; do RPT_SHADE_THRESHOLD=$thr RPT_LARGE_WALK_CAP=$cap python tools/ab_time.py c5 6 >> gpurun_out/r4aa/c5.txt 2>&1; done; done
cat gpurun_out/r4aa/c5.txt
