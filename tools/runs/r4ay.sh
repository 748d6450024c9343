set -e
mkdir -p gpurun_out/r4ay
for thr in 32 44 52 56 60 64; do RPT_SHADE_THRESHOLD=$thr python tools/ab_time.py c5 5 >> gpurun_out/r4ay/t.txt 2>&1; done
for thr in 44 52 60 64; do RPT_SHADE_THRESHOLD=$thr python tools/ab_time.py c4 5 >> gpurun_out/r4ay/t.txt 2>&1; done
for thr in 48 52 60; do RPT_SHADE_THRESHOLD=$thr python tools/ab_time.py c2 5 >> gpurun_out/r4ay/t.txt 2>&1; done
for fin in 8 16 32 40; do RPT_FINISH_THRESHOLD=$fin python tools/ab_time.py c2 5 >> gpurun_out/r4ay/t.txt 2>&1; done
grep -v amdgpu gpurun_out/r4ay/t.txt
