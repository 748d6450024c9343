set -e
mkdir -p gpurun_out/r4l
python -m pytest tests -m gpu -x -q -k "large or grid or config5 or sdf" > gpurun_out/r4l/tests.log 2>&1 || { tail -40 gpurun_out/r4l/tests.log; exit 1; }
tail -2 gpurun_out/r4l/tests.log
python tools/ab_time.py c5 8 2>&1 | grep -v amdgpu
python tools/ab_time.py c4 8 2>&1 | grep -v amdgpu
RPT_PROFILE_KERNEL=render_large bash tools/collect_profiles.sh r4_c5_mega tools/ab_time.py c5 2 > gpurun_out/r4l/collect_c5.log 2>&1 || { tail -20 gpurun_out/r4l/collect_c5.log; exit 1; }
tail -45 gpurun_out/r4l/collect_c5.log
