set -e
RPT_PROFILE_KERNEL=render_large bash tools/collect_profiles.sh r4_c5_full tools/ab_time.py c5full 1 > gpurun_out/c5_full.log 2>&1 || { tail -20 gpurun_out/c5_full.log; exit 1; }
tail -30 gpurun_out/prof_r4_c5_full/summary.txt
