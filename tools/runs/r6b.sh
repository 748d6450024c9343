# round 6: memory-pipeline counters of the shipped large-scene kernel (2048^2 x 32 spp), to see what bounds the grid walks now
set -e
export RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip.so
RPT_PROFILE_KERNEL=render_large bash tools/collect_mem_counters.sh r6_c5 tools/ab_time.py c5 2 > gpurun_out/r6b.log 2>&1 || { tail -30 gpurun_out/r6b.log; exit 1; }
cat gpurun_out/mem_r6_c5/summary.txt
