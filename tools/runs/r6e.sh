# round 6: where the quantised-record walk spends its time: block profile, SQ counters, memory-pipeline counters (2048^2 x 32 spp)
set -e
O=gpurun_out/r6e; mkdir -p $O
python tools/block_profile.py 32 c5 > $O/block_profile_c5.txt 2>&1 || { tail $O/block_profile_c5.txt; exit 1; }
cat $O/block_profile_c5.txt
export RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip.so
RPT_PROFILE_KERNEL=render_large bash tools/collect_profiles.sh r6e_c5 tools/ab_time.py c5 2 > $O/collect.log 2>&1 || { tail -20 $O/collect.log; exit 1; }
grep -n "HBM bytes\|kernel_stats\|lane util\|issuing\|stalled\|waiting\|SQ_INSTS_VALU \|SQ_INSTS_SALU\|BRANCH\|resident\|SQ_INSTS_VMEM \|SQ_INSTS_SMEM\|SQ_INSTS_LDS " gpurun_out/prof_r6e_c5/summary.txt
RPT_PROFILE_KERNEL=render_large bash tools/collect_mem_counters.sh r6e_c5 tools/ab_time.py c5 2 > $O/mem.log 2>&1 || { tail -20 $O/mem.log; exit 1; }
cat gpurun_out/mem_r6e_c5/summary.txt
