set -e
mkdir -p gpurun_out/r4t
bash tools/collect_profiles.sh r4_c2_bench > gpurun_out/r4t/c2.log 2>&1 || { tail -20 gpurun_out/r4t/c2.log; exit 1; }
echo c2 done
RPT_PROFILE_KERNEL=render_sdf bash tools/collect_profiles.sh r4_c4 tools/ab_time.py c4 3 > gpurun_out/r4t/c4.log 2>&1 || { tail -20 gpurun_out/r4t/c4.log; exit 1; }
echo c4 done
RPT_PROFILE_KERNEL=render_large bash tools/collect_profiles.sh r4_c5 tools/ab_time.py c5 3 > gpurun_out/r4t/c5.log 2>&1 || { tail -20 gpurun_out/r4t/c5.log; exit 1; }
echo c5 done
RPT_PROFILE_KERNEL=denoise bash tools/collect_profiles.sh r4_dn tools/ab_time.py dn 3 > gpurun_out/r4t/dn.log 2>&1 || { tail -20 gpurun_out/r4t/dn.log; exit 1; }
python tools/denoise_traffic.py gpurun_out/prof_r4_dn | tee gpurun_out/r4t/dn_traffic.txt
for v in r4_c2_bench r4_c4 r4_c5; do echo "== $v"; tail -32 gpurun_out/prof_$v/summary.txt; done
