# Everything under profiles/r6/ that describes the FINAL library, in one gpurun call: the GPU test suite (test build) and the product
# library's own check, the block profiles, the rocprofv3 passes of bench.py / configs[3] / configs[4] / the denoiser (traffic.json is
# stamped with the library's hash: bench.py prints the counters only for the code they belong to), the dispatch timeline, the bench line.
set -e
O=gpurun_out/evidence_r6; mkdir -p $O
export RPT_LIB=    # (unset for the tools below: ab_time / bench load the PRODUCT library; pytest's conftest picks the test build)
unset RPT_LIB
AMD_LOG_LEVEL=1 python -m pytest tests -m gpu -q --capture=sys > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python tools/block_profile.py 256 c2 > $O/block_profile_c2.txt 2>&1
python tools/block_profile.py 64 c4 > $O/block_profile_c4.txt 2>&1
python tools/block_profile.py 32 c5 > $O/block_profile_c5.txt 2>&1
RPT_DISPATCH_TIMELINE=1 python tools/dispatch_timeline.py c2 > $O/timeline.txt 2>&1 || true
export RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip.so
bash tools/collect_profiles.sh r6_c2_bench > $O/collect_c2.log 2>&1 || { tail -20 $O/collect_c2.log; exit 1; }
RPT_PROFILE_KERNEL=render_sdf bash tools/collect_profiles.sh r6_c4 tools/ab_time.py c4 3 > $O/c4.log 2>&1 || { tail -20 $O/c4.log; exit 1; }
RPT_PROFILE_KERNEL=render_large bash tools/collect_profiles.sh r6_c5 tools/ab_time.py c5 3 > $O/c5.log 2>&1 || { tail -20 $O/c5.log; exit 1; }
RPT_PROFILE_KERNEL=render_large bash tools/collect_profiles.sh r6_c5_full tools/ab_time.py c5full 1 > $O/c5_full.log 2>&1 || { tail -20 $O/c5_full.log; exit 1; }
RPT_PROFILE_KERNEL=denoise bash tools/collect_profiles.sh r6_denoise tools/ab_time.py dn 3 > $O/dn.log 2>&1 || { tail -20 $O/dn.log; exit 1; }
python tools/denoise_traffic.py gpurun_out/prof_r6_denoise 1 > gpurun_out/prof_r6_denoise/traffic.txt 2>&1 || true
RPT_PROFILE_KERNEL=render_large bash tools/collect_mem_counters.sh r6_c5 tools/ab_time.py c5 2 > $O/mem_c5.log 2>&1 || true
for v in c2_bench c4 c5 c5_full denoise; do mkdir -p profiles/r6/$v; for f in kernel_stats.csv pmc_FETCH_SIZE.csv pmc_WRITE_SIZE.csv pmc_sq1.csv pmc_sq2.csv pmc_sq3.csv pmc_sq4.csv summary.txt traffic.json traffic.txt traffic_1080p.json traffic_4k.json loaded_lib.txt; do [ -f gpurun_out/prof_r6_$v/$f ] && cp gpurun_out/prof_r6_$v/$f $O/${v}__$f; done; done
unset RPT_LIB
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
for v in r6_c2_bench r6_c4 r6_c5 r6_c5_full r6_denoise; do echo "== $v"; grep -n "HBM bytes\|kernel_stats\|lane util\|issuing\|stalled\|waiting\|SQ_INSTS_VALU \|SQ_INSTS_SALU\|BRANCH\|resident" gpurun_out/prof_$v/summary.txt; done
cat gpurun_out/prof_r6_denoise/traffic.txt
