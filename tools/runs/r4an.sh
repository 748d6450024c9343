set -e
O=gpurun_out/r4an; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python tools/tile_rows_time.py 1024 2 > $O/tile_rows.txt 2>&1
python tools/launch_size_time.py > $O/launch_size.txt 2>&1
python tools/host_path_time.py > $O/host_path.txt 2>&1
python tools/compact_time.py 800 600 1 400 > $O/compact.txt 2>&1
python tools/compact_time.py 1920 1080 1 200 >> $O/compact.txt 2>&1
python tools/spp_curve.py > $O/spp_curve.txt 2>&1 || true
python tools/block_profile.py 256 c2 > $O/block_profile_c2.txt 2>&1
python tools/block_profile.py 64 c4 > $O/block_profile_c4.txt 2>&1
python tools/block_profile.py 32 c5 > $O/block_profile_c5.txt 2>&1
for c in c2 share; do echo "== $c, most expensive tile first" >> $O/timeline.txt; RPT_DISPATCH_TIMELINE=1 python tools/dispatch_timeline.py $c >> $O/timeline.txt 2>&1; done
bash tools/collect_profiles.sh r4_c2_bench > $O/collect_c2.log 2>&1 || { tail -20 $O/collect_c2.log; exit 1; }
RPT_PROFILE_KERNEL=denoise bash tools/collect_profiles.sh r4_dn tools/ab_time.py dn 3 > $O/dn.log 2>&1 || { tail -20 $O/dn.log; exit 1; }
python tools/denoise_traffic.py gpurun_out/prof_r4_dn | tee $O/dn_traffic.txt
grep -v amdgpu $O/tile_rows.txt $O/launch_size.txt $O/host_path.txt $O/compact.txt $O/spp_curve.txt
tail -34 gpurun_out/prof_r4_c2_bench/summary.txt
head -24 $O/block_profile_c2.txt
