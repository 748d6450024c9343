set -e
O=gpurun_out/evidence_r4; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python -m pytest tests -m gpu -x -q > $O/tests_ab.log 2>&1 || { tail -40 $O/tests_ab.log; exit 1; }
tail -2 $O/tests_ab.log
for tr in 1 2 4 8 16; do python tools/tile_rows_time.py 1024 $tr >> $O/tile_rows.txt 2>&1; done
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
RPT_PROFILE_KERNEL=render_sdf bash tools/collect_profiles.sh r4_c4 tools/ab_time.py c4 3 > $O/c4.log 2>&1 || { tail -20 $O/c4.log; exit 1; }
RPT_PROFILE_KERNEL=render_large bash tools/collect_profiles.sh r4_c5 tools/ab_time.py c5 3 > $O/c5.log 2>&1 || { tail -20 $O/c5.log; exit 1; }
for v in c2_bench c4 c5; do mkdir -p profiles/r4/$v; for f in kernel_stats.csv pmc_FETCH_SIZE.csv pmc_WRITE_SIZE.csv pmc_sq1.csv pmc_sq2.csv pmc_sq3.csv pmc_sq4.csv summary.txt traffic.json; do cp gpurun_out/prof_r4_$v/$f profiles/r4/$v/$f; done; done
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
grep -v amdgpu $O/tile_rows.txt $O/host_path.txt $O/compact.txt
for v in r4_c2_bench r4_c4 r4_c5; do echo "== $v"; grep -n "HBM bytes\|kernel_stats\|lane util\|issuing\|stalled\|waiting\|SQ_INSTS_VALU \|SQ_INSTS_SALU\|BRANCH\|resident\|TRANS_F32 " gpurun_out/prof_$v/summary.txt; done
