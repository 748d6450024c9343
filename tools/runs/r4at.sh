set -e
O=gpurun_out/r4at; mkdir -p $O
RPT_PROFILE_KERNEL=render_sdf bash tools/collect_profiles.sh r4_c4 tools/ab_time.py c4 3 > $O/c4.log 2>&1 || { tail -20 $O/c4.log; exit 1; }
echo c4 done
RPT_PROFILE_KERNEL=render_large bash tools/collect_profiles.sh r4_c5 tools/ab_time.py c5 3 > $O/c5.log 2>&1 || { tail -20 $O/c5.log; exit 1; }
echo c5 done
python tools/block_profile.py 64 c4 > $O/block_profile_c4.txt 2>&1
python tools/block_profile.py 32 c5 > $O/block_profile_c5.txt 2>&1
python tools/compact_time.py 800 600 1 400 > $O/compact.txt 2>&1
python tools/compact_time.py 1920 1080 1 200 >> $O/compact.txt 2>&1
python tools/host_path_time.py > $O/host_path.txt 2>&1
python tools/spp_curve.py > $O/spp_curve.txt 2>&1 || true
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python -m pytest tests -m gpu -x -q > $O/tests_ab.log 2>&1 || { tail -40 $O/tests_ab.log; exit 1; }
tail -2 $O/tests_ab.log
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
for v in r4_c4 r4_c5; do echo "== $v"; grep -n "HBM bytes\|kernel_stats\|lane util\|issuing\|stalled\|waiting\|SQ_INSTS_VALU \|SQ_INSTS_SALU\|BRANCH\|resident" gpurun_out/prof_$v/summary.txt; done
grep -v amdgpu $O/compact.txt $O/host_path.txt
