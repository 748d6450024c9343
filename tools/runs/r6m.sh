# round 6: the SDF kernel's second room: parity of everything SDF, the block profile, the march-phase minimum re-swept with the room on
set -e
O=gpurun_out/r6m; mkdir -p $O
export AMD_LOG_LEVEL=1
python -m pytest tests -m gpu -x -q --capture=sys -k "sdf or media or sharing or dispatch or shipped" > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
unset AMD_LOG_LEVEL
python tools/block_profile.py 64 c4 2>&1 | grep -v amdgpu.ids | tee $O/block_profile_c4.txt
export RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip.so
for ml in 4 6 8 10 12 16 24; do RPT_SDF_MARCH_MIN_LANES=$ml python tools/ab_time.py c4 5 2>&1 | grep -v amdgpu.ids; done | tee $O/sweep_min_lanes.txt
