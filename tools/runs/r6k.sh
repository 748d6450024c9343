# round 6: the SDF kernel's second room: threshold x march-phase minimum (runtime knobs), product library
set -e
O=gpurun_out/r6k; mkdir -p $O
export RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip.so
for room in 0 28 32 36 40 44 48 52 56; do RPT_SDF_SHADE_ROOM=$room python tools/ab_time.py c4 5 2>&1 | grep -v amdgpu.ids; done | tee $O/sweep_room.txt
for ml in 4 6 8 10 12 16; do RPT_SDF_MARCH_MIN_LANES=$ml python tools/ab_time.py c4 5 2>&1 | grep -v amdgpu.ids; done | tee $O/sweep_min_lanes.txt
for ml in 6 12; do for room in 36 44; do RPT_SDF_SHADE_ROOM=$room RPT_SDF_MARCH_MIN_LANES=$ml python tools/ab_time.py c4 5 2>&1 | grep -v amdgpu.ids; done; done | tee $O/sweep_both.txt
