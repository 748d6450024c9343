set -e
O=gpurun_out/r6o; mkdir -p $O
export RPT_LIB=$PWD/rust-pathtracer_amd/variants/leave.so
for la in 65 56 48 40 32 24 16; do RPT_SDF_LEAVE_AT=$la python tools/ab_time.py c4 5 2>&1 | grep -v amdgpu.ids; done | tee $O/sweep.txt
