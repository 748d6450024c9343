set -e
mkdir -p gpurun_out/r4n
python tools/block_profile.py 32 c5 > gpurun_out/r4n/block_profile_c5.txt 2>&1
grep -v amdgpu gpurun_out/r4n/block_profile_c5.txt
