# configs[3] on the final SDF kernel: the march phase's minimum of marching lanes, pairing distance aside (round 5)
O=gpurun_out/r5_sweep; mkdir -p $O; : > $O/sweep3.txt
for rep in 1 2; do for mm in 6 8 10 12 14 16; do RPT_SDF_MARCH_MIN_LANES=$mm timeout -k 10 150 python tools/ab_time.py c4 6 2>&1 | grep -v amdgpu.ids >> $O/sweep3.txt || exit 1; done; done
cat $O/sweep3.txt | cut -c1-40,60-200
