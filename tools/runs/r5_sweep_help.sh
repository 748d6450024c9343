O=gpurun_out/r5_sweep; mkdir -p $O; : > $O/sweep2.txt
for mm in 4 8 12 16 24 32; do RPT_LIB=$PWD/rust-pathtracer_amd/variants/help2e4.so RPT_SDF_MARCH_MIN_LANES=$mm timeout -k 10 150 python tools/ab_time.py c4 4 2>&1 | grep -v amdgpu.ids >> $O/sweep2.txt || exit 1; done
cat $O/sweep2.txt | cut -c1-40,60-200
