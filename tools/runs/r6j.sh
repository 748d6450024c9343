set -e
O=gpurun_out/r6j; mkdir -p $O
for v in sroom0 sroom24 sroom32 sroom40 sroom48; do
  export RPT_LIB=$PWD/rust-pathtracer_amd/variants/$v.so
  for k in 1 2; do python tools/ab_time.py c4 5 2>&1 | grep -v amdgpu.ids; done
done | tee $O/ab_c4.txt
