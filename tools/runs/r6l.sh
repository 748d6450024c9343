set -e
O=gpurun_out/r6l; mkdir -p $O
for v in librpt_hip.so variants/sroom40.so; do
  export RPT_LIB=$PWD/rust-pathtracer_amd/$v
  for k in 1 2; do python tools/ab_time.py c4 5 2>&1 | grep -v amdgpu.ids; done
done | tee $O/ab.txt
export RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip.so
for room in 1 36 44 64; do RPT_SDF_SHADE_ROOM=$room python tools/ab_time.py c4 5 2>&1 | grep -v amdgpu.ids; done | tee -a $O/ab.txt
