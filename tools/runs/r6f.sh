set -e
O=gpurun_out/r6f; mkdir -p $O
for v in "" variants/sdf_pix.so variants/sdf_track.so; do
  if [ -z "$v" ]; then export RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip.so; else export RPT_LIB=$PWD/rust-pathtracer_amd/$v; fi
  for k in 1 2; do python tools/ab_time.py c4 5 2>&1 | grep -v amdgpu.ids; done
done | tee $O/ab_c4.txt
