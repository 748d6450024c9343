set -e
O=gpurun_out/r6n; mkdir -p $O
for v in librpt_hip.so variants/he2.so variants/he8.so variants/hm4.so; do
  export RPT_LIB=$PWD/rust-pathtracer_amd/$v
  for k in 1 2; do python tools/ab_time.py c4 5 2>&1 | grep -v amdgpu.ids; done
done | tee $O/ab.txt
