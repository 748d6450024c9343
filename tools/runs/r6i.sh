set -e
O=gpurun_out/r6i; mkdir -p $O
for v in "" variants/mad24.so; do
  if [ -z "$v" ]; then export RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip.so; else export RPT_LIB=$PWD/rust-pathtracer_amd/$v; fi
  for k in 1 2; do python tools/ab_time.py c5 4 2>&1 | grep -v amdgpu.ids; done
done | tee $O/ab_c5.txt
