set -e
mkdir -p gpurun_out/r4bq
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python tools/forms_soak.py > gpurun_out/r4bq/forms_soak.txt 2>&1 || { tail -20 gpurun_out/r4bq/forms_soak.txt; exit 1; }
grep -v amdgpu gpurun_out/r4bq/forms_soak.txt | tail -25
