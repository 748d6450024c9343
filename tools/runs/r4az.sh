set -e
mkdir -p gpurun_out/r4az
python tools/range_soak.py 60 1 2>/dev/null | grep scene > gpurun_out/r4az/shipped.txt
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python tools/range_soak.py 60 1 2>/dev/null | grep scene > gpurun_out/r4az/ab.txt
if diff gpurun_out/r4az/shipped.txt gpurun_out/r4az/ab.txt > gpurun_out/r4az/diff.txt; then echo "60 scenes: identical"; else echo DIFFERENT; head -20 gpurun_out/r4az/diff.txt; fi
head -12 gpurun_out/r4az/shipped.txt
