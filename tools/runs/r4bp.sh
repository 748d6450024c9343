set -e
mkdir -p gpurun_out/r4bp
python tools/range_soak.py 60 300 ref 2>/dev/null | grep scene > gpurun_out/r4bp/sized.txt
RPT_NO_SIZED_KERNELS=1 python tools/range_soak.py 60 300 ref 2>/dev/null | grep scene > gpurun_out/r4bp/general.txt
RPT_LIB=$PWD/rust-pathtracer_amd/librpt_hip_ab.so python tools/range_soak.py 60 300 ref 2>/dev/null | grep scene > gpurun_out/r4bp/ab.txt
if diff gpurun_out/r4bp/sized.txt gpurun_out/r4bp/general.txt > /dev/null && diff gpurun_out/r4bp/sized.txt gpurun_out/r4bp/ab.txt > /dev/null; then echo "60 scenes of the reference's sizes: sized = general = per-operation library"; else echo DIFFERENT; diff gpurun_out/r4bp/sized.txt gpurun_out/r4bp/general.txt | head; fi
head -4 gpurun_out/r4bp/sized.txt
