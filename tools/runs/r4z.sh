set -e
mkdir -p gpurun_out/r4z
RPT_LIB=$PWD/rust-pathtracer_amd/variants/base.so python -m pytest tests -m gpu -x -q -k "sdf" > gpurun_out/r4z/tests.log 2>&1 || { tail -40 gpurun_out/r4z/tests.log; exit 1; }
tail -2 gpurun_out/r4z/tests.log
for i in 1 2; do bash tools/run_variants.sh tools/ab_time.py c4 8 >> gpurun_out/r4z/c4.txt 2>&1; done
cat gpurun_out/r4z/c4.txt
