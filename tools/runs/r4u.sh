set -e
mkdir -p gpurun_out/r4u
for i in 1 2; do bash tools/run_variants.sh tools/ab_time.py c5 8 >> gpurun_out/r4u/c5.txt 2>&1; done
cat gpurun_out/r4u/c5.txt
