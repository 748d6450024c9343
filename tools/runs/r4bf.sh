set -e
mkdir -p gpurun_out/r4bf
for i in 1 2; do bash tools/run_variants.sh tools/ab_time.py c2 6 >> gpurun_out/r4bf/x.txt 2>&1; done
cat gpurun_out/r4bf/x.txt
