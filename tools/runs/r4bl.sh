set -e
mkdir -p gpurun_out/r4bl
bash tools/run_variants.sh tools/ab_time.py c2 6 >> gpurun_out/r4bl/x.txt 2>&1
cat gpurun_out/r4bl/x.txt
